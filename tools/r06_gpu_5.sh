#!/bin/bash
# round 6, GPU call 5: counter passes over the fp32-operand 1x1 kernels (64 -> 256 forward at resnet50's layer1 size), as they are and fully ablated
R=$PWD; O=$R/gpurun_out/r06_5; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd /tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS"
G2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
G4="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for abl in 0 15; do
  n=0
  for G in "$G1" "$G2" "$G3" "$G4"; do
    n=$((n+1))
    STRAPS_X3F_ABL=$abl timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $O/pmc_${abl}_$n -- python $R/tools/with_tools_lib.py $R/tools/x3f_ablate.py 64 256 > $O/pmc_${abl}_$n.log 2>&1
  done
done
python - $O > $O/x3f_pmc.txt <<'PY'
import csv, sys, glob, os
from collections import OrderedDict, defaultdict
O = sys.argv[1]
for abl in (0, 15):
    agg = OrderedDict()
    for n in (1, 2, 3, 4):
        fs = glob.glob(os.path.join(O, 'pmc_%d_%d' % (abl, n), '**', '*counter_collection.csv'), recursive=True)
        if not fs: continue
        seen = defaultdict(int)
        for r in csv.DictReader(open(fs[0])):
            nm = r['Kernel_Name']
            if 'x3f' not in nm and 'stream' not in nm: continue
            key = nm.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][-70:]
            d = agg.setdefault(key, {'n': 0, 'us': 0.0, 'c': defaultdict(float), 'cn': defaultdict(int)})
            d['c'][r['Counter_Name']] += float(r['Counter_Value']); d['cn'][r['Counter_Name']] += 1
            if r['Counter_Name'] in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQC_ICACHE_REQ', 'SQ_ACTIVE_INST_SCA'):
                d['n'] += 1; d['us'] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    print('== STRAPS_X3F_ABL=%d' % abl)
    for k, d in agg.items():
        print('%s: avg %.1f us over %d dispatch-passes' % (k, d['us'] / max(d['n'], 1), d['n']))
        w = d['c']['SQ_WAVES'] / max(d['cn']['SQ_WAVES'], 1)
        print('   per wave: ' + '  '.join('%s=%.0f' % (c.replace('SQ_', ''), v / d['cn'][c] / max(w, 1)) for c, v in d['c'].items() if c != 'SQ_WAVES') + '  waves=%d' % w)
PY
cat $O/x3f_pmc.txt
rm -rf $O/pmc_*/
