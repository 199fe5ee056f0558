#!/bin/bash
# round 6, GPU call 1: (a) the other VOP3P operand-select forms beside the synthetic aggressors (tools/packed_fp32_hazard_repro.hip, 13 forms);
# (b) the dense sustained-MFMA probe with its MfmaUtil counter pass; (c) same-round baseline bench lines of HEAD before the round's kernel work.
R=$PWD; O=$R/gpurun_out/r06_1; mkdir -p $O; export TMPDIR=/tmp
cd $R
{
  for ag in 0 3 9; do timeout 120 tools/bin/packed_fp32_hazard_repro 6000 $ag; echo "exit status $?"; done
} > $O/packed_forms_repro.txt 2>&1
cat $O/packed_forms_repro.txt
timeout 200 python tools/mfma_dense_probe.py 3 > $O/mfma_dense_probe.txt 2>&1; cat $O/mfma_dense_probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_dense -- python $R/tools/mfma_dense_probe.py 1 > $O/pmc_dense.log 2>&1
f=$(find $O/pmc_dense -name '*counter_collection.csv' | head -1)
python - "$f" > $O/mfma_dense_probe_pmc.txt 2>&1 <<'PY'
import csv, sys
from collections import OrderedDict
rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if 'mfma' not in r['Kernel_Name']:
        continue
    d = rows.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'].split('(')[0], 'grid': r.get('Grid_Size'), 'us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3})
    d[r['Counter_Name']] = float(r['Counter_Value'])
for k, d in rows.items():
    print('dispatch %4s %-40s grid %8s %9.1f us  MfmaUtil %6.2f  GRBM_GUI_ACTIVE %.4g' % (k, d['name'][-40:], d['grid'], d['us'], d.get('MfmaUtil', float('nan')), d.get('GRBM_GUI_ACTIVE', float('nan'))))
PY
cat $O/mfma_dense_probe_pmc.txt
rm -rf $O/pmc_dense
cd $R
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
