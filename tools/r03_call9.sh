#!/bin/bash
R=$PWD; O=$R/gpurun_out/c9; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O/tr -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/tr.log 2>&1
ls $O/tr/*/ | head; 
python - <<'PY'
import csv, glob, os
O=os.environ.get('O','/root/repo/gpurun_out/c9')
d=glob.glob(O+'/tr/*/')[0]
kt=[r for r in csv.DictReader(open(glob.glob(d+'*kernel_trace.csv')[0]))]
mc=glob.glob(d+'*memory_copy_trace.csv')
print('memcopy file', mc)
if mc:
    rows=list(csv.DictReader(open(mc[0])))
    print(len(rows), rows[0].keys() if rows else None)
    for r in rows[-40:]:
        print({k:r[k] for k in list(r.keys())[:8]})
ks=sorted(kt,key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in ks]
idx=[i for i,n in enumerate(names) if 'copyBuffer' in n or 'fillBuffer' in n]
print('copy/fill kernels', len(idx), 'of', len(ks))
last=len(ks)
# the last step: print each copy with the kernel before and after
start=[i for i,n in enumerate(names) if 'philox' in n][-2]
for i in idx:
    if i>=start:
        print(names[i-1][:60].replace('(anonymous namespace)::',''),' -> ',names[i][:30],' -> ',names[i+1][:60].replace('(anonymous namespace)::','') if i+1<len(names) else '')
PY
rm -rf $O/tr
