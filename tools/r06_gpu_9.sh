#!/bin/bash
# round 6, GPU call 9: kernel-trace statistics of the eager training step, resnet50 (32 bodies) and resnet18 (64 bodies), with the round's routes
R=$PWD; O=$R/gpurun_out/r06_9; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for tag in r50 r18; do
  if [ $tag = r50 ]; then A="--config 3"; else A=""; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python $R/bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/prof_$tag.log 2>&1
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${tag}_kernel_stats.csv
  rm -rf $O/prof_$tag
done
python - <<'PY'
import csv
for tag in ('r50','r18'):
    rows=list(csv.DictReader(open('/root/repo/gpurun_out/r06_9/%s_kernel_stats.csv'%tag)))
    tot=sum(float(r['TotalDurationNs']) for r in rows if 'sustained' not in r['Name'] and 'dense' not in r['Name'])
    print(tag, 'kernel time per step %.2f ms (13 steps traced)'%(tot/13e6))
    for r in rows[:34]:
        n=int(r['Calls']); t=float(r['TotalDurationNs'])
        print('  %7.1f us/step  %5.1f calls/step  avg %7.1f us  %s'%(t/13e3, n/13, t/n/1e3, r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:100]))
PY
