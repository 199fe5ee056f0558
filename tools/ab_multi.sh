#!/bin/bash
# several environment settings against the default on ONE box: bash tools/ab_multi.sh "A=1" "B=2 C=3" ...
R=$PWD; export TMPDIR=/tmp; cd $R
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); k=j['kernels']; print('%-40s' % '$1', j['value'], j['ms_per_step'], 'stem_wgrad', k.get('stem_wgrad_kernel',{}).get('avg_launch_us'), 'stem', k.get('stem_kernel',{}).get('avg_launch_us'))"; }
for rep in 1 2; do
  run "X=0"
  for s in "$@"; do run "$s"; done
done
