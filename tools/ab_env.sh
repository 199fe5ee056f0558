#!/bin/bash
# A/B of an environment switch on ONE box: usage  bash tools/ab_env.sh VAR=VALUE "pytest args"   (bench.py default line with the switch / without, twice)
R=$PWD; O=$R/gpurun_out/ab; mkdir -p $O; export TMPDIR=/tmp
cd $R
if [ -n "$2" ]; then
  timeout 600 python -m pytest $2 -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -6 $O/pytest.log
fi
for i in 1 2; do
  env $1 timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1 ', j['value'], j['ms_per_step'], j['kernels'].get('stem_wgrad_kernel'))"
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('default', j['value'], j['ms_per_step'], j['kernels'].get('stem_wgrad_kernel'))"
done
