#!/bin/bash
# round 5, candidate check: whole GPU suite the way the driver runs it, smoke, the two training benches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -12 ) > gpurun_out/r05_run7_tests.txt 2>&1
tail -6 gpurun_out/r05_run7_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
for cfgargs in "" "--config 3"; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab $cfgargs 2>/dev/null | grep '^{' > gpurun_out/r05_bench_run7.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_run7.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
done
