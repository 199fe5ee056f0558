#!/bin/bash
# round 5, final call: hipGraph replay against eager launches over 60 steps (three passes of tools/graph_long_run.py), the whole GPU suite the way
# the driver runs it, smoke, the driver-style default bench line, the forced-exchange bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( echo "# tools/graph_long_run.py 60, three passes, MI355X, round 5 (look-ahead epilogue); every line must read == eager: True"
  for rep in 1 2 3; do timeout 900 python tools/graph_long_run.py 60 2>&1 | grep -v amdgpu; done ) > gpurun_out/r05_graph_long_run.txt 2>&1
grep -c "== eager: True" gpurun_out/r05_graph_long_run.txt; grep -c "== eager: False" gpurun_out/r05_graph_long_run.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -12 ) > gpurun_out/r05_final_tests.txt 2>&1
grep "passed\|failed" gpurun_out/r05_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r05_bench_default.json
for be in torch rccl; do timeout 300 python bench.py --force-exchange --exchange-backend $be --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' > gpurun_out/r05_bench_force_exchange_$be.json; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_default.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('other_configs',{}).items() if isinstance(v,dict)})
for be in ('torch','rccl'):
    d=json.load(open('gpurun_out/r05_bench_force_exchange_%s.json'%be)); print(be, d['value'], d['ms_per_step'], d['ranks'])
PY
