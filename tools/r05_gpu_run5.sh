#!/bin/bash
# round 5, GPU call 5: the row (16-byte) look-ahead epilogue -- whole GPU suite, then the headline bench and the resnet50 shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids" | tail -40 ) > gpurun_out/r05_run5_tests.txt 2>&1
tail -5 gpurun_out/r05_run5_tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_bench_run5.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_run5.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['roofline']['classes'].items(): print(k, v['avg_launch_us'])
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('other_configs',{}).items() if isinstance(v,dict)})
PY
