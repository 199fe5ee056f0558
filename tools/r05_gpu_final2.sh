#!/bin/bash
# round 5, final call: the whole GPU suite with every two-rank comparison made TWO_RANK_REPEAT times (the strict comparison is a plain assertion again),
# smoke, the driver-style default bench line, the exact-fp32 SMPL line (its kernel is one of those now compiled without packed fp32 instructions)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time TWO_RANK_REPEAT=${TWO_RANK_REPEAT:-10} timeout 1500 python -m pytest tests -m gpu -q -rxXf -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -25 ) > gpurun_out/r05_final_tests.txt 2>&1
grep "passed\|failed" gpurun_out/r05_final_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r05_bench_default.json
timeout 300 python bench.py --workload smpl --smpl-precision fp32 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r05_bench_smpl_1M_fp32_after.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_default.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('other_configs',{}).items() if isinstance(v,dict)})
d=json.load(open('gpurun_out/r05_bench_smpl_1M_fp32_after.json')); print('smpl fp32', d['value'], d['ms_per_step'])
PY
