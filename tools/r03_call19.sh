#!/bin/bash
# round 3, call 19: full GPU suite + smoke + bench lines with the stride-2 data-gradient tile rule and the 256-workgroup per-tap weight gradient
mkdir -p gpurun_out/c19
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/c19/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c19/pytest.log
tail -3 gpurun_out/c19/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c19/smoke.log 2>&1; tail -1 gpurun_out/c19/smoke.log
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/c19/bench_train.json
timeout 600 python bench.py --config 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/c19/bench_r50.json
python - <<'PY'
import json
for f in ('train','r50'):
    d=json.loads(open('gpurun_out/c19/bench_%s.json'%f).read())
    print(f, d['value'], d['ms_per_step'], d.get('sclk_mhz'), d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
