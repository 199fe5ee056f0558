#!/bin/bash
# round 5, GPU call: accumulator-layout look-ahead epilogue with straight-line loads / one prefetch site / LDS-only barriers -- parity files, then benches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids" | tail -12 ) > gpurun_out/r05_run6_tests.txt 2>&1
tail -4 gpurun_out/r05_run6_tests.txt
for cfgargs in "" "--config 3"; do
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab $cfgargs 2>/dev/null | grep '^{' > gpurun_out/r05_bench_run6.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_run6.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['roofline']['classes'].items(): print('   ', k, v['avg_launch_us'])
PY
done
