#!/bin/bash
# round 5: parity files after the last epilogue change, then the same-box A/B of the epilogue forms for the record (tools build: STRAPS_EPI=0 row by
# row, 1 look-ahead), alternating, resnet18 and resnet50
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_backward.py tests/test_gpu_train_step.py -q -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > gpurun_out/r05_run8_tests.txt 2>&1
tail -2 gpurun_out/r05_run8_tests.txt
rm -f gpurun_out/r05_ab2_*
for rep in 1 2; do for epi in 0 1; do
  STRAPS_EPI=$epi timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' > gpurun_out/r05_ab2_epi${epi}_r18_$rep.json
  STRAPS_EPI=$epi timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' > gpurun_out/r05_ab2_epi${epi}_r50_$rep.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_ab2_*')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d.get('sclk_mhz'))
PY
