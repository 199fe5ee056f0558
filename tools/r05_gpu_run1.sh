#!/bin/bash
# round 5, GPU call 1: whole GPU suite (strict two-rank test with stage digests, look-ahead epilogue parity), the convolution-family probe of the
# LDS-table victim, and the same-box A/B of the epilogue forms (tools build: STRAPS_EPI=0 row by row, 1 look-ahead) on the resnet18 / resnet50 steps
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -rA 2>&1 | grep -v "^PASSED\|amdgpu.ids" | tail -150 ) > gpurun_out/r05_run1_tests.txt 2>&1
( time timeout 700 bash tools/conv_family_probe.sh 600 ) > gpurun_out/r05_conv_family_probe.txt 2>&1
for epi in 0 1 0 1; do
  STRAPS_EPI=$epi timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' > gpurun_out/r05_epi${epi}_r18_$RANDOM.json
done
for epi in 0 1; do
  STRAPS_EPI=$epi timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' > gpurun_out/r05_epi${epi}_r50.json
done
timeout 600 python bench.py 2>gpurun_out/r05_bench_run1.err | grep '^{' > gpurun_out/r05_bench_run1.json
ls -la gpurun_out | tail -20
