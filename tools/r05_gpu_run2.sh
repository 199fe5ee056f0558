#!/bin/bash
# round 5, GPU call 2: the tests that failed in call 1 (tap-less parity classes under the look-ahead epilogue; the opt-in SMPL modes' bars), the second
# probe batch (what must a co-resident convolution workgroup DO to disturb the LDS-table victim?), and the cold tile sweep of the fused data gradient
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_forward.py tests/test_gpu_exchange.py tests/test_gpu_backward.py -q -p no:cacheprovider -rA 2>&1 | grep -v "^PASSED\|amdgpu.ids" | grep -A3 "real magnitudes\|FAILED\|passed\|failed\|Error" | tail -80 ) > gpurun_out/r05_run2_tests.txt 2>&1
( time timeout 600 bash tools/conv_family_probe2.sh 600 ) > gpurun_out/r05_conv_family_probe2.txt 2>&1
( time timeout 400 python tools/sweep_dgrad_bn_cold.py r18 ) > gpurun_out/r05_dgrad_bn_sweep_r18.txt 2>&1
( time timeout 600 python tools/sweep_dgrad_bn_cold.py r50 ) > gpurun_out/r05_dgrad_bn_sweep_r50.txt 2>&1
ls -la gpurun_out | tail
