#!/bin/bash
mkdir -p gpurun_out/c21
timeout 1500 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_forward.py tests/test_gpu_backward.py -q -m gpu -k "driver_launches or calibration or ief or gemm_multi" -x -v --durations=5 > gpurun_out/c21/pytest.log 2>&1
tail -15 gpurun_out/c21/pytest.log
