#!/bin/bash
mkdir -p gpurun_out/c24
timeout 1500 python -m pytest tests/test_gpu_conv_x3.py -q -m gpu -k "dgrad_x3_vs_float64" --durations=4 > gpurun_out/c24/pytest.log 2>&1
tail -12 gpurun_out/c24/pytest.log
