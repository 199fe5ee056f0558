#!/bin/bash
# round 3, call 11: eval-mode gradients against the float64 oracle evaluated on the GPU's decisions
mkdir -p gpurun_out/c11
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu -k "eval_mode" -s > gpurun_out/c11/pytest.log 2>&1
tail -5 gpurun_out/c11/pytest.log
