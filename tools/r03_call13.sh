#!/bin/bash
# round 3, call 13: gradient tests against the float64 oracle on the GPU's decisions, final bars
mkdir -p gpurun_out/c13
timeout 3000 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py -q -m gpu -k "all_71 or all_165 or configs3 or eval_mode or reference_golden" -s > gpurun_out/c13/pytest.log 2>&1
tail -5 gpurun_out/c13/pytest.log
