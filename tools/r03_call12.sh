#!/bin/bash
# round 3, call 12: whole-step gradients against the float64 oracle evaluated on the GPU's decisions (diagnostic print)
mkdir -p gpurun_out/c12
timeout 2400 python -m pytest tests/test_gpu_train_step.py -q -m gpu -k "all_71 or all_165" -s > gpurun_out/c12/pytest.log 2>&1
tail -5 gpurun_out/c12/pytest.log
