#!/bin/bash
mkdir -p gpurun_out/c22
timeout 1500 python -m pytest tests/test_gpu_train_step.py -q -m gpu -k "bench_size_b64" -s --durations=3 > gpurun_out/c22/pytest.log 2>&1
tail -12 gpurun_out/c22/pytest.log | cut -c1-900; nproc; free -g | head -2
