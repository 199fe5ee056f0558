#!/bin/bash
R=$PWD; O=$R/gpurun_out/c8; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python tools/find_torch_copies.py > $O/torch_copies.txt 2>&1; tail -25 $O/torch_copies.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_conv_x3.py tests/test_gpu_augment.py -m gpu -q -p no:cacheprovider -k "eval_mode or saturate or split_planes or regressor_forward_on or global_masked or cam_utils" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
