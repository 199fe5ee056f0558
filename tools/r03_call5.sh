#!/bin/bash
R=$PWD; O=$R/gpurun_out/c5; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -x -k "conv_fwd_x3 or conv_dgrad_x3 or fused_batchnorm or x3p or eval_mode" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python tools/x3_ablate.py > $O/x3_ablate.txt 2>&1; tail -5 $O/x3_ablate.txt
