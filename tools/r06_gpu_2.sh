#!/bin/bash
# round 6, GPU call 2: the fp32-operand 1x1 route -- parity tests, then the cold sweep against the plane route on the resnet50 shapes
R=$PWD; O=$R/gpurun_out/r06_2; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_x3f.py -m gpu -q -x -p no:cacheprovider > $O/pytest_x3f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x3f.log
tail -30 $O/pytest_x3f.log
timeout 600 python tools/sweep_conv_x3f_cold.py r50 > $O/x3f_cold_sweep_r50.txt 2>&1; cat $O/x3f_cold_sweep_r50.txt
