#!/bin/bash
R=$PWD; O=$R/gpurun_out/r06_7; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_x3f.py -m gpu -q -p no:cacheprovider > $O/pytest_x3f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x3f.log
tail -8 $O/pytest_x3f.log
timeout 600 python tools/sweep_conv_x3f_cold.py r50 l1 > $O/x3f_cold_sweep_r50.txt 2>&1; cat $O/x3f_cold_sweep_r50.txt
