#!/bin/bash
# round 3, call 14: tile sweep of the bf16x3 implicit GEMM on resnet50's layer shapes (B = 32)
mkdir -p gpurun_out/c14
timeout 900 python tools/sweep_conv_x3.py 32 r50 > gpurun_out/c14/sweep_r50.txt 2>&1
tail -3 gpurun_out/c14/sweep_r50.txt
