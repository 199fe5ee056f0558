#!/bin/bash
mkdir -p gpurun_out/c15
timeout 900 python tools/sweep_conv_x3.py 64 > gpurun_out/c15/sweep_r18.txt 2>&1
tail -2 gpurun_out/c15/sweep_r18.txt | cut -c1-200
