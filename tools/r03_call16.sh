#!/bin/bash
# round 3, call 16: stages / split targets of the per-tap bf16x3 weight gradient
mkdir -p gpurun_out/c16
for cfg in "2 512 1536" "3 512 1536" "4 512 1536" "3 256 1536" "3 256 768" "3 512 768" "3 1024 2304"; do
  set -- $cfg
  echo "== NST=$1 WGS_BIG=$2 WGS_SMALL=$3" >> gpurun_out/c16/sweep.txt
  STRAPS_WGRAD_TAP_NST=$1 STRAPS_WGRAD_WGS_BIG=$2 STRAPS_WGRAD_WGS_SMALL=$3 timeout 600 python tools/sweep_wgrad_x3.py 64 2>&1 | grep -v "^l[1-4] 3x3 s1" >> gpurun_out/c16/sweep.txt
done
tail -3 gpurun_out/c16/sweep.txt
