#!/bin/bash
mkdir -p gpurun_out/c17
for cfg in "2 256 1536" "2 128 1536" "3 128 1536" "2 384 1536"; do
  set -- $cfg
  echo "== NST=$1 WGS_BIG=$2 WGS_SMALL=$3" >> gpurun_out/c17/sweep.txt
  STRAPS_WGRAD_TAP_NST=$1 STRAPS_WGRAD_WGS_BIG=$2 STRAPS_WGRAD_WGS_SMALL=$3 timeout 600 python tools/sweep_wgrad_x3.py 64 2>&1 | grep "r50" | grep -v "3x3" >> gpurun_out/c17/sweep.txt
done
tail -3 gpurun_out/c17/sweep.txt
