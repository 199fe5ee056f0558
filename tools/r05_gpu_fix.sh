#!/bin/bash
# round 5: the fix -- smpl_pose_bwd_kernel compiled without packed fp32 instructions (csrc/common.h STRAPS_NO_PACKED_FP32), LDS fence OFF (tools switch):
# the reproducers that showed 100-150 differing calls in 30 000, and the two-process probe
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 STRAPS_POSE_BWD_FENCE=0
run() { PROBE_SMPL_BWD=1 timeout 300 python tools/datagen_determinism_probe.py 4 30000 2>&1 | grep -v amdgpu | grep "^stages\|^calls whose" | cut -c1-260; }
( python tools/audit_packed_fp32.py
  for shape in "4 32 128" "4 16 256" "4 8 512"; do set -- $shape; echo "== convolution $1 x $2 x $2 x $3"; PROBE_CONV_B=$1 PROBE_CONV_HW=$2 PROBE_CONV_CH=$3 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run; done
  echo "== encoder forward"; PROBE_LOAD=enc_fwd run
  echo "== two processes (tools library, fence off)"
  PROBE_TOOLS=1 timeout 500 python tools/smpl_bwd_two_process_probe.py 600 2>&1 | grep -v amdgpu | tail -3 | cut -c1-300
) > gpurun_out/r05_packed_fp32_fix.txt 2>&1
cat gpurun_out/r05_packed_fp32_fix.txt
