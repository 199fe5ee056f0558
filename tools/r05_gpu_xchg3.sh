#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: WHICH elements of the rotation gradient differ in an event (every event so far changed exactly 114 of them)?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0
run() { timeout 200 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-1500} 2>&1 | grep -v amdgpu | grep -A12 "^calls whose\|^stages" | cut -c1-400; }
( for rep in 1 2 3; do echo "== control (fence off), run $rep"; run; done
  echo "== readlane-only exchanges"; STRAPS_POSE_BWD_XCHG=1 run
) > gpurun_out/r05_pose_bwd_events.txt 2>&1
cut -c1-300 gpurun_out/r05_pose_bwd_events.txt
