#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: the FIRST intermediate value of smpl_pose_bwd_kernel that differs in an event (tools build, STRAPS_POSE_BWD_DBG=1: workgroup 0 dumps them)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0 STRAPS_POSE_BWD_DBG=1
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-20000} > gpurun_out/_x.log 2>&1; grep -v amdgpu gpurun_out/_x.log | grep -A14 "^calls whose\|^stages" | cut -c1-600 || true; grep -q "^stages" gpurun_out/_x.log || tail -5 gpurun_out/_x.log; }
( for rep in 1 2; do echo "== dump on (fence off), run $rep"; run; done
  echo "== readlane-only exchanges"; STRAPS_POSE_BWD_XCHG=1 run
) > gpurun_out/r05_pose_bwd_first_field.txt 2>&1
cut -c1-400 gpurun_out/r05_pose_bwd_first_field.txt
