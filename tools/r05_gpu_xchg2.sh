#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: the exchanges are NOT it (tools/r05_gpu_xchg.sh: 3.1 million checked, none wrong, and the readlane-only kernel differs too).  What the kernel
# READS, then: the chunk partials its producer (smpl_verts_bwd_kernel, the launch in front of it on the same stream) wrote.  Tools-build switches:
#   STRAPS_POSE_BWD_POISON=1  partials filled with NaN before the producer runs: a NaN result read what this call never wrote
#   STRAPS_POSE_BWD_SC=1      partials read with system-scope loads (past L1 and L2)
#   STRAPS_POSE_BWD_GAP=1     an empty kernel between producer and consumer
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0
run() { timeout 200 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-1500} 2>&1 | grep -v amdgpu | grep -A3 "^stages" | cut -c1-330; }
( echo "== control (fence off)"; run
  echo "== poison"; STRAPS_POSE_BWD_POISON=1 run
  echo "== system-scope loads of the partials"; STRAPS_POSE_BWD_SC=1 run
  echo "== empty kernel in between"; STRAPS_POSE_BWD_GAP=1 run
  echo "== poison + fence on"; STRAPS_POSE_BWD_FENCE=1 STRAPS_POSE_BWD_POISON=1 run
  echo "== poison, no background load"; PROBE_LOAD=0 STRAPS_POSE_BWD_POISON=1 run
) > gpurun_out/r05_pose_bwd_reads.txt 2>&1
cut -c1-260 gpurun_out/r05_pose_bwd_reads.txt
