#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: WHAT goes wrong in smpl_pose_bwd_kernel beside a bf16x3 convolution workgroup?  The reproducer of DESIGN section 1 (fence off) with the
# kernel's lane exchanges in five forms (csrc/smpl_bwd.hip, lane_get; tools build): 0 the product's __shfl, 1 v_readlane only (no LDS-unit instruction),
# 4 ds_bpermute with one in flight, 5 the product's __shfl checked against v_readlane, 2 ds_bpermute twice, one in flight, both checked
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0
( for x in 0 1 4 5 2; do
    echo "== exchange form $x (fence off)"
    STRAPS_POSE_BWD_XCHG=$x timeout 200 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-1500} 2>&1 | grep -v amdgpu | grep -A40 "^stages" | cut -c1-330
  done
  echo "== exchange form 5, fence ON"
  STRAPS_POSE_BWD_FENCE=1 STRAPS_POSE_BWD_XCHG=5 timeout 200 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-1500} 2>&1 | grep -v amdgpu | grep -A8 "^stages" | cut -c1-330
) > gpurun_out/r05_pose_bwd_exchange_forms.txt 2>&1
cat gpurun_out/r05_pose_bwd_exchange_forms.txt | cut -c1-250 | head -120
