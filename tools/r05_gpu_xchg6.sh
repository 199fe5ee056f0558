#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: ONE instruction.  The l = 1 step of (J[0], J[1]) is v_pk_fma_f32 ... op_sel:[0,1,0]; its LOW result comes out short of the product in lanes 48..63.
# STRAPS_POSE_BWD_DBG=4: the instruction written out, followed by a plain v_fma_f32 of the same registers; differences logged with the operands.
# =6: the same with every load awaited and a sleep in front.  =5: the compiler's code, loads awaited and a sleep before the first arithmetic.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-30000} > gpurun_out/_x.log 2>&1; grep -v amdgpu gpurun_out/_x.log | grep -A24 "^calls whose\|^stages\|^v_pk_fma" | grep -v "^   g\|^   G\|^   rel\|^   body" | cut -c1-700 || true; grep -q "^stages" gpurun_out/_x.log || tail -5 gpurun_out/_x.log; }
( echo "== DBG=4: packed instruction + plain fma of the same registers"; STRAPS_POSE_BWD_DBG=4 run
  echo "== DBG=6: the same, loads awaited + sleep in front"; STRAPS_POSE_BWD_DBG=6 run
  echo "== DBG=5: compiler's code, loads awaited + sleep before the arithmetic"; STRAPS_POSE_BWD_DBG=5 run
) > gpurun_out/r05_pose_bwd_packed_fma.txt 2>&1
cut -c1-500 gpurun_out/r05_pose_bwd_packed_fma.txt
