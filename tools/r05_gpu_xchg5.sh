#!/bin/bash
# (run when smpl_pose_bwd_kernel still held packed fp32 instructions; to repeat: build the tools library with STRAPS_TOOLS_SMPL_BWD_FLAGS=-DSTRAPS_POSE_BWD_PACKED first)
# round 5: the first wrong value is J[0] short of its l = 1 term = the SECOND dword of one global_load_dwordx4, lanes 48..63.  The load written out in assembly and
# awaited on the spot (tools build, STRAPS_POSE_BWD_DBG=2; =3: ~500 cycles of sleep behind the wait): its four result registers copied early and late
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_SMPL_BWD=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 STRAPS_POSE_BWD_FENCE=0
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${XCHG_ITERS:-30000} > gpurun_out/_x.log 2>&1; grep -v amdgpu gpurun_out/_x.log | grep -A20 "^calls whose\|^stages" | grep -v "^   g\|^   G\|^   rel" | cut -c1-900 || true; grep -q "^stages" gpurun_out/_x.log || tail -5 gpurun_out/_x.log; }
( for rep in 1 2; do echo "== load in assembly, awaited on the spot (DBG=2), run $rep"; STRAPS_POSE_BWD_DBG=2 run; done
  echo "== the same with a sleep behind the wait (DBG=3)"; STRAPS_POSE_BWD_DBG=3 run
) > gpurun_out/r05_pose_bwd_load_registers.txt 2>&1
cut -c1-700 gpurun_out/r05_pose_bwd_load_registers.txt
