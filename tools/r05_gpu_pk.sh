#!/bin/bash
# round 5: a victim of nothing but packed fp32 instructions beside the bf16x3 convolution (tools build, csrc/smpl_bwd.hip pk_victim_kernel)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_PK_VICTIM=1 PROBE_CONV_KIND=x3 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${PK_ITERS:-30000} > gpurun_out/_x.log 2>&1; grep -v amdgpu gpurun_out/_x.log | grep -A16 "^packed fp32 victim" | cut -c1-400 || true; grep -q "^packed fp32" gpurun_out/_x.log || tail -5 gpurun_out/_x.log; }
( echo "== beside the convolution"; PROBE_LOAD=conv run
  echo "== beside the convolution, 8 workgroups"; PROBE_LOAD=conv PROBE_PK_BLOCKS=8 run
  echo "== alone"; PROBE_LOAD=0 run
) > gpurun_out/r05_packed_fp32_victim.txt 2>&1
cut -c1-330 gpurun_out/r05_packed_fp32_victim.txt
