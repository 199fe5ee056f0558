#!/bin/bash
# round 5: the packed-fp32 victim (0.12 % of its vulnerable instructions fail beside the bf16x3 convolution) as a DETECTOR: which neighbour does it take?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_PK_VICTIM=1 PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256
run() { timeout 120 python tools/datagen_determinism_probe.py 4 ${PK_ITERS:-8000} > gpurun_out/_x.log 2>&1; grep -v amdgpu gpurun_out/_x.log | grep -A14 "^packed fp32 victim" | grep "^packed\|by form" | cut -c1-420 || true; grep -q "^packed fp32" gpurun_out/_x.log || tail -3 gpurun_out/_x.log; }
( for kind in x3 halo wgrad3 fp32 fp32reg abl1 abl2 abl3 x3:133 x3:69; do echo "== conv kind $kind"; PROBE_LOAD=conv PROBE_CONV_KIND=$kind run; done
  for l in frag fragsum fragregs occupy smpl raster fill 1; do echo "== load $l"; PROBE_LOAD=$l run; done
  echo "== layer1-sized convolution at 64 bodies (the production shape: 256x128 pipelined tiles, large grid)"; PROBE_LOAD=conv PROBE_CONV_KIND=x3 PROBE_CONV_B=64 PROBE_CONV_HW=64 PROBE_CONV_CH=64 run
) > gpurun_out/r05_packed_fp32_detector.txt 2>&1
cut -c1-360 gpurun_out/r05_packed_fp32_detector.txt
