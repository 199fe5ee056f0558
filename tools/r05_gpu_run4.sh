#!/bin/bash
# round 5, GPU call 4c: the fragment-read aggressor with the convolution kernel's register footprint (PROBE_LOAD=fragregs), long- and short-lived, and
# the 128x128 convolution kernel again as the control
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
mkdir -p gpurun_out
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${REPS:-600} 2>&1 | grep -v amdgpu | tail -1 | cut -c38-260; }
( echo "long-lived (400 chunks x 256 workgroups per launch):"; PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=fragregs run
  echo "short-lived (12 chunks x 4096 workgroups per launch):"; PROBE_FRAG_TRIPS=12 PROBE_FRAG_BLOCKS=4096 PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=fragregs run
  PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3:5 run ) > gpurun_out/r05_conv_family_probe5.txt 2>&1
cat gpurun_out/r05_conv_family_probe5.txt
