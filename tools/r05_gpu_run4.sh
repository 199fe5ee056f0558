#!/bin/bash
# round 5, GPU call 4b: the fragment-read aggressor with SHORT-LIVED workgroups (the convolution kernels' churn of 147 KB LDS allocations)
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
mkdir -p gpurun_out
run() { timeout 300 python tools/datagen_determinism_probe.py 4 ${REPS:-600} 2>&1 | grep -v amdgpu | tail -1 | cut -c38-260; }
( for load in frag fragsum; do echo "short-lived workgroups (12 chunks x 4096 workgroups per launch):"; PROBE_FRAG_TRIPS=12 PROBE_FRAG_BLOCKS=4096 PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=$load run; done
  echo "occupier, 147 KB, short-lived:"; PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=occupy PROBE_OCCUPY_US=10 PROBE_OCCUPY_BLOCKS=8192 run ) > gpurun_out/r05_conv_family_probe4.txt 2>&1
cat gpurun_out/r05_conv_family_probe4.txt
