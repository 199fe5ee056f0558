#!/bin/bash
# tools/conv_family_probe2.sh [reps] -- round 5, second batch.  The first (tools/conv_family_probe.sh, profiles/r05_conv_family_probe.txt) showed the
# LDS-table victim disturbed by exactly the aggressors that leave room for it on their CU (one four-wave workgroup of 284 registers per lane and
# 140-147 KB of LDS: x3 128x128 plain loop, halo 128x128), never by those that fill the register file (256x128 eight waves at 256 registers, the 3x3
# weight gradient) -- and not by the exact-fp32 kernel, which does leave room.  This batch asks what the co-resident convolution workgroup must DO:
#   x3:69 / x3:133 / x3:197   the 128x128 plain-loop kernel with operand copies only / without copies / MFMAs + barriers alone (ablations)
#   fp32:1                    exact-fp32 kernel, 128x128 explicitly
#   occupy 147 / 128 / 64     workgroups that only HOLD that much LDS and sleep (no traffic at all)
#   x3:5 + STRAPS_RASTER_LDS_EXTRA=20480 / 8192   the victim made too big (or not) to fit beside a 147 KB workgroup: co-residency off / on
cd "$(dirname "$0")/.."
REPS=${1:-600}
run() { timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c38-260; }
for kind in x3:5 x3:69 x3:133 x3:197 fp32:1; do PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv PROBE_CONV_KIND=$kind run; done
for kb in 147 128 64; do PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=occupy PROBE_OCCUPY_KB=$kb run; done
for extra in 20480 8192; do echo "victim asks for $extra extra LDS bytes:"; STRAPS_RASTER_LDS_EXTRA=$extra PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv PROBE_CONV_KIND=x3:5 run; done
