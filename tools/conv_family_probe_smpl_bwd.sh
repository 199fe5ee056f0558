#!/bin/bash
# tools/conv_family_probe_smpl_bwd.sh [reps] -- round 5, last batch: the victim the two-rank test named (straps_smpl_bwd: its second kernel keeps NO LDS
# and moves data between lanes with ds_bpermute) beside one aggressor at a time, in ONE process (aggressor = a replayed graph on another stream)
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
REPS=${1:-1500}
run() { PROBE_TOOLS=1 PROBE_SMPL_BWD=1 timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260; }
PROBE_LOAD=0 run
for kind in x3 x3:5 x3:11 halo wgrad3 fp32:1 x3:69 x3:133 x3:197; do PROBE_LOAD=conv PROBE_CONV_KIND=$kind run; done
for load in frag fragregs occupy smpl raster fill; do PROBE_LOAD=$load run; done
PROBE_LOAD=train PROBE_LAYERS=18 run
