#!/bin/bash
# tools/conv_family_probe_smpl_bwd2.sh [reps] -- which PART of a training step disturbs the SMPL-backward victim (the whole step does, no single convolution family does)?
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
REPS=${1:-1500}
run() { PROBE_SMPL_BWD=1 timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260; }
for load in smplbwd datagen fwdbwd enc_fwd ief adam; do PROBE_LOAD=$load run; done
PROBE_LOAD=train PROBE_LAYERS=18 run
