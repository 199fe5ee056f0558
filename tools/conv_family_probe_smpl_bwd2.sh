#!/bin/bash
# tools/conv_family_probe_smpl_bwd2.sh [reps] -- which PART of a training step disturbs the SMPL-backward victim (the whole step does, no single convolution family
# at the 256-channel shape does)?  First batch: smplbwd / datagen / fwdbwd / enc_fwd / ief / adam / train -> only enc_fwd (and train).  Second batch, below: inside enc_fwd.
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
REPS=${1:-1500}
run() { PROBE_SMPL_BWD=1 timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260; }
PROBE_LOAD=stem_only run
PROBE_LOAD=pack run
echo "layer1 at 4 bodies (64 -> 64, 64 x 64: the single-buffer halo kernel):"; PROBE_CONV_B=4 PROBE_CONV_HW=64 PROBE_CONV_CH=64 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
echo "layer2 at 4 bodies (128 -> 128, 32 x 32: 64 x 64 tiles):"; PROBE_CONV_B=4 PROBE_CONV_HW=32 PROBE_CONV_CH=128 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
echo "layer3 at 4 bodies (256 -> 256, 16 x 16):"; PROBE_CONV_B=4 PROBE_CONV_HW=16 PROBE_CONV_CH=256 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
echo "layer4 at 4 bodies (512 -> 512, 8 x 8):"; PROBE_CONV_B=4 PROBE_CONV_HW=8 PROBE_CONV_CH=512 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
PROBE_LOAD=enc_fwd run
