#!/bin/bash
# tools/conv_family_probe_smpl_fwd.sh [reps] -- round 5: the kernels of the DATA stream that move data between lanes through the LDS unit (smpl_pose_kernel's kinematic
# chain: ds_bpermute) beside the small-grid bf16x3 convolutions (64 x 64 tiles) that disturb the SMPL-backward victim: is the production layout exposed?
cd "$(dirname "$0")/.."
export STRAPS_TOOLS_NO_BUILD=1
REPS=${1:-1500}
run() { timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300; }
for shape in "4 32 128" "4 16 256" "4 8 512"; do set -- $shape
  echo "victim: SMPL forward x2 (pose + vertices + joints kernels); load: bf16x3 conv $3 -> $3 at $1 x $2 x $2"; PROBE_SMPL=1 PROBE_CONV_B=$1 PROBE_CONV_HW=$2 PROBE_CONV_CH=$3 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
  echo "victim: the whole data generation (rasteriser .. network input); same load"; PROBE_CONV_B=$1 PROBE_CONV_HW=$2 PROBE_CONV_CH=$3 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run
done
