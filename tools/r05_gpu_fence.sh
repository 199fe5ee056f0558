#!/bin/bash
# round 5: does reserving 96 KB of (unused) LDS for smpl_pose_bwd_kernel -- no bf16x3 convolution workgroup then fits beside it -- silence the reproducer?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { PROBE_SMPL_BWD=1 timeout 300 python tools/datagen_determinism_probe.py 4 1500 2>&1 | grep -v amdgpu | tail -1 | cut -c1-260; }
( for shape in "4 32 128" "4 16 256" "4 8 512"; do set -- $shape; PROBE_CONV_B=$1 PROBE_CONV_HW=$2 PROBE_CONV_CH=$3 PROBE_LOAD=conv PROBE_CONV_KIND=x3 run; done
  PROBE_LOAD=enc_fwd run; PROBE_LOAD=train PROBE_LAYERS=18 run
  timeout 500 python tools/smpl_bwd_two_process_probe.py 600 2>&1 | grep -v amdgpu | tail -3 ) | tee gpurun_out/r05_smpl_bwd_fence.txt
timeout 300 python -m pytest tests/test_gpu_backward.py -q -p no:cacheprovider -x -k "smpl" 2>&1 | tail -2
