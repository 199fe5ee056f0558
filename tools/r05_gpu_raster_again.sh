#!/bin/bash
# round 5: is the rasteriser finding of round 4 (the LDS-table rasteriser differs from itself beside bf16x3 convolution kernels) the SAME instruction?
# Two tools libraries built beforehand (they travel under tools/bin/), both with the round-4 rasteriser (-DSTRAPS_RASTER_LDS_TABLE):
#   ..._raster_table_packed.so   its kernels as they were: packed fp32 instructions allowed (-DSTRAPS_ALLOW_PACKED_FP32): v_pk_mul_f32 ... op_sel:[0,1] in raster_face_kernel
#   ..._raster_table.so          the same source compiled without packed fp32 instructions (STRAPS_NO_PACKED_FP32, as the product)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export STRAPS_TOOLS_NO_BUILD=1 PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv
( for lib in raster_table_packed raster_table; do
    for kind in x3:5 halo; do
      echo "== $lib beside conv[$kind]"
      PROBE_TOOLS_LIB=$PWD/tools/bin/libstraps_hip_tools_$lib.so PROBE_CONV_KIND=$kind timeout 200 python tools/datagen_determinism_probe.py 4 ${1:-1500} 2>&1 | grep -v amdgpu | grep "^stages" | cut -c1-260
    done
  done ) > gpurun_out/r05_raster_table_packed_fp32.txt 2>&1
cat gpurun_out/r05_raster_table_packed_fp32.txt
