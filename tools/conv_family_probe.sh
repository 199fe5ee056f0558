#!/bin/bash
# tools/conv_family_probe.sh [reps] -- round 5, VERDICT item 1(b): WHICH convolution kernel family makes the LDS-table rasteriser (the victim of
# round 4, kept in the tools build: STRAPS_TOOLS_RASTER_FLAGS=-DSTRAPS_RASTER_LDS_TABLE) differ from itself?  One run of
# tools/datagen_determinism_probe.py per aggressor, each a captured graph of ONE kernel family replayed beside the victim:
#   x3          bf16x3 im2col kernel, automatic tile (256x128, pipelined loop, LDS-DMA dwordx4 operand copies)
#   x3:5        the same kernel, 128x128 four waves, plain loop, three-stage ring
#   halo        bf16x3 halo-patch kernel
#   wgrad3      bf16x3 3x3 weight gradient (LDS-DMA + ds_read_b64_tr_b16)
#   fp32        exact-fp32 implicit GEMM, LDS-DMA staging
#   fp32reg     the same kernel, register-staged operands: NO LDS-DMA at all
#   abl1        x3 kernel, operand copies + barriers only (no MFMA, no fragment reads)      } ablation instantiations
#   abl2        x3 kernel without its operand copies (MFMAs + fragment reads + barriers)    } of the tools build
#   abl3        x3 kernel, MFMAs + barriers alone                                           }
# The tools library must have been built with the LDS-table rasteriser BEFORE this runs on the GPU box (it travels with the snapshot):
#   STRAPS_TOOLS_RASTER_FLAGS=-DSTRAPS_RASTER_LDS_TABLE python -c "import straps_amd; from straps_amd import hipabi; hipabi.build(tools=True)"
cd "$(dirname "$0")/.."
REPS=${1:-600}
for kind in x3 x3:5 halo wgrad3 fp32 fp32reg abl1 abl2 abl3; do
  PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv PROBE_CONV_KIND=$kind timeout 300 python tools/datagen_determinism_probe.py 4 $REPS 2>&1 | grep -v amdgpu | tail -1 | cut -c38-260
done
