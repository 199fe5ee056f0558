#!/bin/bash
# tools/raster_variants.sh -- which ingredient of the round 2-4 rasteriser makes it non-reproducible beside bf16x3 convolution kernels?
# Rebuilds csrc/raster.hip of the TOOLS library with one reproducer macro at a time and runs tools/datagen_determinism_probe.py
# (PROBE_LOAD=conv) on each.  Result of round 4 (profiles/r04_raster_determinism.txt): LDS allocated but unused, a barrier alone, both, the
# table written + barrier but never read -- all reproducible; only the form that READS the table inside the sample loop is not (and there every
# value read back equals the value written, and the vertex coordinates re-read at the end of the lane equal the ones computed with).
cd "$(dirname "$0")/.."
for v in "-DSTRAPS_RASTER_LDS_TABLE" "-DSTRAPS_RASTER_DUMMY_LDS" "-DSTRAPS_RASTER_DUMMY_BARRIER" "-DSTRAPS_RASTER_DUMMY_LDS -DSTRAPS_RASTER_DUMMY_BARRIER" "-DSTRAPS_RASTER_WRITE_TABLE" ""; do
  echo "variant: ${v:-product}"
  rm -f tools/bin/build/raster.o
  STRAPS_TOOLS_RASTER_FLAGS="$v" python -c "
import straps_amd
from straps_amd import hipabi
hipabi.build(tools=True)" 2>&1 | grep -i " error"
  PROBE_TOOLS=1 PROBE_RASTER_PARTS=1 PROBE_LOAD=conv python tools/datagen_determinism_probe.py 4 ${REPS:-800} 2>&1 | grep -v amdgpu | tail -1 | cut -c60-250
done
rm -f tools/bin/build/raster.o
