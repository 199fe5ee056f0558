#!/bin/bash
# round 6, GPU call 4: what bounds the fp32-operand 1x1 kernels -- ablations of the 64 -> 256 and 256 -> 64 forward at resnet50's layer1 size
R=$PWD; O=$R/gpurun_out/r06_4; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
{
for shape in "64 256" "256 64"; do
  echo "== $shape"
  for abl in 0 1 2 4 8 3 5 6 7 15; do
    STRAPS_X3F_ABL=$abl python tools/with_tools_lib.py tools/x3f_ablate.py $shape 2>/dev/null
  done
  for wgs in 128 256 512 1024 100000; do
    STRAPS_X3F_STREAM_WGS=$wgs python tools/with_tools_lib.py tools/x3f_ablate.py $shape 2>/dev/null
  done
done
} > $O/x3f_ablate.txt 2>&1
cat $O/x3f_ablate.txt
