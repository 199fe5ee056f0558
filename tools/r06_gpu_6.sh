#!/bin/bash
R=$PWD; O=$R/gpurun_out/r06_6; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_x3f.py -m gpu -q -x -p no:cacheprovider > $O/pytest_x3f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x3f.log
tail -8 $O/pytest_x3f.log
{
for shape in "64 256" "256 64"; do
  echo "== $shape"
  for abl in 0 15; do STRAPS_X3F_ABL=$abl python tools/with_tools_lib.py tools/x3f_ablate.py $shape 2>/dev/null; done
  STRAPS_X3F_LEAN=0 python tools/with_tools_lib.py tools/x3f_ablate.py $shape 2>/dev/null
done
} > $O/x3f_ablate_lean.txt 2>&1
cat $O/x3f_ablate_lean.txt
timeout 600 python tools/sweep_conv_x3f_cold.py r50 l > $O/x3f_cold_sweep_r50.txt 2>&1; cat $O/x3f_cold_sweep_r50.txt
