#!/bin/bash
mkdir -p gpurun_out/c23
bash tools/smpl_ablate.sh > gpurun_out/c23/smpl_ablate.txt 2>&1
cat gpurun_out/c23/smpl_ablate.txt
