#!/bin/bash
mkdir -p gpurun_out/c20
for a in 0 1 2 3 4 5; do STRAPS_WGRAD3_ABL=$a timeout 200 python tools/wgrad3_ablate.py >> gpurun_out/c20/ablate.txt 2>&1; done
cat gpurun_out/c20/ablate.txt
