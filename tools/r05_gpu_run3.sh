#!/bin/bash
# round 5, GPU call 3: the stand-alone victim / aggressor pair with the ingredient the library probes isolated -- a co-resident workgroup that
# streams operand fragments out of LDS (ds_read_b128), with (mode 4) and without (mode 5) MFMAs behind them; modes 0 / 3 again as the control
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( for mode in 4 5 3; do timeout 200 tools/bin/lds_read_hazard_repro 200 $mode 400; done ) > gpurun_out/r05_lds_read_hazard_repro.txt 2>&1
cat gpurun_out/r05_lds_read_hazard_repro.txt
