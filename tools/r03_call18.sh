#!/bin/bash
mkdir -p gpurun_out/c18
timeout 120 tools/bin/lds_tr_probe > gpurun_out/c18/lds_tr_probe.txt 2>&1
cat gpurun_out/c18/lds_tr_probe.txt
