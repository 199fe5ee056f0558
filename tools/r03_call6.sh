#!/bin/bash
R=$PWD; O=$R/gpurun_out/c6; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 120 tools/bin/mfma_lds_probe > $O/mfma_lds_probe.txt 2>&1; cat $O/mfma_lds_probe.txt
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -s -k "eval_mode" > $O/pytest.log 2>&1; grep -E "eval-mode|passed|failed|image_encoder|ief_module" $O/pytest.log | head -60
