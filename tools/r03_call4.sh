#!/bin/bash
R=$PWD; O=$R/gpurun_out/c4; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python tools/x3_ablate.py > $O/x3_ablate.txt 2>&1; tail -6 $O/x3_ablate.txt
timeout 600 python -m pytest tests/test_gpu_augment.py tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider -k "cam_utils or heatmaps or proxy_input" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
