#!/bin/bash
R=$PWD; O=$R/gpurun_out/c2; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 120 tools/bin/l2_line_probe > $O/l2_line_probe.txt 2>&1; tail -30 $O/l2_line_probe.txt
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_forward.py tests/test_gpu_predict.py -m gpu -q --maxfail=20 -p no:cacheprovider -s -k "resnet50 or ief or gemm_multi or whole_step or regressor or predict or autograd" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "relative gradient error|IEF ReLU|passed|failed|FAILED|Error" $O/pytest.log | tail -40
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_train.err | tail -1 > $O/bench_train.json; python -c "
import json;d=json.load(open('$O/bench_train.json'));print({k:d.get(k) for k in ('value','ms_per_step','eager_ms_per_step','sclk_mhz')}, d['roofline'].get('sclk_mhz_during_measurement'), d['roofline']['avg_launch_us'])"
timeout 300 python tools/find_torch_copies.py > $O/torch_copies.txt 2>&1; tail -25 $O/torch_copies.txt
