#!/bin/bash
R=$PWD; O=$R/gpurun_out/c3; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "relative gradient error|IEF ReLU|passed|failed|FAILED|Error" $O/pytest.log | tail -40
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_train.err | tail -1 > $O/bench_train.json; python -c "
import json;d=json.load(open('$O/bench_train.json'));print({k:d.get(k) for k in ('value','ms_per_step','eager_ms_per_step','sclk_mhz')}, d['roofline'].get('sclk_mhz_during_measurement'), d['roofline']['avg_launch_us'])
for k,v in d['roofline']['classes'].items(): print(k,v)
print({k:(v['avg_launch_us'],v['ms_per_step']) for k,v in d['kernels'].items()})"
timeout 300 python tools/sweep_conv_x3.py > $O/sweep_conv_x3.txt 2>&1; cut -c1-400 $O/sweep_conv_x3.txt | tail -9
