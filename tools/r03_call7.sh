#!/bin/bash
R=$PWD; O=$R/gpurun_out/c7; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "eval-mode|passed|failed|FAILED|Error|error" $O/pytest.log | tail -30
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_train.err | tail -1 > $O/bench_train.json; python -c "
import json;d=json.load(open('$O/bench_train.json'));print({k:d.get(k) for k in ('value','ms_per_step','eager_ms_per_step','sclk_mhz')}, d['roofline']['avg_launch_us'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/prof_train.log 2>&1
f=$(find $O/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_train_b64_kernel_stats.csv
rm -rf $O/prof_train
