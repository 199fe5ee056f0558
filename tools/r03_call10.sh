#!/bin/bash
R=$PWD; O=$R/gpurun_out/c10; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_train.err | tail -1 > $O/bench_train.json; python -c "
import json;d=json.load(open('$O/bench_train.json'));print({k:d.get(k) for k in ('value','ms_per_step','eager_ms_per_step','sclk_mhz')}, d['roofline']['avg_launch_us'], d['roofline'].get('sustained_mfma'), d['roofline']['frac'])"
tail -3 $O/bench_train.err
timeout 300 python bench.py --config 4 --no-cpu-baseline 2>$O/bench_smpl.err | tail -1 > $O/bench_smpl.json; python -c "
import json;d=json.load(open('$O/bench_smpl.json'));print({k:d.get(k) for k in ('value','ms_per_step','sclk_mhz','reduced_precision_modes')}, d['roofline'].get('sustained_mfma'), d['roofline']['frac'], d['roofline']['mfma_side'])"
tail -3 $O/bench_smpl.err
timeout 300 python bench.py --config 3 --no-cpu-baseline 2>$O/bench_r50.err | tail -1 > $O/bench_r50.json; python -c "
import json;d=json.load(open('$O/bench_r50.json'));print({k:d.get(k) for k in ('value','ms_per_step','eager_ms_per_step','sclk_mhz')}, d['roofline']['avg_launch_us'], d['roofline']['frac'])
for k,v in d['kernels'].items(): print(k,v)"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r50 -- python $R/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/prof_r50.log 2>&1
f=$(find $O/prof_r50 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_train_r50_b32_kernel_stats.csv
rm -rf $O/prof_r50
