#!/bin/bash
# one gpurun call at the end of a round: the GPU suite, smoke, the bench lines of configs[1..3] and the kernel trace of the eager training step.
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 200 python bench.py 2>$O/bench_train.err | tail -1 > $O/r02_bench_train_b64.json; head -c 300 $O/r02_bench_train_b64.json; echo
timeout 200 python bench.py --conv-precision fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_train_b64_fp32conv.json; head -c 120 $O/r02_bench_train_b64_fp32conv.json; echo
timeout 200 python bench.py --config 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_train_r50_b32.json; head -c 120 $O/r02_bench_train_r50_b32.json; echo
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/prof_train.log 2>&1
f=$(find $O/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r02_train_b64_kernel_stats.csv
rm -rf $O/prof_train
grep -E "pack_batched|stem_wgrad|bn_bwd_apply_kernel<true>|stem_tileany" $O/r02_train_b64_kernel_stats.csv | cut -c1-150
