#!/bin/bash
# round 3, first GPU call: line-geometry probe of the L2 -> LDS path, the GPU suite (new float64 resnet50 / halo-kernel tests), bench
# lines with the clock field, kernel trace of the eager step at HEAD.
R=$PWD; O=$R/gpurun_out/c1; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 120 tools/bin/l2_line_probe > $O/l2_line_probe.txt 2>&1; tail -30 $O/l2_line_probe.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "relative gradient error|IEF ReLU|passed|failed|FAILED|Error" $O/pytest.log | tail -40
timeout 300 python bench.py 2>$O/bench_train.err | tail -1 > $O/bench_train.json; head -c 400 $O/bench_train.json; echo
timeout 300 python bench.py --config 4 2>$O/bench_smpl.err | tail -1 > $O/bench_smpl.json; head -c 400 $O/bench_smpl.json; echo
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap --no-stem-ab > $O/prof_train.log 2>&1
f=$(find $O/prof_train -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/r03_train_b64_kernel_stats.csv
rm -rf $O/prof_train
head -12 $O/r03_train_b64_kernel_stats.csv | cut -c1-160
