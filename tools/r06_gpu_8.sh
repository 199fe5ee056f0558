#!/bin/bash
# round 6, GPU call 8: the fp32-operand weight gradient + operand-path BatchNorm in the step -- kernel parity, the resnet50 / resnet18 step tests, then the
# same-box A/B over the row threshold of the route
R=$PWD; O=$R/gpurun_out/r06_8; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_conv_x3f.py -m gpu -q -p no:cacheprovider > $O/pytest_x3f.log 2>&1; echo "pytest rc=$?" >> $O/pytest_x3f.log; tail -5 $O/pytest_x3f.log
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_forward.py -m gpu -q -x -p no:cacheprovider > $O/pytest_step.log 2>&1; echo "pytest rc=$?" >> $O/pytest_step.log; tail -8 $O/pytest_step.log
for rows in 0 65536 16384 4096 2048; do
timeout 300 python bench.py --config 3 --no-cpu-baseline --x3f-min-rows $rows > $O/bench_r50_rows$rows.json 2> $O/bench_r50_rows$rows.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r50_rows$rows.json').read().strip().splitlines()[-1]); print('r50 min_rows $rows', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
timeout 300 python bench.py --config 3 --no-cpu-baseline --no-x3f-operand-bn > $O/bench_r50_noopbn.json 2> $O/bench_r50_noopbn.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r50_noopbn.json').read().strip().splitlines()[-1]); print('r50 no operand bn', d['value'], d['ms_per_step'], d['sclk_mhz'])"
for rows in 0 2048; do
timeout 300 python bench.py --no-cpu-baseline --x3f-min-rows $rows > $O/bench_r18_rows$rows.json 2> $O/bench_r18_rows$rows.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r18_rows$rows.json').read().strip().splitlines()[-1]); print('r18 min_rows $rows', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
