#!/bin/bash
# round 6, GPU call 3: the resnet50 step with the long 1x1 layers on the fp32-operand route -- parity (whole step vs float64, graph == eager), then the
# same-box A/B of the step (bench.py --config 3 with / without --no-x3f) and the headline
R=$PWD; O=$R/gpurun_out/r06_3; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_forward.py -m gpu -q -x -p no:cacheprovider -k "resnet50 or r50 or 50" > $O/pytest_r50.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r50.log
tail -15 $O/pytest_r50.log
for i in 1 2; do
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_r50_x3f_$i.json 2> $O/bench_r50_x3f_$i.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r50_x3f_$i.json').read().strip().splitlines()[-1]); print('x3f   ', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
timeout 300 python bench.py --config 3 --no-cpu-baseline --no-x3f > $O/bench_r50_planes_$i.json 2> $O/bench_r50_planes_$i.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r50_planes_$i.json').read().strip().splitlines()[-1]); print('planes', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
python - <<'PY'
import json
for tag in ('x3f_2','planes_2'):
    d=json.loads(open('gpurun_out/r06_3/bench_r50_%s.json'%tag).read().strip().splitlines()[-1])
    cl=d['roofline']['classes']
    tot=0
    print(tag)
    for k,c in sorted(cl.items(), key=lambda kv:-kv[1]['launches']*kv[1]['avg_launch_us']):
        if ' k1 ' in k:
            print('   %-46s %3d x %6.1f us'%(k, c['launches']//d['steps'], c['avg_launch_us'])); tot+=c['launches']*c['avg_launch_us']/d['steps']
    print('   1x1 total per step: %.0f us'%tot)
PY
