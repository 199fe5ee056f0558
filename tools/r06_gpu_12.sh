#!/bin/bash
# round 6, GPU call 12: the lean data-gradient epilogue WITH look-ahead in the plane kernels -- convolution / step parity, then the same-box A/B of both
# training steps through the tools build (STRAPS_X3_LEAN_DGRAD=0|1; the lean forward epilogue on in both)
R=$PWD; O=$R/gpurun_out/r06_12; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
timeout 1500 python -m pytest tests/test_gpu_conv_x3.py tests/test_gpu_conv_x3f.py tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -m gpu -q --maxfail=10 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
for i in 1 2; do
for lean in 1 0; do
STRAPS_X3_LEAN_DGRAD=$lean timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic > $O/bench_r18_lean$lean.json 2> $O/bench_r18_lean$lean.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r18_lean$lean.json').read().strip().splitlines()[-1]); print('r18 lean dgrad=$lean', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
STRAPS_X3_LEAN_DGRAD=$lean timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline > $O/bench_r50_lean$lean.json 2> $O/bench_r50_lean$lean.err; python -c "
import json,sys; d=json.loads(open('$O/bench_r50_lean$lean.json').read().strip().splitlines()[-1]); print('r50 lean dgrad=$lean', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
done
python - <<'PY'
import json
for tag in ('r18_lean1','r18_lean0'):
    d=json.loads(open('gpurun_out/r06_12/bench_%s.json'%tag).read().strip().splitlines()[-1])
    cl=d['roofline']['classes']
    print(tag)
    for k,c in sorted(cl.items(), key=lambda kv:-kv[1]['launches']*kv[1]['avg_launch_us']):
        if 'dgrad' in k: print('   %-46s %3d x %6.1f us'%(k, c['launches']//d['steps'], c['avg_launch_us']))
PY
