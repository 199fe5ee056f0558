#!/bin/bash
R=$PWD; O=$R/gpurun_out/r06_16; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
for i in 1 2; do
for thr in 128 129 257; do
STRAPS_X3_SMALL_T128=$thr timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic > $O/r18_$thr.json 2> $O/r18_$thr.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r18_$thr.json').read().strip().splitlines()[-1]); print('r18 thr=$thr', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
STRAPS_X3_SMALL_T128=$thr timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline > $O/r50_$thr.json 2> $O/r50_$thr.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r50_$thr.json').read().strip().splitlines()[-1]); print('r50 thr=$thr', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
done
