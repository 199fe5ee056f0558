#!/bin/bash
# round 6, GPU call 19: the smallest bucket of the automatic tile rule (128 <= 128x128-tile equivalents < 256: 128x64 tiles, three stages) against the other
# 128x64 / 64x64 forms with the lean epilogues in place (tools switch STRAPS_X3_LOW_CFG), same box
R=$PWD; O=$R/gpurun_out/r06_19; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
for f in 0 11 3 9 0; do
STRAPS_X3_LOW_CFG=$f timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic > $O/r18_$f.json 2> $O/r18_$f.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r18_$f.json').read().strip().splitlines()[-1]); print('r18 low_cfg=$f', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
STRAPS_X3_LOW_CFG=$f timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline > $O/r50_$f.json 2> $O/r50_$f.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r50_$f.json').read().strip().splitlines()[-1]); print('r50 low_cfg=$f', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
