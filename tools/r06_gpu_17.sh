#!/bin/bash
# round 6, GPU call 17: the size rules of the automatic tile choice re-checked with the lean epilogues (tools switch STRAPS_X3_RULE_CFG: one configuration for every
# 128-multiple class the rules decide; 0 = the rules), same box, both training steps
R=$PWD; O=$R/gpurun_out/r06_17; mkdir -p $O; export TMPDIR=/tmp STRAPS_TOOLS_NO_BUILD=1
cd $R
for f in 0 7 11 5 9 12 0; do
STRAPS_X3_RULE_CFG=$f timeout 300 python tools/with_tools_lib.py bench.py --no-cpu-baseline --no-other-configs --no-measure-traffic > $O/r18_$f.json 2> $O/r18_$f.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r18_$f.json').read().strip().splitlines()[-1]); print('r18 rule_cfg=$f', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
STRAPS_X3_RULE_CFG=$f timeout 300 python tools/with_tools_lib.py bench.py --config 3 --no-cpu-baseline > $O/r50_$f.json 2> $O/r50_$f.err < /dev/null; python -c "
import json; d=json.loads(open('$O/r50_$f.json').read().strip().splitlines()[-1]); print('r50 rule_cfg=$f', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'])"
done
