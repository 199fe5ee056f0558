#!/bin/bash
# round 6, last GPU call: smoke(), the default bench line, the whole GPU suite, then the profile refresh (RND=r06 tools/refresh_profiles.sh) with the final library
R=$PWD; O=$R/gpurun_out/r06_final; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default bench', d['value'], d['ms_per_step'], d['sclk_mhz'], d['roofline']['frac'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in d.get('other_configs',{}).items()})"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=" $O/pytest_gpu.log | tail -3
RND=r06 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; tail -2 $O/refresh.log
