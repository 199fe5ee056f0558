#!/bin/bash
# round 6, last GPU calls: the whole GPU suite, smoke(), the default bench line, then the profile refresh (RND=r06 tools/refresh_profiles.sh) with the final library
R=$PWD; O=$R/gpurun_out/r06_final; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
RND=r06 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; tail -3 $O/refresh.log
