#!/bin/bash
# one gpurun call: GPU parity tests, smoke, the bench lines of configs 1-4.  Results land in gpurun_out/check/.
R=$PWD; O=$R/gpurun_out/check; mkdir -p $O; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 --durations=12 -p no:cacheprovider ${PYTEST_ARGS} > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -6 $O/smoke.log
timeout 300 python bench.py > $O/bench_train.json 2> $O/bench_train.err; tail -c 1500 $O/bench_train.json
timeout 300 python bench.py --config 4 > $O/bench_smpl.json 2> $O/bench_smpl.err; tail -c 1500 $O/bench_smpl.json
timeout 300 python bench.py --config 4 --smpl-exact --no-cpu-baseline > $O/bench_smpl_exact.json 2> $O/bench_smpl_exact.err; tail -c 600 $O/bench_smpl_exact.json
timeout 300 python bench.py --config 3 --no-cpu-baseline > $O/bench_r50.json 2> $O/bench_r50.err; tail -c 600 $O/bench_r50.json
timeout 300 python bench.py --config 1 > $O/bench_fwd.json 2> $O/bench_fwd.err; tail -c 600 $O/bench_fwd.json
