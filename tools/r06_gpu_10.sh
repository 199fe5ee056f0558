#!/bin/bash
# round 6, GPU call 10: (a) the newer VOP3P forms alone in the victim loop (shorter loop = more meetings with the aggressor per form), with the known-bad
# form 0 as the control; (b) the whole GPU suite; (c) the default bench line
R=$PWD; O=$R/gpurun_out/r06_10; mkdir -p $O; export TMPDIR=/tmp
cd $R
{
  for ag in 3 9 1; do
    timeout 200 tools/bin/packed_fp32_hazard_repro 12000 $ag 0x1f81; echo "exit status $?"
    timeout 200 tools/bin/packed_fp32_hazard_repro 12000 $ag 0x0061; echo "exit status $?"
  done
} > $O/packed_forms_repro2.txt 2>&1
grep -v "not run\|as lane 0" $O/packed_forms_repro2.txt
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest rc=$?" >> $O/pytest_all.log; tail -15 $O/pytest_all.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json
