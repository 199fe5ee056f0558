#!/bin/bash
# kernel-trace stats of the eager training step (rocprofv3; summaries land in gpurun_out/prof_<tag>)
R=$PWD; export TMPDIR=/tmp; TAG=${1:-train}; shift
mkdir -p $R/gpurun_out/prof_$TAG; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph --no-overlap "$@" > $R/gpurun_out/prof_$TAG/bench.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
tail -1 $R/gpurun_out/prof_$TAG/bench.log | cut -c1-160
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print('%-60s n=%5s avg %9.1f us  tot %8.2f ms  %5.1f%%' % (r['Name'].replace('(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
print('total %.2f ms over all launches' % (tot / 1e6))
PY
