#!/bin/bash
# round 5: whole suite with every two-rank comparison made TWO_RANK_REPEAT times; -rxX lists the xfail / xpass outcomes with their messages
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
TWO_RANK_REPEAT=10 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rxX -k "not bench_launched" 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r05_suite_hunt_fenced.txt
grep "passed\|failed" gpurun_out/r05_suite_hunt_fenced.txt | tail -1
grep -E "^XPASS|^XFAIL" gpurun_out/r05_suite_hunt_fenced.txt | cut -c1-600
grep "first stage that differs" gpurun_out/r05_suite_hunt_fenced.txt | cut -c1-900 | head -4
