#!/bin/bash
# round 5: the weight re-pack beside the stem -- its test, the graph / two-rank files, then a same-box A/B (alternating)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_two_ranks.py tests/test_gpu_exchange.py -q -p no:cacheprovider -x 2>&1 | grep -v "amdgpu.ids" | tail -3 ) > gpurun_out/r05_run9_tests.txt 2>&1
tail -2 gpurun_out/r05_run9_tests.txt
for rep in 1 2 3; do for flag in "--no-pack-overlap" ""; do
  for cfg in "" "--config 3"; do
  timeout 300 python bench.py $cfg $flag --no-cpu-baseline --no-other-configs --no-measure-traffic --no-stem-ab 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', '$flag' or 'pack beside the stem', d['value'], d['ms_per_step'])"
done; done; done | tee gpurun_out/r05_pack_overlap_ab.txt
