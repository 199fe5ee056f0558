#!/bin/bash
# round 5, last call: the LDS fence of smpl_pose_bwd_kernel is gone (the fix is the instruction, not the fence) -- the SMPL backward tests, the two-rank file with
# every comparison made TWO_RANK_REPEAT times, the training-step tests, smoke
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time TWO_RANK_REPEAT=${TWO_RANK_REPEAT:-4} timeout 400 python -m pytest tests/test_gpu_backward.py tests/test_gpu_two_ranks.py tests/test_gpu_train_step.py -m gpu -q -x -k "smpl or two_ranks or loss or train_step or step" -rxXf -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -8 ) > gpurun_out/r05_last_tests.txt 2>&1
cat gpurun_out/r05_last_tests.txt | tail -8
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo "smoke rc=$?"
