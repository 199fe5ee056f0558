#!/bin/bash
# where does the matrix-pipe SMPL kernel's time go?  STRAPS_SMPL_ABLATE (compile-time variants of the kernel): bit 0 no stores, bit 1 no
# fragment loads after the first k-step, bit 2 no skinning MFMAs, bit 3 no blend MFMAs.  Results are wrong under ablation: timing only.
for ab in 0 1 2 3 7 15; do
  STRAPS_SMPL_PF=1 STRAPS_SMPL_ABLATE=$ab timeout 200 python tools/with_tools_lib.py bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$ab  %.3f ms/step' % d['ms_per_step'])"
done
for cfg in "1 0" "2 0" "2 4" "2 14"; do set -- $cfg
  STRAPS_SMPL_PF=$1 STRAPS_SMPL_RPC=$2 timeout 200 python tools/with_tools_lib.py bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PF=$1 RPC=$2  %.3f ms/step' % d['ms_per_step'])"
done
