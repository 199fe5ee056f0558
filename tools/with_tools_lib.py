#!/usr/bin/env python3
"""Run a script of this repository against the TOOLS build of the library (tools/bin/libstraps_hip_tools.so, compiled with
-DSTRAPS_TOOLS): the build that carries the ablation instantiations -- which compute WRONG results by design -- and honours the
STRAPS_* A/B environment switches (STRAPS_SMPL_ABLATE, STRAPS_SMPL_PF, STRAPS_SMPL_RPC, STRAPS_WGRAD3_ABL, STRAPS_WGRAD_*,
STRAPS_STEM_WGRAD_*).  The product library has none of them and never reads the environment.

    python tools/with_tools_lib.py bench.py --config 4 --no-cpu-baseline
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    hipabi.use_library(hipabi.build(tools=True))
    # the two Python-side A/B switches (both variants are correct; the product package itself never reads the environment)
    from straps_amd import autograd_ops
    if os.environ.get('STRAPS_NO_FUSED_BN_SUMS', '0') == '1':
        autograd_ops._FUSE_BN_SUMS = False
    if os.environ.get('STRAPS_DENSE_STEM_TAIL', '0') == '1':
        autograd_ops._SPARSE_STEM_TAIL = False
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    runpy.run_path(script, run_name='__main__')
