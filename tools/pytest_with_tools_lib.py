#!/usr/bin/env python3
"""Run pytest with the TOOLS build of the library loaded (tools/bin/libstraps_hip_tools.so), so that a test file can be run under one of its A/B
environment switches (e.g. STRAPS_BN_TILED=2 python tools/pytest_with_tools_lib.py tests/test_gpu_backward.py -m gpu -q -k bn)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

if __name__ == '__main__':
    import pytest
    hipabi.use_library(hipabi.build(tools=True))
    os.chdir(ROOT)
    sys.exit(pytest.main(sys.argv[1:]))
