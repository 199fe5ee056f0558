#!/usr/bin/env python3
"""tools/mfma_peak.py -- sustained fp32-MFMA rate of this board (register-resident operands, random data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
seed = torch.randn(512, device=dev)
for blocks, iters in ((256, 20000), (512, 20000), (1024, 20000)):
    out = torch.empty(blocks * 256, device=dev)
    L.straps_selftest_mfma_peak(hipabi.ptr(seed), hipabi.ptr(out), blocks, 100, None)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    L.straps_selftest_mfma_peak(hipabi.ptr(seed), hipabi.ptr(out), blocks, iters, None)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) * 1e-3
    flops = blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2
    print('blocks %4d: %.1f TFLOP/s (%.2f ms)' % (blocks, flops / t / 1e12, t * 1e3))
