#!/usr/bin/env python3
"""tools/proxy_nz_ab.py -- the training step's input construction: straps_build_proxy_input + straps_stem_nzmask (rounds 1-3: write 302 MB,
read them back for the non-zero map) against straps_build_proxy_input_nz (round 4: one pass, cells outside a joint's window not evaluated),
at 64 and 32 bodies, timed cold-ish (HIP events around 20 back-to-back calls; the output is larger than the Infinity Cache at 64 bodies)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
for B in (64, 32):
    g = torch.Generator().manual_seed(0)
    seg = torch.zeros(B, 256, 256)
    seg[:, 40:230, 90:170] = torch.randint(1, 7, (B, 190, 80), generator=g).float()      # a person-sized silhouette
    j = torch.rand(B, 17, 2, generator=g) * 150 + 50
    seg, j = seg.to(dev), j.to(dev)
    x = torch.empty(B, 18, 256, 256, device=dev)
    m = torch.empty(L.straps_stem_nzmask_words(B, 18, 256, 256), device=dev, dtype=torch.int32)

    def two():
        hipabi.check(L.straps_build_proxy_input(hipabi.ptr(seg), hipabi.ptr(j), hipabi.ptr(x), B, 17, 256, None), 'build')
        hipabi.check(L.straps_stem_nzmask(hipabi.ptr(x), hipabi.ptr(m), B, 18, 256, 256, None), 'nzmask')

    def one():
        hipabi.check(L.straps_build_proxy_input_nz(hipabi.ptr(seg), hipabi.ptr(j), hipabi.ptr(x), hipabi.ptr(m), B, 17, 256, 4, None), 'build_nz')

    for name, fn in (('two passes', two), ('one pass  ', one), ('two passes', two), ('one pass  ', one)):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print('B = %d  %s  %.1f us  (%.2f TB/s of the %.0f MB written)' % (B, name, us, x.numel() * 4 / us / 1e6, x.numel() * 4 / 1e6), flush=True)
