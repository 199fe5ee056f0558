#!/usr/bin/env python3
"""tools/sweep_conv_x3_cold.py [r50|r18] -- tile configurations of the bf16x3 implicit GEMM on the 1x1 (and stride-2) layers, timed COLD.

The tile rule of csrc/conv_x3.hip was read off sweeps that launch a layer back to back on the same tensors (tools/sweep_conv_x3.py): for the
byte-bound 1x1 layers those operands then sit in the 256 MB Infinity Cache, and the sweep favours big tiles that the same layer, fed from HBM
inside a training step, cannot keep busy (64 -> 256 forward: 49 us in the warm sweep, 74 us in the step).  Here a 1 GiB fill runs between
launches and every launch gets its own HIP-event pair: forward (raw output + BatchNorm partials, as in a training step) and data gradient,
for the automatic rule (c0) and every explicit tile configuration the shape allows."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3 as _split3, weight_planes  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'r50'
if which == 'r50':
    B = 32
    SHAPES = [('l1 1x1 64-64', 64, 64, 64, 1, 1), ('l1 1x1 64-256', 64, 64, 256, 1, 1), ('l1 1x1 256-64', 64, 256, 64, 1, 1), ('l2.0 1x1 256-128', 64, 256, 128, 1, 1),
              ('l2 1x1 128-512', 32, 128, 512, 1, 1), ('l2 ds 256-512 s2', 64, 256, 512, 1, 2), ('l2 1x1 512-128', 32, 512, 128, 1, 1), ('l3.0 1x1 512-256', 32, 512, 256, 1, 1),
              ('l3 1x1 256-1024', 16, 256, 1024, 1, 1), ('l3 ds 512-1024 s2', 32, 512, 1024, 1, 2), ('l3 1x1 1024-256', 16, 1024, 256, 1, 1), ('l4.0 1x1 1024-512', 16, 1024, 512, 1, 1),
              ('l4 1x1 512-2048', 8, 512, 2048, 1, 1), ('l4 ds 1024-2048 s2', 16, 1024, 2048, 1, 2), ('l4 1x1 2048-512', 8, 2048, 512, 1, 1),
              ('l1 3x3', 64, 64, 64, 3, 1), ('l2 3x3', 32, 128, 128, 3, 1), ('l3 3x3', 16, 256, 256, 3, 1), ('l4 3x3', 8, 512, 512, 3, 1)]
else:
    B = 64
    SHAPES = [('l1 3x3 s1', 64, 64, 64, 3, 1), ('l2.0 3x3 s2', 64, 64, 128, 3, 2), ('l2 3x3 s1', 32, 128, 128, 3, 1), ('l2 ds 1x1 s2', 64, 64, 128, 1, 2),
              ('l3.0 3x3 s2', 32, 128, 256, 3, 2), ('l3 3x3 s1', 16, 256, 256, 3, 1), ('l4.0 3x3 s2', 16, 256, 512, 3, 2), ('l4 3x3 s1', 8, 512, 512, 3, 1)]
CFGS = (0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 256, 512, 1536)      # (256 = im2col kernel only, 512 = halo-patch kernel wherever it applies, 1536 = its single-buffer form)
flush = torch.empty(1 << 28, device=dev)


def cold(fn, iters=4):
    ts = []
    for _ in range(iters + 1):
        flush.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return min(ts[1:])


for name, H, Cin, Cout, k, stride in SHAPES:
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=dev).relu_()
    w = torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5
    x3, xps = _split3(L, x)
    wp3, wps = weight_planes(L, w)
    wd3, wdps = weight_planes(L, w, dgrad=True)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    g = torch.randn(B, Ho, Ho, Cout, device=dev) * 1e-3
    g3, gps = _split3(L, g)
    dx = torch.empty_like(x)
    row = '%-18s M=%6d %4d->%4d | fwd' % (name, B * Ho * Ho, Cin, Cout)
    for cfg in CFGS:
        if ((cfg & 15) in (1, 4, 5, 6, 8, 9, 12) and Cout % 128) or (cfg >= 256 and (k != 3 or stride != 1)):
            continue
        nblk = L.straps_conv_x3_stat_blocks(B, H, H, Cin, Cout, k, k, stride, pad, cfg)
        part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
        t = cold(lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin,
                                                          Cout, k, k, stride, pad, cfg, None), 'fwd_x3'))
        row += ' c%d %5.1f' % (cfg, t)
    row += ' | dgrad'
    for cfg in CFGS:
        if ((cfg & 15) in (1, 4, 5, 6, 8, 9, 12) and Cin % 128) or (cfg >= 256 and (k != 3 or stride != 1)):
            continue
        t = cold(lambda: hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, cfg,
                                                            None), 'dgrad_x3'))
        row += ' c%d %5.1f' % (cfg, t)
    print(row, flush=True)
