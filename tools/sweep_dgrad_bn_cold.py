#!/usr/bin/env python3
"""tools/sweep_dgrad_bn_cold.py [r18|r50] -- the data gradient with the fused BatchNorm-backward sums (straps_conv_dgrad_x3_bn_bits: addend + ReLU bits +
raw + bit words in its epilogue) against its forward twin of the same FLOPs, per layer class of a training step, timed COLD (a 1 GiB fill between
launches, one HIP-event pair per launch), for the automatic tile rule (c0) and every explicit configuration the shape allows.

Round 5: the look-ahead epilogue (csrc/conv_igemm.h) took the classes whose grids run several rounds of workgroups per CU to within a few per cent
of their forward twins; the one-round classes (256x128 tiles: every CU is in its epilogue at the same time, 100 MB of operands and results cross
the memory system with no matrix work beside them) stayed 15 % behind.  This sweep asks whether another tile wins for THOSE launches now.
Run against the tools build with STRAPS_EPI=0 (python tools/with_tools_lib.py tools/sweep_dgrad_bn_cold.py ...) for the row-by-row epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3 as _split3, weight_planes  # noqa: E402

L = hipabi.lib()          # (tools/with_tools_lib.py selects the tools build before this runs)
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'r18'
# (name, H of the convolution's INPUT, Cin, Cout, k, stride, with_addend): the data gradient's output is [B, H, H, Cin]
if which == 'r50':
    B = 32
    SHAPES = [('l1 3x3 64-64', 64, 64, 64, 3, 1, 0), ('l1 1x1 64-256', 64, 64, 256, 1, 1, 0), ('l1 1x1 256-64', 64, 256, 64, 1, 1, 1), ('l2 1x1 256-128', 64, 256, 128, 1, 1, 1),
              ('l2 3x3 128-128', 32, 128, 128, 3, 1, 0), ('l2 1x1 128-512', 32, 128, 512, 1, 1, 0), ('l2 1x1 512-128', 32, 512, 128, 1, 1, 1), ('l3 3x3 256-256', 16, 256, 256, 3, 1, 0),
              ('l3 1x1 256-1024', 16, 256, 1024, 1, 1, 0), ('l3 1x1 1024-256', 16, 1024, 256, 1, 1, 1), ('l2.0 3x3 s2 128-128', 64, 128, 128, 3, 2, 0), ('l4 3x3 512-512', 8, 512, 512, 3, 1, 0),
              ('l4 1x1 512-2048', 8, 512, 2048, 1, 1, 0), ('l4 1x1 2048-512', 8, 2048, 512, 1, 1, 1)]
else:
    B = 64
    SHAPES = [('l1 3x3 64-64', 64, 64, 64, 3, 1, 1), ('l2 3x3 128-128', 32, 128, 128, 3, 1, 1), ('l3 3x3 256-256', 16, 256, 256, 3, 1, 1), ('l4 3x3 512-512', 8, 512, 512, 3, 1, 1),
              ('l2.0 3x3 s2 64-128', 64, 64, 128, 3, 2, 1), ('l3.0 3x3 s2 128-256', 32, 128, 256, 3, 2, 1), ('l4.0 3x3 s2 256-512', 16, 256, 512, 3, 2, 1)]
CFGS = (0, 3, 5, 7, 9, 11, 12, 256, 512, 1536)
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


for name, H, Cin, Cout, k, stride, with_add in SHAPES:
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    torch.manual_seed(0)
    rows = B * H * H
    x = torch.randn(B, H, H, Cin, device=dev).relu_()
    w = torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5
    x3, xps = _split3(L, x)
    wp3, wps = weight_planes(L, w)
    wd3, wdps = weight_planes(L, w, dgrad=True)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    g = torch.randn(B, Ho, Ho, Cout, device=dev) * 1e-3
    g3, gps = _split3(L, g)
    dx = torch.empty_like(x)
    raw = torch.randn(B, H, H, Cin, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, Cin // 32), device=dev, dtype=torch.int32)
    addend = torch.randn(B, H, H, Cin, device=dev) * 1e-3 if with_add else None
    abits = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, Cin // 32), device=dev, dtype=torch.int32) if with_add else None
    mean, invstd = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    row = '%-20s M=%6d %4d->%4d |' % (name, rows, Cin, Cout)
    # forward twin (raw output + BatchNorm partials, automatic tile): the same FLOPs
    nblk = L.straps_conv_x3_stat_blocks(B, H, H, Cin, Cout, k, k, stride, pad, 0)
    part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
    t = cold(lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k,
                                                      stride, pad, 0, None), 'fwd_x3'))
    row += ' fwd c0 %5.1f | plain dgrad c0' % t
    t = cold(lambda: hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'dgrad_x3'))
    row += ' %5.1f | dgrad+bn' % t
    for cfg in CFGS:
        if ((cfg & 15) in (1, 4, 5, 6, 8, 9, 12) and Cin % 128) or (cfg >= 256 and (k != 3 or stride != 1)):
            continue
        nb = L.straps_conv_dgrad_x3_bn_blocks(B, H, H, Cin, Cout, k, k, stride, pad, cfg)
        if nb <= 0:
            continue
        bp = torch.empty(nb * Cin * 2, device=dev, dtype=torch.float64)
        t = cold(lambda: hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad,
                                                                    cfg, hipabi.ptr(raw), None, None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(bp), hipabi.ptr(abits),
                                                                    hipabi.ptr(bits), None), 'dgrad_x3_bn_bits'))
        row += ' c%d %5.1f' % (cfg, t)
    print(row, flush=True)
