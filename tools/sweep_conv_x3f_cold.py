#!/usr/bin/env python3
"""tools/sweep_conv_x3f_cold.py [r50|r18] -- the 1x1 layers on the fp32-operand route (csrc/conv_x3f.hip, round 6) against the plane route, timed COLD
(a 1 GiB fill between launches, one HIP-event pair per launch), forward (raw output + statistics partials; with and without the producer's
BatchNorm in the operand path) and data gradient with the step's full epilogue (addend + ReLU bits + fused BatchNorm sums where the step has them),
for the automatic tile (c0) and every explicit tile of each route.  The plane route's time does NOT include what produced its planes (the
bn_apply pass that writes them: 4 B read + 6 B written per element) -- the x3f columns replace that pass too.
Output of the round: profiles/r06_x3f_cold_sweep_*.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3 as _split3, weight_planes  # noqa: E402

L = hipabi.lib()
dev = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'r50'
# (name, H of the convolution's INPUT, Cin, Cout, stride, dgrad has addend + BatchNorm sums)
if which == 'r50':
    B = 32
    SHAPES = [('l1 64-256', 64, 64, 256, 1, 0), ('l1 256-64', 64, 256, 64, 1, 1), ('l1.0 64-64', 64, 64, 64, 1, 0), ('l2.0 256-128', 64, 256, 128, 1, 1), ('l2 128-512', 32, 128, 512, 1, 0),
              ('l2 512-128', 32, 512, 128, 1, 1), ('l2.0 ds 256-512 s2', 64, 256, 512, 2, 0), ('l3.0 512-256', 32, 512, 256, 1, 1), ('l3 256-1024', 16, 256, 1024, 1, 0),
              ('l3 1024-256', 16, 1024, 256, 1, 1), ('l3.0 ds 512-1024 s2', 32, 512, 1024, 2, 0), ('l4.0 1024-512', 16, 1024, 512, 1, 1), ('l4 512-2048', 8, 512, 2048, 1, 0),
              ('l4 2048-512', 8, 2048, 512, 1, 1), ('l4.0 ds 1024-2048 s2', 16, 1024, 2048, 2, 0)]
else:
    B = 64
    SHAPES = [('l2.0 ds 64-128 s2', 64, 64, 128, 2, 0), ('l3.0 ds 128-256 s2', 32, 128, 256, 2, 0), ('l4.0 ds 256-512 s2', 16, 256, 512, 2, 0)]
only = sys.argv[2] if len(sys.argv) > 2 else ''
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


for name, H, Cin, Cout, stride, full in SHAPES:
    if only and only not in name:
        continue
    Ho = (H - 1) // stride + 1
    torch.manual_seed(0)
    rows, orows = B * H * H, B * Ho * Ho
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * (2.0 / Cin) ** 0.5
    x3, xps = _split3(L, x)
    wp3, wps = weight_planes(L, w)
    wd3, wdps = weight_planes(L, w, dgrad=True)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    g = torch.randn(B, Ho, Ho, Cout, device=dev) * 1e-3
    g3, gps = _split3(L, g)
    dx = torch.empty(B, H, H, Cin, device=dev)
    raw = torch.randn(B, H, H, Cin, device=dev)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, Cin // 32), device=dev, dtype=torch.int32)
    addend = torch.randn(B, H, H, Cin, device=dev) * 1e-3 if full else None
    abits = torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, Cin // 32), device=dev, dtype=torch.int32) if full else None
    mean, invstd = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    asc, ash = torch.rand(Cin, device=dev) + 0.5, torch.rand(Cin, device=dev) - 0.5
    mb_in, mb_out = rows * Cin * 4e-6, orows * Cout * 4e-6
    print('%-22s M=%6d %4d->%4d  fp32 tensors: in %.1f MB, out %.1f MB' % (name, orows, Cin, Cout, mb_in, mb_out), flush=True)
    row = '   fwd   planes:'
    for cfg in (0, 2, 3, 5, 11, 12):
        if (cfg in (5, 12)) and Cout % 128:
            continue
        nblk = L.straps_conv_x3_stat_blocks(B, H, H, Cin, Cout, 1, 1, stride, 0, cfg)
        part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
        t = cold(lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, 1, 1,
                                                          stride, 0, cfg, None), 'fwd_x3'))
        row += ' c%d %5.1f' % (cfg, t)
    for bn in (0, 1):
        row += ' | x3f%s:' % ('+bn' if bn else '')
        for cfg in (0, 1, 2, 3, 5):
            if cfg == 3 and Cout % 128:
                continue
            nblk = L.straps_conv_x3f_stat_blocks(B, H, H, Cin, Cout, 1, 1, stride, 0, cfg)
            part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
            t = cold(lambda: hipabi.check(L.straps_conv_fwd_x3f(hipabi.ptr(x), hipabi.ptr(asc if bn else None), hipabi.ptr(ash if bn else None), bn, hipabi.ptr(wp3), wps, None, None,
                                                               None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, 1, 1, stride, 0, cfg, None), 'fwd_x3f'))
            row += ' c%d %5.1f' % (cfg, t)
    print(row, flush=True)
    row = '   dgrad planes:'
    for cfg in (0, 2, 3, 5, 11, 12):
        if (cfg in (5, 12)) and Cin % 128:
            continue
        nb = L.straps_conv_dgrad_x3_bn_blocks(B, H, H, Cin, Cout, 1, 1, stride, 0, cfg)
        bp = torch.empty(max(nb, 1) * Cin * 2, device=dev, dtype=torch.float64)
        if full:
            fn = lambda: hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, hipabi.ptr(addend), hipabi.ptr(dx), B, H, H, Cin, Cout, 1, 1, stride, 0,  # noqa: E731
                                                                    cfg, hipabi.ptr(raw), None, None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(bp), hipabi.ptr(abits),
                                                                    hipabi.ptr(bits), None), 'dgrad_x3_bn_bits')
        else:
            fn = lambda: hipabi.check(L.straps_conv_dgrad_x3_bn(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx), B, H, H, Cin, Cout, 1, 1, stride, 0,  # noqa: E731
                                                               cfg, hipabi.ptr(raw), None, hipabi.ptr(asc), hipabi.ptr(ash), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(bp), None),
                                      'dgrad_x3_bn')
        row += ' c%d %5.1f' % (cfg, cold(fn))
    row += ' | x3f:'
    for cfg in (0, 1, 2, 3, 5):
        if cfg == 3 and Cin % 128:
            continue
        nb = L.straps_conv_dgrad_x3f_bn_blocks(B, H, H, Cin, Cout, 1, 1, stride, 0, cfg)
        bp = torch.empty(max(nb, 1) * Cin * 2, device=dev, dtype=torch.float64)
        t = cold(lambda: hipabi.check(L.straps_conv_dgrad_x3f(hipabi.ptr(g), hipabi.ptr(wd3), wdps, hipabi.ptr(addend), hipabi.ptr(abits), hipabi.ptr(dx), B, H, H, Cin, Cout, 1, 1,
                                                             stride, 0, cfg, hipabi.ptr(raw), hipabi.ptr(bits if full else None), hipabi.ptr(None if full else asc),
                                                             hipabi.ptr(None if full else ash), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(bp), None), 'dgrad_x3f'))
        row += ' c%d %5.1f' % (cfg, t)
    print(row, flush=True)
