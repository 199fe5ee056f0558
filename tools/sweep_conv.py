#!/usr/bin/env python3
"""tools/sweep_conv.py -- per-layer timing of the implicit-GEMM conv (forward / data-gradient / weight-gradient)
for every tile configuration on the resnet18 shapes at B=64.  Run on the GPU box; prints TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NET = sys.argv[2] if len(sys.argv) > 2 else 'r18'
SHAPES = [('l1 3x3 s1', 64, 64, 64, 3, 1), ('l2.0 3x3 s2', 64, 64, 128, 3, 2), ('l2 3x3 s1', 32, 128, 128, 3, 1), ('l2 ds 1x1 s2', 64, 64, 128, 1, 2),
          ('l3.0 3x3 s2', 32, 128, 256, 3, 2), ('l3 3x3 s1', 16, 256, 256, 3, 1), ('l4.0 3x3 s2', 16, 256, 512, 3, 2), ('l4 3x3 s1', 8, 512, 512, 3, 1)]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


R50 = [('l1 1x1 64>64', 64, 64, 64, 1, 1), ('l1 1x1 64>256', 64, 64, 256, 1, 1), ('l1 1x1 256>64', 64, 256, 64, 1, 1),
       ('l2 1x1 256>128', 64, 256, 128, 1, 1), ('l2 1x1 128>512', 32, 128, 512, 1, 1), ('l2 1x1 512>128', 32, 512, 128, 1, 1),
       ('l2 ds 256>512 s2', 64, 256, 512, 1, 2), ('l3 1x1 512>256', 32, 512, 256, 1, 1), ('l3 1x1 256>1024', 16, 256, 1024, 1, 1),
       ('l3 1x1 1024>256', 16, 1024, 256, 1, 1), ('l3 ds 512>1024 s2', 32, 512, 1024, 1, 2), ('l4 1x1 1024>512', 16, 1024, 512, 1, 1),
       ('l4 1x1 512>2048', 8, 512, 2048, 1, 1), ('l4 1x1 2048>512', 8, 2048, 512, 1, 1), ('l4 ds 1024>2048 s2', 16, 1024, 2048, 1, 2),
       ('l2 3x3 s2 128', 64, 128, 128, 3, 2), ('l3 3x3 s2 256', 32, 256, 256, 3, 2), ('l4 3x3 s2 512', 16, 512, 512, 3, 2)]
if NET == 'r50':
    SHAPES = R50
for name, H, Cin, Cout, k, stride in SHAPES:
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wp, wd = torch.empty_like(w), torch.empty_like(w)
    L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
    L.straps_pack_conv_weight_dgrad(hipabi.ptr(w), hipabi.ptr(wd), Cout, Cin, k, k, None)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    dx = torch.empty_like(x)
    flops = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    row = '%-18s M=%7d N=%3d K=%4d |' % (name, B * Ho * Ho, Cout, Cin * k * k)
    for cfg in (1, 2, 3, 4):
        if cfg == 1 and Cout % 128:
            row += ' fwd%d   n/a ' % cfg
            continue
        part = torch.empty(L.straps_conv_stat_blocks(B, Ho, Ho, Cout, k * k * Cin, cfg), Cout, 2, device=dev)
        t = timeit(lambda: L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None))
        row += ' fwd%d %5.1f' % (cfg, flops / t / 1e12)
    row += ' |'
    for cfg in (1, 2, 3, 4):
        if cfg == 1 and Cin % 128:
            row += ' dg%d   n/a ' % cfg
            continue
        t = timeit(lambda: L.straps_conv_dgrad(hipabi.ptr(y), hipabi.ptr(wd), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None))
        row += ' dg%d %5.1f' % (cfg, flops / t / 1e12)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    dw = torch.empty_like(w)
    t = timeit(lambda: L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(y), hipabi.ptr(dw), hipabi.ptr(ws), B, H, H, Cin, Cout, k, k, stride, pad, 0, None))
    row += ' | wg %5.1f TF (%.0f us)' % (flops / t / 1e12, t * 1e6)
    print(row, flush=True)
