#!/usr/bin/env python3
"""tools/sweep_wgrad_x3.py -- the bf16x3 halo-patch weight gradient (conv_wgrad3x3_x3_kernel) against the exact-fp32 one on the resnet18
3x3 / stride-1 layer shapes at B=64: error of both against float64 (on a sub-batch), time per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [('l1 3x3 s1', 64, 64, 64), ('l2 3x3 s1', 32, 128, 128), ('l3 3x3 s1', 16, 256, 256), ('l4 3x3 s1', 8, 512, 512)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def split3(t):
    n = t.numel()
    ps = (n + 7) // 8 * 8
    out = torch.empty(3, ps, dtype=torch.int16, device=dev)
    hipabi.check(L.straps_split3_bf16(hipabi.ptr(t), hipabi.ptr(out), n, ps, None), 'split3')
    return out, ps


for name, H, Cin, Cout in SHAPES:
    k, stride, pad = 3, 1, 1
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=dev).relu_()
    g = torch.randn(B, H, H, Cout, device=dev) * 1e-3
    x3, xps = split3(x)
    g3, gps = split3(g)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    dw32 = torch.empty(Cout, Cin, k, k, device=dev)
    dw3 = torch.full((Cout, Cin, k, k), float('nan'), device=dev)
    hipabi.check(L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(dw32), hipabi.ptr(ws), B, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'wgrad')
    hipabi.check(L.straps_conv_wgrad_x3(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw3), hipabi.ptr(ws), B, H, H, Cin,
                                        Cout, k, k, stride, pad, 0, None), 'wgrad_x3')
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, k, k), g.permute(0, 3, 1, 2).double(), stride=stride, padding=pad)
    sc = ref.abs().max().item()
    e32 = (dw32.double() - ref).abs().max().item() / sc
    e3 = (dw3.double() - ref).abs().max().item() / sc
    flops = 2.0 * B * H * H * Cout * Cin * 9
    t32 = timeit(lambda: L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(dw32), hipabi.ptr(ws), B, H, H, Cin, Cout, k, k, stride, pad, 0, None))
    t3 = timeit(lambda: L.straps_conv_wgrad_x3(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw3), hipabi.ptr(ws), B, H, H,
                                               Cin, Cout, k, k, stride, pad, 0, None))
    print('%-10s err/max vs float64: fp32 %.1e  x3 %.1e | fp32 %6.1f us (%5.1f TF)  x3 %6.1f us (%5.1f TF fp32-equivalent)' % (
        name, e32, e3, t32 * 1e6, flops / t32 / 1e12, t3 * 1e6, flops / t3 / 1e12), flush=True)
