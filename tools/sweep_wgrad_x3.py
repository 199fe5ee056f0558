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
SHAPES = [('l1 3x3 s1', 64, 64, 64, 3, 1), ('l2 3x3 s1', 32, 128, 128, 3, 1), ('l3 3x3 s1', 16, 256, 256, 3, 1), ('l4 3x3 s1', 8, 512, 512, 3, 1),
          ('l2.0 3x3 s2', 64, 64, 128, 3, 2), ('l2 ds 1x1 s2', 64, 64, 128, 1, 2), ('l3.0 3x3 s2', 32, 128, 256, 3, 2), ('l3 ds 1x1 s2', 32, 128, 256, 1, 2),
          ('l4.0 3x3 s2', 16, 256, 512, 3, 2), ('l4 ds 1x1 s2', 16, 256, 512, 1, 2),
          ('r50 l1 64>256', 64, 64, 256, 1, 1), ('r50 l1 256>64', 64, 256, 64, 1, 1), ('r50 l2 256>128', 64, 256, 128, 1, 1), ('r50 l2 128>512', 32, 128, 512, 1, 1),
          ('r50 l2 512>128', 32, 512, 128, 1, 1), ('r50 l2 ds 256>512 s2', 64, 256, 512, 1, 2), ('r50 l3 512>256', 32, 512, 256, 1, 1), ('r50 l3 256>1024', 16, 256, 1024, 1, 1),
          ('r50 l3 1024>256', 16, 1024, 256, 1, 1), ('r50 l3 ds 512>1024 s2', 32, 512, 1024, 1, 2), ('r50 l4 1024>512', 16, 1024, 512, 1, 1),
          ('r50 l4 512>2048', 8, 512, 2048, 1, 1), ('r50 l4 2048>512', 8, 2048, 512, 1, 1), ('r50 l4 ds 1024>2048 s2', 16, 1024, 2048, 1, 2),
          ('r50 l2 3x3 s2', 64, 128, 128, 3, 2), ('r50 l3 3x3 s2', 32, 256, 256, 3, 2), ('r50 l4 3x3 s2', 16, 512, 512, 3, 2)]


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


from straps_amd.encoder_exec import split3 as _split3  # noqa: E402


def split3(t):          # chunk-major planes of an NHWC tensor
    return _split3(L, t)


for name, H, Cin, Cout, k, stride in SHAPES:
    pad = 1 if k == 3 else 0
    Bn = B if not name.startswith('r50') else B // 2
    Ho = (H + 2 * pad - k) // stride + 1
    torch.manual_seed(0)
    x = torch.randn(Bn, H, H, Cin, device=dev).relu_()
    g = torch.randn(Bn, Ho, Ho, Cout, device=dev) * 1e-3
    x3, xps = split3(x)
    g3, gps = split3(g)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(Bn, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    dw32 = torch.empty(Cout, Cin, k, k, device=dev)
    dw3 = torch.full((Cout, Cin, k, k), float('nan'), device=dev)
    hipabi.check(L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(dw32), hipabi.ptr(ws), Bn, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'wgrad')
    hipabi.check(L.straps_conv_wgrad_x3(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw3), hipabi.ptr(ws), Bn, H, H, Cin,
                                        Cout, k, k, stride, pad, 0, None), 'wgrad_x3')
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, k, k), g.permute(0, 3, 1, 2).double(), stride=stride, padding=pad)
    sc = ref.abs().max().item()
    e32 = (dw32.double() - ref).abs().max().item() / sc
    e3 = (dw3.double() - ref).abs().max().item() / sc
    flops = 2.0 * Bn * Ho * Ho * Cout * Cin * k * k
    t32 = timeit(lambda: L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(dw32), hipabi.ptr(ws), Bn, H, H, Cin, Cout, k, k, stride, pad, 0, None))
    t3 = timeit(lambda: L.straps_conv_wgrad_x3(hipabi.ptr(x), hipabi.ptr(g), hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw3), hipabi.ptr(ws), Bn, H, H,
                                               Cin, Cout, k, k, stride, pad, 0, None))
    print('%-16s err/max vs float64: fp32 %.1e  x3 %.1e | fp32 %6.1f us (%5.1f TF)  x3 %6.1f us (%5.1f TF fp32-equivalent)' % (
        name, e32, e3, t32 * 1e6, flops / t32 / 1e12, t3 * 1e6, flops / t3 / 1e12), flush=True)
