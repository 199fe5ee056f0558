#!/usr/bin/env python3
"""tools/sweep_conv_x3.py -- the three-plane bf16 implicit GEMM (csrc/conv_x3.hip) against the exact-fp32 one (csrc/conv.hip) on the
resnet18 layer shapes at B=64: error of both against a float64 convolution, time per launch per tile configuration, time of the
operand split.  Run on the GPU box."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [('l1 3x3 s1', 64, 64, 64, 3, 1), ('l2.0 3x3 s2', 64, 64, 128, 3, 2), ('l2 3x3 s1', 32, 128, 128, 3, 1), ('l2 ds 1x1 s2', 64, 64, 128, 1, 2),
          ('l3.0 3x3 s2', 32, 128, 256, 3, 2), ('l3 3x3 s1', 16, 256, 256, 3, 1), ('l4.0 3x3 s2', 16, 256, 512, 3, 2), ('l4 3x3 s1', 8, 512, 512, 3, 1)]
if len(sys.argv) > 2 and sys.argv[2] == 'r50':          # the Bottleneck encoder's distinct shapes (models/resnet.py:80-121), use with B = 32
    SHAPES = [('l1 1x1 64-64', 64, 64, 64, 1, 1), ('l1 3x3', 64, 64, 64, 3, 1), ('l1 1x1 64-256', 64, 64, 256, 1, 1), ('l1 1x1 256-64', 64, 256, 64, 1, 1),
              ('l2.0 1x1 256-128', 64, 256, 128, 1, 1), ('l2.0 3x3 s2', 64, 128, 128, 3, 2), ('l2 1x1 128-512', 32, 128, 512, 1, 1), ('l2 ds 256-512 s2', 64, 256, 512, 1, 2),
              ('l2 1x1 512-128', 32, 512, 128, 1, 1), ('l2 3x3', 32, 128, 128, 3, 1),
              ('l3.0 1x1 512-256', 32, 512, 256, 1, 1), ('l3.0 3x3 s2', 32, 256, 256, 3, 2), ('l3 1x1 256-1024', 16, 256, 1024, 1, 1), ('l3 ds 512-1024 s2', 32, 512, 1024, 1, 2),
              ('l3 1x1 1024-256', 16, 1024, 256, 1, 1), ('l3 3x3', 16, 256, 256, 3, 1),
              ('l4.0 1x1 1024-512', 16, 1024, 512, 1, 1), ('l4.0 3x3 s2', 16, 512, 512, 3, 2), ('l4 1x1 512-2048', 8, 512, 2048, 1, 1), ('l4 ds 1024-2048 s2', 16, 1024, 2048, 1, 2),
              ('l4 1x1 2048-512', 8, 2048, 512, 1, 1), ('l4 3x3', 8, 512, 512, 3, 1)]


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


from straps_amd.encoder_exec import split3 as _split3, weight_planes  # noqa: E402


def split3(t):          # chunk-major planes of an NHWC tensor
    return _split3(L, t)


def ck(rc, what):
    hipabi.check(rc, what)


for name, H, Cin, Cout, k, stride in SHAPES:
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=dev).relu_()          # post-ReLU activations
    w = torch.randn(Cout, Cin, k, k, device=dev) * (2.0 / (Cin * k * k)) ** 0.5
    wp, wd = torch.empty_like(w), torch.empty_like(w)
    L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
    L.straps_pack_conv_weight_dgrad(hipabi.ptr(w), hipabi.ptr(wd), Cout, Cin, k, k, None)
    x3, xps = split3(x)
    wp3, wps = weight_planes(L, w)
    wd3, wdps = weight_planes(L, w, dgrad=True)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    y3 = torch.empty_like(y)
    flops = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    # ---- accuracy (forward), sampled bodies against float64
    nb = min(B, 4)
    ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double(), w.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ck(L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), None, B, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'fwd')
    scale = ref.abs().max().item()
    e32 = (y[:nb].double() - ref).abs().max().item() / scale
    errs = []
    for cfg in (1, 2, 3, 4):
        if cfg in (1, 4) and Cout % 128:
            continue
        y3.zero_()
        ck(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y3), None, B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), 'fwd_x3')
        errs.append((y3[:nb].double() - ref).abs().max().item() / scale)
    # ---- accuracy (data gradient)
    g = torch.randn(B, Ho, Ho, Cout, device=dev) * 1e-3
    g3, gps = split3(g)
    dx = torch.empty_like(x)
    dx3 = torch.empty_like(x)
    refd = torch.nn.grad.conv2d_input((nb, Cin, H, H), w.double(), g[:nb].permute(0, 3, 1, 2).double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ck(L.straps_conv_dgrad(hipabi.ptr(g), hipabi.ptr(wd), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'dgrad')
    ck(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx3), B, H, H, Cin, Cout, k, k, stride, pad, 0, None), 'dgrad_x3')
    sd = refd.abs().max().item()
    ed32 = (dx[:nb].double() - refd).abs().max().item() / sd
    ed3 = (dx3[:nb].double() - refd).abs().max().item() / sd
    # ---- time
    t32 = timeit(lambda: L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), None, B, H, H, Cin, Cout, k, k, stride, pad, 0, None))
    row = '%-13s M=%6d N=%3d K=%4d | fwd err/max: fp32 %.1e  x3 %s | dgrad err: fp32 %.1e x3 %.1e | fwd fp32 %6.1f us (%5.1f TF) | x3' % (
        name, B * Ho * Ho, Cout, Cin * k * k, e32, ' '.join('%.1e' % e for e in errs), ed32, ed3, t32 * 1e6, flops / t32 / 1e12)
    for cfg in (0, 2, 4, 5, 7, 8, 9, 10, 11, 12):
        if (cfg & 15) in (1, 4, 5, 6, 8, 9, 12) and Cout % 128:
            continue
        t = timeit(lambda: L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y3), None, B, H, H, Cin, Cout, k, k, stride, pad, cfg, None))
        row += ' c%d %5.1f' % (cfg, t * 1e6)
    td32 = timeit(lambda: L.straps_conv_dgrad(hipabi.ptr(g), hipabi.ptr(wd), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, 0, None))
    row += ' us | dgrad fp32 %6.1f us, x3' % (td32 * 1e6)
    for cfg in (0, 2, 4, 5, 7, 8, 9, 10, 11, 12):
        if (cfg & 15) in (1, 4, 5, 6, 8, 9, 12) and Cin % 128:
            continue
        t = timeit(lambda: L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx3), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None))
        row += ' c%d %5.1f' % (cfg, t * 1e6)
    out = torch.empty(3, xps, dtype=torch.int16, device=dev)
    ts = timeit(lambda: L.straps_split3_bf16_cm(hipabi.ptr(x), hipabi.ptr(out), x.numel() // Cin, Cin, xps, None))
    row += ' us | split(x) %5.1f us' % (ts * 1e6)
    print(row, flush=True)
