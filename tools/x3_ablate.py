#!/usr/bin/env python3
"""tools/x3_ablate.py -- where the time of the bf16x3 implicit GEMM goes, per resnet18 3x3 / stride-1 layer shape (B = 64): the im2col
kernel with its auto-rule tile and three ablation builds (tile_cfg bit 6 = operand copies only, bit 7 = no operand copies, both = the
MFMA stream + barriers alone; wrong results by construction), plus the halo-patch kernels where they apply.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3, weight_planes  # noqa: E402

L = hipabi.use_library(hipabi.build(tools=True))      # the -DSTRAPS_TOOLS build: ablation instantiations + STRAPS_* A/B switches
dev = torch.device('cuda:0')
SHAPES = [('l1', 64, 64, 64, 11), ('l2', 32, 128, 128, 12), ('l3', 16, 256, 256, 5), ('l4', 8, 512, 512, 7)]
B = 64


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, H, Cin, Cout, cfg in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, H, H, Cin, device=dev).relu_()
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * (2.0 / (Cin * 9)) ** 0.5
    x3, xps = split3(L, x)
    w3, wps = weight_planes(L, w)
    y = torch.empty(B, H, H, Cout, device=dev)
    nblk = max(L.straps_conv_x3_stat_blocks(B, H, H, Cin, Cout, 3, 3, 1, 1, c) for c in (0, cfg, 512))
    part = torch.empty(nblk, Cout, 2, device=dev)

    def run(c, stats=False):
        return timeit(lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y),
                                                                hipabi.ptr(part if stats else None), B, H, H, Cin, Cout, 3, 3, 1, 1, c, None), 'conv'))
    row = '%s (cfg %2d): full %6.1f  +stats %6.1f | copies only %6.1f | no copies %6.1f | mfma+barriers %6.1f | auto %6.1f' % (
        name, cfg, run(cfg), run(cfg, True), run(cfg + 64), run(cfg + 128), run(cfg + 192), run(0))
    if Cout <= 256:
        row += ' | halo(512) %6.1f' % run(512)
    print(row + ' us', flush=True)
