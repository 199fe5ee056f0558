#!/usr/bin/env python3
"""tools/x3_one.py NAME CFG [ITERS] -- launches the bf16x3 forward convolution of one resnet18 layer shape (B=64) ITERS times with
one tile configuration (for rocprofv3 counter passes: tools/pmc_x3.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

SHAPES = {'l1': (64, 64, 64, 3, 1), 'l2': (32, 128, 128, 3, 1), 'l3': (16, 256, 256, 3, 1), 'l4': (8, 512, 512, 3, 1)}
name, cfg = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
H, Cin, Cout, k, stride = SHAPES[name]
B, pad = 64, 1
L = hipabi.load()
dev = torch.device('cuda:0')
x = torch.randn(B, H, H, Cin, device=dev).relu_()
w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
from straps_amd.encoder_exec import split3, weight_planes  # noqa: E402

x3, xps = split3(L, x)                  # chunk-major planes
w3, wps = weight_planes(L, w)
y = torch.empty(B, H, H, Cout, device=dev)
for _ in range(iters):
    hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y), None, B, H, H, Cin, Cout, k, k,
                                      stride, pad, cfg, None), 'fwd_x3')
torch.cuda.synchronize()
