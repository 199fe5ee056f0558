#!/usr/bin/env python3
"""tools/wgrad3_ablate.py -- where does the time of the 3x3 weight gradient on the planes (conv_wgrad3x3_x3_kernel<2, 64>) go?  Times the
kernel (without its split reduce) on the resnet18 layer shapes at B = 64; run once per STRAPS_WGRAD3_ABL value (the switch is read once
per process): 0 = product kernel (+ reduce), 1 = no operand copies after the first chunk, 2 = no MFMAs, 3 = neither (fragment reads +
barriers only), 4 = a tenth of the fragment reads (MFMAs + copies), 5 = MFMAs + barriers, a tenth of the reads, no copies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3  # noqa: E402

L = hipabi.use_library(hipabi.build(tools=True))      # the -DSTRAPS_TOOLS build: ablation instantiations + STRAPS_* A/B switches
dev = torch.device('cuda:0')
B = 64
row = 'ABL=%s' % os.environ.get('STRAPS_WGRAD3_ABL', '0')
for name, H, C in (('l1', 64, 64), ('l2', 32, 128), ('l3', 16, 256), ('l4', 8, 512)):
    torch.manual_seed(0)
    x = torch.randn(B, H, H, C, device=dev).relu_()
    g = torch.randn(B, H, H, C, device=dev) * 1e-3
    x3, xps = split3(L, x)
    g3, gps = split3(L, g)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, C, C, 3, 3, 1, 1) // 4, device=dev)
    dw = torch.empty(C, C, 3, 3, device=dev)

    def fn():
        hipabi.check(L.straps_conv_wgrad_x3(None, None, hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw), hipabi.ptr(ws), B, H, H, C, C, 3, 3, 1, 1, 0, None), 'wgrad')
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    row += ' | %s %6.1f us' % (name, s.elapsed_time(e) / 20 * 1e3)
print(row, flush=True)
