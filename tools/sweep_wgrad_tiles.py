#!/usr/bin/env python3
"""tools/sweep_wgrad_tiles.py -- channel block of the per-tap weight gradient on the planes (conv_wgrad_x3_kernel<BCO, BCI>): the square blocks
of round 3 against rectangular ones, on the resnet50 (32 bodies) and resnet18 (64 bodies) layers that run this kernel.  Every operand of the
kernel is re-read once per block of the OTHER channel dimension, so a 256 x 128 block streams 25 % fewer bytes than 128 x 128 on a 512 x 128
layer -- if bytes bound it.  Timed COLD: a 1 GiB fill between launches evicts the operands from L2 and the memory-side cache (back-to-back launches on
the same tensors are served from the 256 MB Infinity Cache and hide exactly the traffic this sweep is about), one HIP-event pair per launch.
Tools build (STRAPS_WGRAD_TILE is read per call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3 as _split3  # noqa: E402

L = hipabi.use_library(hipabi.build(tools=True))
dev = torch.device('cuda:0')
SHAPES = [('r50 l2 256>128', 32, 64, 256, 128, 1, 1), ('r50 l2 128>512', 32, 32, 128, 512, 1, 1), ('r50 l2 512>128', 32, 32, 512, 128, 1, 1),
          ('r50 l2 ds 256>512 s2', 32, 64, 256, 512, 1, 2), ('r50 l3 512>256', 32, 32, 512, 256, 1, 1), ('r50 l3 256>1024', 32, 16, 256, 1024, 1, 1),
          ('r50 l3 1024>256', 32, 16, 1024, 256, 1, 1), ('r50 l3 ds 512>1024 s2', 32, 32, 512, 1024, 1, 2), ('r50 l4 1024>512', 32, 16, 1024, 512, 1, 1),
          ('r50 l4 512>2048', 32, 8, 512, 2048, 1, 1), ('r50 l4 2048>512', 32, 8, 2048, 512, 1, 1), ('r50 l4 ds 1024>2048 s2', 32, 16, 1024, 2048, 1, 2),
          ('r50 l2 3x3 s2', 32, 64, 128, 128, 3, 2), ('r50 l3 3x3 s2', 32, 32, 256, 256, 3, 2), ('r50 l4 3x3 s2', 32, 16, 512, 512, 3, 2),
          ('r18 l2.0 3x3 s2', 64, 64, 64, 128, 3, 2), ('r18 l3.0 3x3 s2', 64, 32, 128, 256, 3, 2), ('r18 l4.0 3x3 s2', 64, 16, 256, 512, 3, 2),
          ('r18 l3 ds 1x1 s2', 64, 32, 128, 256, 1, 2), ('r18 l4 ds 1x1 s2', 64, 16, 256, 512, 1, 2)]
TILES = [(128, 128), (64, 64), (256, 64), (64, 256), (256, 128), (128, 256), (128, 64), (64, 128)]
flush = torch.empty(1 << 28, device=dev)          # 1 GiB


def cold(fn, iters=5):
    ts = []
    for _ in range(iters + 1):
        flush.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return min(ts[1:]), sum(ts[1:]) / iters


for name, Bn, H, Cin, Cout, k, stride in SHAPES:
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    torch.manual_seed(0)
    x = torch.randn(Bn, H, H, Cin, device=dev).relu_()
    g = torch.randn(Bn, Ho, Ho, Cout, device=dev) * 1e-3
    x3, xps = _split3(L, x)
    g3, gps = _split3(L, g)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(Bn, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    ref = None
    row = []
    for bco, bci in TILES:
        if Cout % bco or Cin % bci or ((bco, bci) == (128, 128) and (Cin % 128 or Cout % 128 or k > 1)):
            continue
        os.environ['STRAPS_WGRAD_TILE'] = str(bco * 1000 + bci)
        if os.environ.get('SWEEP_VERBOSE'):
            print('  ..', name, bco, bci, flush=True)
        dw = torch.full((Cout, Cin, k, k), float('nan'), device=dev)

        def fn():
            hipabi.check(L.straps_conv_wgrad_x3(None, None, hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw), hipabi.ptr(ws), Bn, H, H, Cin, Cout, k, k,
                                                stride, pad, 0, None), 'wgrad_x3')
        tmin, tavg = cold(fn)
        if ref is None:
            ref = dw.clone()
            ok = ''
        else:
            err = float((dw - ref).abs().max() / ref.abs().max())
            ok = '' if err < 2e-6 else ' MISMATCH %.1e' % err
        row.append('%dx%d %.1f/%.1f%s' % (bco, bci, tmin, tavg, ok))
    del os.environ['STRAPS_WGRAD_TILE']
    print('%-24s M=%6d  cold us (min/avg): %s' % (name, Bn * Ho * Ho, '  '.join(row)), flush=True)
