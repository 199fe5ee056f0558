#!/usr/bin/env python3
"""tools/igemm_trace.py -- where does a wave of the implicit-GEMM kernel spend its cycles?  Runs resnet18 layer shapes (B=64) through the
traced build (tile_cfg bit 5, straps_conv_trace_buffer) and prints, per shape and tile, the mean shader-clock cycles per chunk that wave 0
of a workgroup spends (a) waiting for its own operand copies, (b) in the barrier, (c) issuing the next copies, (d) in fragment reads + the
MFMA burst, plus prologue / epilogue / total per workgroup."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd import hipabi  # noqa: E402

L = hipabi.load()
dev = torch.device('cuda:0')
B = 64
SHAPES = [('l1 3x3 s1', 64, 64, 64, 3, 1), ('l2 3x3 s1', 32, 128, 128, 3, 1), ('l3 3x3 s1', 16, 256, 256, 3, 1), ('l4 3x3 s1', 8, 512, 512, 3, 1),
          ('l2.0 3x3 s2', 64, 64, 128, 3, 2)]
for name, H, Cin, Cout, k, stride in SHAPES:
    pad = 1
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wp = torch.empty_like(w)
    L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    for cfg, bm, bn in ((3, 64, 64), (2, 128, 64), (4, 256, 64), (1, 128, 128)):
        if Cout % bn:
            continue
        nwg = ((B * Ho * Ho + bm - 1) // bm) * (Cout // bn)
        trace = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
        hipabi.check(L.straps_conv_trace_buffer(hipabi.ptr(trace)), 'trace buffer')
        part = torch.empty(L.straps_conv_stat_blocks(B, Ho, Ho, Cout, k * k * Cin, cfg), Cout, 2, device=dev)

        def run(c):
            return L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k, stride, pad, c, None)
        for _ in range(3):
            run(cfg | 32)
        # ablation: trace pointer with its low bit set = two of three A copies skipped (timing only)
        hipabi.check(L.straps_conv_trace_buffer(C.c_void_p(trace.data_ptr() | 1)), 'trace buffer')
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run(cfg | 32)
        e.record()
        torch.cuda.synchronize()
        us_abl = s.elapsed_time(e) / 5 * 1e3
        hipabi.check(L.straps_conv_trace_buffer(hipabi.ptr(trace)), 'trace buffer')
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run(cfg)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 5 * 1e3
        s.record()
        for _ in range(5):
            run(cfg | 32)
        e.record()
        torch.cuda.synchronize()
        us_t = s.elapsed_time(e) / 5 * 1e3
        t = trace.double().cpu()
        ch = t[:, 4].clamp_min(1)
        seg = (t[:, :4] / ch[:, None]).mean(0)
        tf = 2.0 * B * Ho * Ho * Cout * Cin * k * k / us / 1e6
        print('%-12s %3dx%-3d %6.1f us (%5.1f TFLOP/s; traced %6.1f us; A copies /3: %6.1f us) | per chunk: copy-wait %6.0f  barrier %6.0f  copy-issue %5.0f  mfma %6.0f cyc'
              ' | per WG: prologue %6.0f  epilogue %6.0f  total %8.0f cyc, %d chunks, %d WGs'
              % (name, bm, bn, us, tf, us_t, us_abl, seg[0], seg[1], seg[2], seg[3], t[:, 6].mean(), t[:, 7].mean(), t[:, 5].mean(), int(ch.mean()), nwg), flush=True)
hipabi.check(L.straps_conv_trace_buffer(None), 'trace buffer off')
