#!/usr/bin/env python3
"""tools/lds_guard_probe.py -- which kernel of the library writes LDS it does not own?

Round 4: the part rasteriser (1 KB of LDS per workgroup: its pixel-centre table) gives a handful of different pixels about once in 200 launches
when -- and only when -- bf16x3 convolution kernels run on another stream (tools/datagen_determinism_probe.py).  tools/lds_guard.hip parks
small workgroups with a known LDS pattern on the chip; this script runs them beside one library kernel at a time (captured in a hipGraph and
replayed on the main stream so that the chip stays full) and prints what, if anything, was written into the guards' LDS."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3, weight_planes  # noqa: E402

L = hipabi.load()
G = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'liblds_guard.so'))
G.lds_guard_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda:0')
guard_stream = torch.cuda.Stream()
LDS = int(os.environ.get('GUARD_LDS', '1024'))


def conv_case(B, H, Cin, Cout, k, stride, cfg, what):
    pad = 1 if k == 3 else 0
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, Cin, device=dev).relu_()
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    x3, xps = split3(L, x)
    w3, wps = weight_planes(L, w)
    wd3, wdps = weight_planes(L, w, dgrad=True)
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    g = torch.randn(B, Ho, Ho, Cout, device=dev) * 1e-3
    g3, gps = split3(L, g)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    nblk = L.straps_conv_x3_stat_blocks(B, H, H, Cin, Cout, k, k, stride, pad, cfg)
    part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    if what == 'fwd':
        return lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout,
                                                         k, k, stride, pad, cfg, hipabi.stream_ptr()), 'fwd')
    if what == 'dgrad':
        return lambda: hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, cfg,
                                                           hipabi.stream_ptr()), 'dgrad')
    return lambda: hipabi.check(L.straps_conv_wgrad_x3(None, None, hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw), hipabi.ptr(ws), B, H, H, Cin, Cout, k, k, stride,
                                                       pad, 0, hipabi.stream_ptr()), 'wgrad')


CASES = [('fwd   3x3 256->256 32x32 auto (halo 128x128)', (32, 32, 256, 256, 3, 1, 0, 'fwd')),
         ('fwd   3x3 64->64 64x64 auto (halo 128x64, one patch buffer)', (16, 64, 64, 64, 3, 1, 0, 'fwd')),
         ('fwd   3x3 256->256 im2col 256x128 (cfg 256)', (32, 32, 256, 256, 3, 1, 256, 'fwd')),
         ('fwd   1x1 256->1024 16x16 auto', (32, 16, 256, 1024, 1, 1, 0, 'fwd')),
         ('fwd   1x1 64->256 64x64 auto', (8, 64, 64, 256, 1, 1, 0, 'fwd')),
         ('fwd   3x3 s2 128->128 64x64 auto', (8, 64, 128, 128, 3, 2, 0, 'fwd')),
         ('fwd   3x3 512->512 8x8 auto (64x64 tiles)', (4, 8, 512, 512, 3, 1, 0, 'fwd')),
         ('dgrad 3x3 256->256 32x32 auto', (32, 32, 256, 256, 3, 1, 0, 'dgrad')),
         ('dgrad 3x3 s2 128->128 64x64 auto (parity classes)', (8, 64, 128, 128, 3, 2, 0, 'dgrad')),
         ('wgrad 3x3 256->256 32x32 (halo-patch kernel)', (32, 32, 256, 256, 3, 1, 0, 'wgrad')),
         ('wgrad 1x1 256->1024 16x16 (per-tap kernel)', (32, 16, 256, 1024, 1, 1, 0, 'wgrad')),
         ('wgrad 3x3 s2 128->128 64x64 (per-tap kernel)', (8, 64, 128, 128, 3, 2, 0, 'wgrad'))]
only = os.environ.get('GUARD_ONLY')
for name, args in CASES:
    if only and only not in name:
        continue
    fn = conv_case(*args)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            fn()
    report = torch.zeros(1 + 4 * 64, device=dev, dtype=torch.int32)
    for _ in range(60):
        g.replay()
        with torch.cuda.stream(guard_stream):
            G.lds_guard_launch(report.data_ptr(), 4096, LDS, 40, 64, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r = report.cpu().tolist()
    n = r[0]
    line = '%-62s guards with %d B of LDS: %d words overwritten' % (name, LDS, n)
    if n:
        line += '; first: ' + '  '.join('(wg %d, byte %d, value 0x%08x, round %d)' % (r[1 + 4 * k], r[2 + 4 * k], r[3 + 4 * k] & 0xffffffff, r[4 + 4 * k]) for k in range(min(n, 4)))
    print(line, flush=True)
