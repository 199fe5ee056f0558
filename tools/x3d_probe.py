"""Direct-A tiles of the bf16x3 implicit GEMM (tile_cfg 13 = 128x64, 14 = 256x64; csrc/conv_x3.hip conv_igemm_x3d_kernel):
result against the auto-rule kernel on a few shapes (max |difference| relative to the largest output; the summation order differs, so
~1e-7 is agreement) and time per launch on the layer1 shape (B = 64, 64 -> 64 channels, 64 x 64, 3x3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402

dev = torch.device('cuda:0')
L = hipabi.load()


def split(t):
    n = t.numel()
    ps = (n + 7) // 8 * 8
    pl = torch.zeros(3, ps, device=dev, dtype=torch.int16)
    hipabi.check(L.straps_split3_bf16(hipabi.ptr(t), hipabi.ptr(pl), n, ps, None), 'split')
    return pl, ps


def run(B, Cin, Cout, H, W, k, stride, cfgs, time_it=False):
    pad = 1 if k == 3 else 0
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dev)
    wp = torch.empty_like(w)
    hipabi.check(L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None), 'pack')
    x3, xps = split(x)
    w3, wps = split(wp)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    outs = {}
    for cfg in cfgs:
        y = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
        nblk = L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, k, k, stride, pad, cfg)
        part = torch.empty(nblk, Cout, 2, device=dev)
        fn = lambda: hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part),
                                                       B, H, W, Cin, Cout, k, k, stride, pad, cfg, None), 'conv')
        fn()
        torch.cuda.synchronize()
        outs[cfg] = (y.clone(), part.double().sum(0))
        if time_it:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                fn()
            torch.cuda.synchronize()
            print('  cfg %2d: %.1f us' % (cfg, (time.perf_counter() - t0) / 30 * 1e6), flush=True)
    y0, s0 = outs[cfgs[0]]
    for cfg in cfgs[1:]:
        y, s = outs[cfg]
        print('(%d,%d,%d,%dx%d,k%d,s%d) cfg %d vs cfg %d: max |dy| / max|y| = %.2e, nan %d, stats rel %.1e' % (
            B, Cin, Cout, H, W, k, stride, cfg, cfgs[0], float((y - y0).abs().max() / y0.abs().max()), int(torch.isnan(y).sum()),
            float((s - s0).abs().max() / s0.abs().max())), flush=True)


if os.environ.get('PROBE', 'halo1') == 'halo1':
    # single-patch-buffer halo kernel (tile_cfg bit 10) against the auto rule: 8 / 4 / 2 rows per tile, whole images, 4 channel chunks (three reloads)
    run(2, 64, 64, 16, 16, 3, 1, (0, 1024))
    run(1, 64, 64, 8, 32, 3, 1, (0, 1024))
    run(4, 64, 64, 8, 8, 3, 1, (0, 1024))
    run(2, 128, 64, 16, 16, 3, 1, (0, 1024))
    run(1, 256, 64, 64, 64, 3, 1, (0, 1024))
    run(64, 64, 64, 64, 64, 3, 1, (0, 2, 512, 1024, 11), time_it=True)
    sys.exit(0)
run(2, 64, 64, 16, 16, 3, 1, (0, 13, 14))
run(5, 64, 64, 7, 7, 3, 1, (0, 13, 14))
run(3, 96, 128, 7, 13, 3, 2, (0, 13))
run(2, 256, 64, 16, 16, 1, 1, (0, 13, 14))
run(1, 32, 64, 8, 8, 1, 1, (0, 13))
run(64, 64, 64, 64, 64, 3, 1, (0, 2, 13, 14), time_it=True)
