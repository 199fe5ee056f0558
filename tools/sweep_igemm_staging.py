"""A/B: operand staging of the implicit GEMM -- LDS-DMA (default, ns2) vs registers (tile_cfg bit 4, ns0) on the resnet18
layer shapes at B=64, both tile sizes, forward and data gradient; checks that the two are bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
dev = torch.device('cuda:0')
B = 64
SHAPES = [('l1', 64, 64, 64, 3, 1), ('l2', 32, 128, 128, 3, 1), ('l3', 16, 256, 256, 3, 1), ('l4', 8, 512, 512, 3, 1), ('l2.0s2', 64, 64, 128, 3, 2),
          ('l3.0ds', 32, 128, 256, 1, 2)]
L = hipabi.load()
wa = torch.randn(8192, 8192, device=dev)
for _ in range(40): wa @ wa
torch.cuda.synchronize()
CFGS = [(3, 0), (3, 2), (2, 2), (1, 0), (1, 2), (3, 0), (3, 2), (2, 2), (1, 0), (1, 2)]
for name, H, Cin, Cout, k, stride in SHAPES:
    pad = k // 2
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    wp, wd = torch.empty_like(w), torch.empty_like(w)
    L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
    L.straps_pack_conv_weight_dgrad(hipabi.ptr(w), hipabi.ptr(wd), Cout, Cin, k, k, None)
    dy = torch.randn(B, Ho, Ho, Cout, device=dev)
    fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    ref = {}
    row = '%-7s' % name
    for t, ns in CFGS:
        if t == 1 and Cout % 128: continue
        cfg = t + (16 if ns == 0 else 0)
        y = torch.zeros(B, Ho, Ho, Cout, device=dev); dx = torch.zeros_like(x)
        part = torch.zeros(L.straps_conv_stat_blocks(B, Ho, Ho, Cout, k * k * Cin, cfg), Cout, 2, device=dev)
        fns = [('f', lambda: L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), lambda: (y, part))]
        if not (t == 1 and Cin % 128):
            fns.append(('d', lambda: L.straps_conv_dgrad(hipabi.ptr(dy), hipabi.ptr(wd), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), lambda: (dx,)))
            if os.environ.get('SWEEP_ADDEND'):
                add = torch.randn_like(x)
                fns.append(('da', lambda: L.straps_conv_dgrad(hipabi.ptr(dy), hipabi.ptr(wd), hipabi.ptr(add), hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), lambda: (dx,)))
        for tag, fn, outs in fns:
            assert fn() == 0, L.straps_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            ok = ''
            key = (t, tag)
            if ns == 0:
                ref.setdefault(key, [o.clone() for o in outs()])
            else:
                ok = '' if key not in ref or all(torch.equal(a, b) for a, b in zip(ref[key], outs())) else '!MISMATCH'
            row += ' | t%d ns%d %s %.1f%s' % (t, ns, tag, fl / (e0.elapsed_time(e1) / 50 * 1e-3) / 1e12, ok)
    print(row, flush=True)
