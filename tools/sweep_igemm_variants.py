"""Experiment: implicit-GEMM forward of the resnet18 3x3 layers for alternative builds in build_dbg/lib_*.so."""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
dev = torch.device('cuda:0')
B = 64
SHAPES = [('l1', 64, 64, 64, 3, 1), ('l2', 32, 128, 128, 3, 1), ('l3', 16, 256, 256, 3, 1), ('l4', 8, 512, 512, 3, 1), ('l2.0s2', 64, 64, 128, 3, 2)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(ROOT, 'build_dbg', 'lib_*.so')))
# clocks ramp over the first second of load: warm up, and measure the first library again at the end
wa = torch.randn(8192, 8192, device=dev)
for _ in range(40): wa @ wa
torch.cuda.synchronize()
for path in paths + paths[:1]:
    L = hipabi.load(path)
    row = os.path.basename(path) + ':'
    for name, H, Cin, Cout, k, stride in SHAPES:
        pad = 1
        Ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        wp, wd = torch.empty_like(w), torch.empty_like(w)
        L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), Cout, Cin, k, k, None)
        L.straps_pack_conv_weight_dgrad(hipabi.ptr(w), hipabi.ptr(wd), Cout, Cin, k, k, None)
        y = torch.empty(B, Ho, Ho, Cout, device=dev); dx = torch.empty_like(x)
        part = torch.empty(L.straps_conv_stat_blocks(B, Ho, Ho, Cout, k * k * Cin, 0), Cout, 2, device=dev)
        fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
        for tag, fn in (('f', lambda: L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout, k, k, stride, pad, 0, None)),
                        ('d', lambda: L.straps_conv_dgrad(hipabi.ptr(y), hipabi.ptr(wd), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad, 0, None))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            row += ' %s%s %.1f' % (name, tag, fl / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    print(row, flush=True)
