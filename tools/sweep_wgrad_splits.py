"""Experiment: generic conv weight-gradient kernel vs the split-count target (build_dbg/lib_sp<N>.so)."""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
dev = torch.device('cuda:0')
B = int(os.environ.get("SWEEP_B", "32"))
SHAPES = [('l1 64>256', 64, 64, 256, 1, 1), ('l2 256>128', 64, 256, 128, 1, 1), ('l2 128>512', 32, 128, 512, 1, 1), ('l3 512>256', 32, 512, 256, 1, 1),
          ('l3 256>1024', 16, 256, 1024, 1, 1), ('l4 1024>512', 16, 1024, 512, 1, 1), ('l4 512>2048', 8, 512, 2048, 1, 1), ('l3 3x3 s2', 32, 256, 256, 3, 2),
          ('r18 l2.0 s2', 64, 64, 128, 3, 2), ('r18 l3.0 s2', 32, 128, 256, 3, 2), ('r18 l4.0 s2', 16, 256, 512, 3, 2), ('r18 l3.0 ds', 32, 128, 256, 1, 2)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(ROOT, 'build_dbg', 'lib_sp*.so')))
for path in paths + paths[:1]:
    L = hipabi.load(path)
    row = os.path.basename(path) + ':'
    for name, H, Cin, Cout, k, stride in SHAPES:
        pad = 1 if k == 3 else 0
        Ho = (H + 2 * pad - k) // stride + 1
        x = torch.randn(B, H, H, Cin, device=dev); y = torch.randn(B, Ho, Ho, Cout, device=dev); dw = torch.empty(Cout, Cin, k, k, device=dev)
        ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, k, k, stride, pad) // 4, device=dev)
        def run():
            assert L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(y), hipabi.ptr(dw), hipabi.ptr(ws), B, H, H, Cin, Cout, k, k, stride, pad, 0, None) == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        row += ' %s %.0fus(%.0fTF)' % (name, t * 1e6, 2.0 * B * Ho * Ho * Cout * Cin * k * k / t / 1e12)
    print(row, flush=True)
