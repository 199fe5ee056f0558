"""A/B of the 3x3 weight-gradient kernel across builds in build_dbg/lib_*.so: resnet18 3x3/s1 layers at B=64; the first library
is the reference for a bit-exact comparison of dW."""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
dev = torch.device('cuda:0')
B = 64
SHAPES = [('l1', 64, 64, 64), ('l2', 32, 128, 128), ('l3', 16, 256, 256), ('l4', 8, 512, 512)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(ROOT, 'build_dbg', 'lib_*.so')))
wa = torch.randn(8192, 8192, device=dev)
for _ in range(40): wa @ wa
torch.cuda.synchronize()
ref = {}
for path in paths + paths:
    L = hipabi.load(path)
    row = os.path.basename(path) + ':'
    for name, H, Cin, Cout in SHAPES:
        torch.manual_seed(1)
        x = torch.randn(B, H, H, Cin, device=dev); dy = torch.randn(B, H, H, Cout, device=dev)
        dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
        ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, H, Cin, Cout, 3, 3, 1, 1) // 4, device=dev)
        fn = lambda: L.straps_conv_wgrad(hipabi.ptr(x), hipabi.ptr(dy), hipabi.ptr(dw), hipabi.ptr(ws), B, H, H, Cin, Cout, 3, 3, 1, 1, 0, None)
        assert fn() == 0, L.straps_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): fn()
        e1.record(); torch.cuda.synchronize()
        ok = '' if torch.equal(ref.setdefault(name, dw.clone()), dw) else ' !MISMATCH(%.2e)' % float((ref[name] - dw).abs().max())
        row += ' %s %.1f TF (%.0f us)%s' % (name, 2.0 * B * H * H * Cout * Cin * 9 / (e0.elapsed_time(e1) / 30 * 1e-3) / 1e12, e0.elapsed_time(e1) / 30 * 1e3, ok)
    print(row, flush=True)
