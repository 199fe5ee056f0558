"""Experiment: time straps_stem_wgrad from alternative builds (build_dbg/lib_nb*.so) on a proxy-like sparse input
(silhouette + 17 heat-maps) and on a dense one.  Not part of the product or the tests."""
import ctypes as C, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
dev = torch.device('cuda:0')
B = 64
L0 = hipabi.load()
g = torch.Generator(device='cpu').manual_seed(0)
j2d = (torch.rand(B, 17, 2, generator=g) * 160 + 48).to(dev)
seg = torch.zeros(B, 256, 256, device=dev)
seg[:, 40:220, 90:170] = 1.0
x = torch.empty(B, 18, 256, 256, device=dev)
hipabi.check(L0.straps_build_proxy_input(hipabi.ptr(seg), hipabi.ptr(j2d), hipabi.ptr(x), B, 17, 256, None), "proxy")
xd = torch.rand(B, 18, 256, 256, device=dev)
dy = torch.randn(B, 128, 128, 64, device=dev)
dw = torch.empty(64, 18, 7, 7, device=dev)
print('non-zero fraction of the proxy input: %.3f' % float((x != 0).float().mean()))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in sorted(glob.glob(os.path.join(ROOT, 'build_dbg', 'lib_*.so'))):
    lib = hipabi.load(path)
    ws = torch.empty(lib.straps_stem_wgrad_workspace_bytes(B, 18, 256, 256) // 4, device=dev)
    for name, inp in [c for c in (('proxy', x), ('dense', xd)) if os.environ.get('STEM_SWEEP', c[0]) == c[0]]:
        def run():
            assert lib.straps_stem_wgrad(hipabi.ptr(inp), hipabi.ptr(dy), hipabi.ptr(dw), hipabi.ptr(ws), None, B, 18, 256, 256, 0, None) == 0
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(os.path.basename(path), name, '%.3f ms' % (e0.elapsed_time(e1) / 10), flush=True)
