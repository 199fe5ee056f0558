"""Experiment: time straps_smpl_fwd from alternative builds (build_dbg/lib_*.so, compiled with debug -D flags)
over the `chunks` parameter, to attribute the vertex kernel's time.  Not part of the product or the tests."""
import ctypes as C, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
B = 65536
dev = torch.device('cuda:0')
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=1).to(dev)
struct = smpl._model_struct()
g = torch.Generator().manual_seed(0)
betas = torch.randn(B, 10, generator=g).to(dev)
R = torch.linalg.qr(torch.randn(B * 24, 3, 3, generator=g))[0].reshape(B, 24, 3, 3).contiguous().to(dev)
verts = torch.empty(B, 6890, 3, device=dev); joints = torch.empty(B, 90, 3, device=dev)
for path in sorted(glob.glob('build_dbg/lib_*.so')):
    lib = hipabi.load(path)
    for chunks in (0, 32, 8, 4, 1):
        ws = torch.empty(lib.straps_smpl_workspace_bytes(C.byref(struct), B) // 4, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def run():
            rc = lib.straps_smpl_fwd(C.byref(struct), hipabi.ptr(betas), hipabi.ptr(R), hipabi.ptr(verts), hipabi.ptr(joints), hipabi.ptr(ws), B, chunks, st)
            assert rc == 0, lib.straps_last_error()
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(os.path.basename(path), 'chunks', chunks, '%.3f ms' % (e0.elapsed_time(e1) / 5), flush=True)
