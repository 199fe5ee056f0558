"""Experiment: straps_stem_fwd (training mode) on the training step's own proxy batch, across builds in build_dbg/lib_*.so."""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd import hipabi
from straps_amd.train_step import TrainStep
dev = torch.device('cuda:0')
mp = straps_amd.synthetic_mean_params(0)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=64).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
ts = TrainStep(reg, smpl, crit, 64, mean_shape=mp['shape'])
with torch.no_grad():
    x = ts.make_batch()['input'].contiguous()
B, C, H, W = x.shape
net = reg.image_encoder
wfrag = net._packed_weight(net.conv1, stem=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(ROOT, 'build_dbg', 'lib_*.so'))) or [hipabi.LIB_PATH]
wa = torch.randn(8192, 8192, device=dev)
for _ in range(40): wa @ wa
torch.cuda.synchronize()
for path in paths + paths[:1]:
    L = hipabi.load(path)
    nz = torch.empty(L.straps_stem_nzmask_words(B, C, H, W), device=dev, dtype=torch.int32)
    L.straps_stem_nzmask(hipabi.ptr(x), hipabi.ptr(nz), B, C, H, W, None)
    y = torch.empty(B, 128, 128, 64, device=dev)
    part = torch.empty(L.straps_stem_stat_blocks(B, H, W), 64, 2, device=dev)
    row = os.path.basename(path) + ':'
    for tag, mask in (('sparse', nz), ('dense', torch.full_like(nz, -1)), ('empty', torch.zeros_like(nz))):
        fn = lambda: L.straps_stem_fwd(hipabi.ptr(x), hipabi.ptr(wfrag), None, None, 0, hipabi.ptr(y), hipabi.ptr(part), hipabi.ptr(mask), B, C, H, W, None)
        assert fn() == 0, L.straps_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        row += ' %s %.0f us' % (tag, e0.elapsed_time(e1) / 20 * 1e3)
    print(row, flush=True)
