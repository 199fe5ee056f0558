"""Experiment: which part of the training step is not bit-reproducible?  Runs forward_backward twice on the same batch
and compares the gradient slices per parameter; also probes straps_smpl_bwd alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd.train_step import TrainStep
dev = 'cuda:0'
MP = straps_amd.synthetic_mean_params(0)
torch.manual_seed(7)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=64).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
ts = TrainStep(reg, smpl, crit, 64, seed=99, mean_shape=MP['shape'], use_graph=False, overlap_wgrad=False)
with torch.no_grad():
    batch = ts.make_batch()
    outs = []
    for _ in range(3):
        ts.forward_backward(batch)
        torch.cuda.synchronize()
        outs.append(ts.flat_g.clone())
names = [n for n, _ in reg.named_parameters()] + [n for n, _ in crit.named_parameters()]
off = 0
bad = []
for n, p in zip(names, ts.params):
    k = p.numel()
    if not (torch.equal(outs[0][off:off + k], outs[1][off:off + k]) and torch.equal(outs[0][off:off + k], outs[2][off:off + k])):
        bad.append((n, float((outs[0][off:off + k] - outs[1][off:off + k]).abs().max())))
    off += k
print('non-reproducible gradient slices:', len(bad), bad[:8], '...', bad[-3:])
# batch generation
b2 = None
for s in range(2):
    with torch.cuda.device(dev):
        torch.cuda.manual_seed(123)
    with torch.no_grad():
        b = ts.make_batch()
    if b2 is not None:
        print('make_batch reproducible:', {k: bool(torch.equal(b[k], b2[k])) for k in b if torch.is_tensor(b[k])})
    b2 = b
