"""Sanity run: N training steps at the bench configuration; prints the loss trajectory (finite, decreasing)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd.train_step import TrainStep
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = 'cuda:0'
mp = straps_amd.synthetic_mean_params(0)
torch.manual_seed(0)
reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
    init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=mp['shape'], use_graph=True, track_metrics=True)
rec = []
for i in range(steps):
    l = ts.step()
    if i % max(1, steps // 10) == 0 or i == steps - 1:
        rec.append((i, float(l[0])))
print('loss:', ' '.join('%d:%.4f' % r for r in rec))
print('finite params:', bool(torch.isfinite(ts.flat_p).all()), 'graph:', ts.graph is not None)
print('metrics:', {k: round(v, 4) for k, v in ts.metrics_summary().items()})
