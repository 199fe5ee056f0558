"""Experiment: sparsity statistics of the training step's proxy input (how much the stem's zero skipping can skip)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd
from straps_amd.train_step import TrainStep
dev = torch.device('cuda:0')
mp = straps_amd.synthetic_mean_params(0)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=64).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
ts = TrainStep(reg, smpl, crit, 64, mean_shape=mp['shape'])
with torch.no_grad():
    x = ts.make_batch()['input']
nz = (x != 0).float()
print('non-zero fraction per channel:', [round(float(v), 4) for v in nz.mean(dim=(0, 2, 3))])
# wgrad tiles: 9 x 72 input patch at stride (4, 64), origin (-3, -4)
p = F.pad(nz, (4, 4, 3, 5))
act = F.max_pool2d(p, kernel_size=(9, 72), stride=(4, 64))          # [B, C, 64, 4]
print('wgrad tile activity per channel:', [round(float(v), 3) for v in act.mean(dim=(0, 2, 3))])
grp = torch.stack([act[:, g * 4:(g + 1) * 4].amax(1) for g in range(5)], 1)
print('wgrad (tile, group) activity per group:', [round(float(v), 3) for v in grp.mean(dim=(0, 2, 3))], 'mean', float(grp.mean()))
pf = F.pad(nz, (4, 4, 3, 9))
actf = F.max_pool2d(pf, kernel_size=(13, 72), stride=(8, 64))
print('fwd tile any-channel activity:', float(actf.amax(1).mean()), 'per-channel mean', float(actf.mean()))
