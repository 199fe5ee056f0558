#!/usr/bin/env python3
"""tools/grad_conditioning.py [layers] [B] -- how well conditioned is the end-to-end gradient of one training step?
Per parameter tensor: relative error (L2 norm) of the GPU step's gradient and of the float32 CPU oracle's gradient, both
against the float64 CPU oracle on the SAME batch.  (Train-mode BatchNorm at small batch makes the problem ill-conditioned:
the reference's own fp32 arithmetic is that far from fp64.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import straps_amd  # noqa: E402
import straps_oracle as O  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.set_num_threads(16)
dev = torch.device('cuda:0')
mp = straps_amd.synthetic_mean_params(0)
W = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
torch.manual_seed(5)
reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'], init_loss_weights=W).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=mp['shape'])
with torch.no_grad():
    batch = ts.make_batch()
sd = {k: v.detach().cpu().clone() for k, v in reg.state_dict().items()}
cpu_batch = {k: batch[k].cpu() for k in ('input', 'verts', 'joints2d', 'joints3d', 'shape', 'rot')}
lv = {n: float(getattr(crit, n + '_log_var')) for n in O.LOSS_TASKS}
init = O.ief_init_estimate(mp['pose'], mp['shape'])
model = straps_amd.synthetic_smpl_model(0)
t64, _, g64, _ = O.train_step_loss_and_grads(cpu_batch, sd, init, model, layers, 3, lv, dtype=torch.float64)
t32, _, g32, _ = O.train_step_loss_and_grads(cpu_batch, sd, init, model, layers, 3, lv, dtype=torch.float32)
with torch.no_grad():
    loss = ts.forward_backward(batch)
torch.cuda.synchronize()
print('loss: gpu %.9g  cpu32 %.9g  cpu64 %.9g' % (float(loss[0]), float(t32), float(t64)))
rows = []
for n, p in reg.named_parameters():
    g = ts.gviews[p].detach().cpu().double().reshape(-1)
    r = g64[n].reshape(-1)
    eg = float((g - r).norm() / r.norm().clamp_min(1e-30))
    e32 = float((g32[n].double().reshape(-1) - r).norm() / r.norm().clamp_min(1e-30))
    rows.append((n, eg, e32, float(r.norm())))
for n, eg, e32, nr in rows:
    print('%-52s gpu %.2e  cpu32 %.2e  |g| %.3e%s' % (n, eg, e32, nr, '   <-- gpu > 3x cpu32' if eg > 3 * e32 + 1e-5 else ''))
print('worst gpu %.2e, worst cpu32 %.2e' % (max(r[1] for r in rows), max(r[2] for r in rows)))
