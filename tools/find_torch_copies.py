#!/usr/bin/env python3
"""tools/find_torch_copies.py -- which Python lines of the training step still make torch launch a kernel of its own (copy / fill /
elementwise)?  Runs eager steps under torch.profiler with stacks and prints, per aten operator that launched a device kernel or
memcpy, the innermost frames inside this package.  Run on the GPU box."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device('cuda:0')
mp = straps_amd.synthetic_mean_params(0)
torch.manual_seed(0)
reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs').to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=mp['shape'], use_graph=False)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step()
    torch.cuda.synchronize()
hits = Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if 'straps' in f or 'bench.py' in f]
    where = frames[0] if frames else '(no package frame)'
    hits[(ev.name, ','.join(sorted({k.name[:40] for k in ev.kernels})), where)] += 1
for (name, kern, where), n in sorted(hits.items(), key=lambda kv: -kv[1]):
    print('%3d x %-22s -> %-42s @ %s' % (n, name, kern, where))
