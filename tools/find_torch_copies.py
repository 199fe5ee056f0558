#!/usr/bin/env python3
"""tools/find_torch_copies.py -- which Python lines of the training step still make torch launch a kernel of its own (device-to-device
copies, fills)?  Wraps the tensor methods that can do so, runs one eager step and prints every call site that really copied / filled a
GPU tensor (a `.contiguous()` of a contiguous tensor is free and is not listed).  Run on the GPU box."""
import os
import sys
import traceback
from collections import Counter

import torch

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

hits = Counter()
ON = [False]


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if 'straps' in f.filename and 'find_torch_copies' not in f.filename:
            return '%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.line)
    return '(outside the package)'


def wrap(name, copies):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        out = orig(self, *a, **k)
        if ON[0] and isinstance(self, torch.Tensor) and self.is_cuda and copies(self, out, a, k):
            hits[(name, site())] += 1
        return out
    setattr(torch.Tensor, name, w)


wrap('contiguous', lambda s, o, a, k: o.data_ptr() != s.data_ptr())
wrap('clone', lambda s, o, a, k: True)
wrap('copy_', lambda s, o, a, k: True)
wrap('fill_', lambda s, o, a, k: True)
wrap('zero_', lambda s, o, a, k: True)
wrap('float', lambda s, o, a, k: o.data_ptr() != s.data_ptr())
wrap('to', lambda s, o, a, k: o.data_ptr() != s.data_ptr())
wrap('index_select', lambda s, o, a, k: True)
for fn in ('zeros', 'zeros_like', 'ones', 'full', 'cat', 'stack', 'tensor'):
    orig = getattr(torch, fn)

    def w(*a, _orig=orig, _fn=fn, **k):
        out = _orig(*a, **k)
        if ON[0] and isinstance(out, torch.Tensor) and out.is_cuda:
            hits[('torch.' + _fn, site())] += 1
        return out
    setattr(torch, fn, w)

ON[0] = True
ts.step()
torch.cuda.synchronize()
ON[0] = False
for (name, where), n in sorted(hits.items(), key=lambda kv: -kv[1]):
    print('%3d x %-18s @ %s' % (n, name, where))
print('total', sum(hits.values()))
