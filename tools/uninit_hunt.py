#!/usr/bin/env python3
"""tools/uninit_hunt.py [B] [conv_precision] [layers] -- does any kernel of the training step read memory nobody wrote?  torch.empty is made to
fill new tensors with NaN (floats) / the largest value (integers) (torch.utils.deterministic.fill_uninitialized_memory); a read-before-write then
shows as a NaN loss, a changed loss, or a memory fault -- with STRAPS_TRACE_CALLS=<file> the last line of the trace names the entry point."""
import os
import sys

import torch

torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 18
dev = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)
torch.manual_seed(6)
reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
reg.image_encoder.conv_precision = prec
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs' if prec == 'bf16x3' else 'fp32').to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=False, pipeline_data=False)
for i in range(4):
    loss = ts.step()
    torch.cuda.synchronize()
    print('step', i, '%.9f' % float(loss[0]), 'grad finite:', bool(torch.isfinite(ts.flat_g).all()), 'params finite:', bool(torch.isfinite(ts.flat_p).all()), flush=True)
print('done', B, prec, layers, flush=True)
