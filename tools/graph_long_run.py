#!/usr/bin/env python3
"""tools/graph_long_run.py [steps] -- hipGraph replay (data pipeline on) against eager launches without the pipeline over many steps, bit for bit:
resnet18 on both convolution routes and resnet50, small batches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)


def run(layers, prec, B, graph, pipe):
    torch.manual_seed(6)
    reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
    reg.image_encoder.conv_precision = prec
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs' if prec == 'bf16x3' else 'fp32').to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                    init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=graph, pipeline_data=pipe)
    losses = torch.stack([ts.step().clone() for _ in range(steps)]).cpu()
    torch.cuda.synchronize()
    return losses, ts.flat_p.clone().cpu(), ts.graph is not None


for layers, prec, B in ((18, 'fp32', 4), (18, 'bf16x3', 6), (50, 'bf16x3', 4)):
    l0, p0, _ = run(layers, prec, B, False, False)
    for graph, pipe in ((True, True), (True, False)):
        l1, p1, captured = run(layers, prec, B, graph, pipe)
        same = bool(torch.equal(l0, l1) and torch.equal(p0, p1))
        first = next((i for i in range(steps) if not torch.equal(l0[i], l1[i])), None)
        print('r%d %s B=%d %d steps | graph (pipeline %s, captured %s) == eager: %s%s | loss %.5f -> %.5f' % (
            layers, prec, B, steps, pipe, captured, same, '' if same else ' (first difference at step %s)' % first, float(l0[0, 0]), float(l0[-1, 0])), flush=True)
