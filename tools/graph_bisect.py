#!/usr/bin/env python3
"""tools/graph_bisect.py <parts> [B] -- which part of the training step misbehaves when it is captured on ONE stream?  parts: none (all eager: the
reference), mb (make_batch in a hipGraph, forward/backward eager), fb (forward/backward in a graph), both (one graph), two (two graphs sharing a pool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

parts = sys.argv[1] if len(sys.argv) > 1 else 'none'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)
torch.manual_seed(6)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(dev).train()
reg.image_encoder.conv_precision = 'fp32'
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=False, pipeline_data=False)
buf = ts._new_buffers()


def clear():
    reg.image_encoder._cache.clear()
    reg.ief_module._cache = {}


with torch.no_grad():
    for i in range(2):
        ts.make_batch(out=buf)
        loss = ts.forward_backward(buf)
        ts.optimise()
        torch.cuda.synchronize()
        print('step', i, '%.9f' % float(loss[0]), flush=True)
    g_mb = g_fb = None
    if parts in ('mb', 'two'):
        g_mb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_mb, capture_error_mode='thread_local'):
            ts.make_batch(out=buf)
    if parts in ('fb', 'two'):
        clear()
        g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fb, pool=g_mb.pool() if g_mb is not None else None, capture_error_mode='thread_local'):
            gl = ts.forward_backward(buf)
    if parts == 'both':
        clear()
        g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fb, capture_error_mode='thread_local'):
            ts.make_batch(out=buf)
            gl = ts.forward_backward(buf)
    for i in range(2, 10):
        if parts != 'both':
            if g_mb is not None:
                g_mb.replay()
            else:
                ts.make_batch(out=buf)
        if g_fb is not None:
            g_fb.replay()
            loss = gl
        else:
            loss = ts.forward_backward(buf)
        ts.optimise()
        torch.cuda.synchronize()
        print('step', i, '%.9f' % float(loss[0]), 'input %.6f' % float(buf['input'].double().sum()), flush=True)
print('done', parts, flush=True)
