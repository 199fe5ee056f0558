#!/usr/bin/env python3
"""tools/stream_alias_repro.py [offset] -- torch.cuda.Stream() hands out 32 pooled streams round-robin, and torch.cuda.graph captures on ONE
class-level stream taken from the same pool: after enough Stream() calls in a process a TrainStep's data stream IS that capture stream.
This script arranges exactly that (offset 0) or a near miss (offset != 0) and runs the graph-replayed training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

offset = int(sys.argv[1]) if len(sys.argv) > 1 else 0
mode = sys.argv[2] if len(sys.argv) > 2 else 'graph'      # graph | eager | eager_nopipe | graph_nopipe
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
prec = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
dev = torch.device('cuda:0')
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = torch.zeros(4, device=dev) + 1
cap = torch.cuda.graph.default_capture_stream.cuda_stream
keep = []
for i in range(200):
    s = torch.cuda.Stream(device=dev)
    keep.append(s)
    if s.cuda_stream == cap:
        print('capture stream reappears after', i + 1, 'Stream() calls', flush=True)
        break
for _ in range(31 + offset):
    keep.append(torch.cuda.Stream(device=dev))
MP = straps_amd.synthetic_mean_params(0)
torch.manual_seed(6)
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(dev).train()
reg.image_encoder.conv_precision = prec
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=mode.startswith('graph'), pipeline_data=not mode.endswith('nopipe'))
print(mode, 'data stream aliases the capture stream:', ts.data_stream is not None and ts.data_stream.cuda_stream == cap, flush=True)
for i in range(9):
    loss = ts.step()
    torch.cuda.synchronize()
    extra = ''
    if ts.pipeline:
        cur = ts._bufs[1 - ts._cur]
        extra = ' | batch trained on: input %.6f verts %.6f j2d %.4f shape %.6f nz %d | loss parts %s' % (
            float(cur['input'].double().sum()), float(cur['verts'].double().sum()), float(cur['joints2d'].double().sum()), float(cur['shape'].double().sum()),
            int(cur['nzmask'].long().sum()), ' '.join('%.6f' % float(v) for v in loss[1:8]))
    print('step', i, '%.9f' % float(loss[0]), ts.graph is not None, extra, flush=True)
print('done', flush=True)
