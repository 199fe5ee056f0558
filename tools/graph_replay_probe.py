#!/usr/bin/env python3
"""tools/graph_replay_probe.py [layers] [batch] [replays] -- where does a replayed training-step hipGraph differ from itself?

tools/graph_long_run.py found (round 4) that a resnet50 step captured WITH the data pipeline (the next batch generated on a forked stream
inside the graph) differs from eager launches in about one run of three, at a random step; eager launches with the same two streams never
do, nor does the graph without the fork, nor resnet18.  This probe replays ONE captured graph many times without the optimiser step: the
parameters and the batch the main branch trains on stay what they are, so the flat gradient must come out bit-identical every time (the
forked branch keeps generating new batches into the other buffer set -- irrelevant to the gradient).  On a difference it lists the parameter
tensors whose gradients differ, in registration order: a corrupted forward shows everywhere, a corrupted backward kernel only from its layer
towards the input."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
replays = int(sys.argv[3]) if len(sys.argv) > 3 else 400
pipe = os.environ.get('PROBE_PIPELINE', '1') == '1'
dev = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)
torch.manual_seed(6)
reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs').to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=True, pipeline_data=pipe)
for _ in range(4):
    ts.step()
torch.cuda.synchronize()
assert ts.graph is not None, 'not captured'
par = ts._cur if ts.pipeline else 0
g1, g2, loss = ts.graph[par]
names = [(n, p) for n, p in list(reg.named_parameters()) + list(crit.named_parameters())]


def replay():
    g1.replay()
    if g2 is not None:
        g2.replay()
    torch.cuda.synchronize()
    return ts.flat_g.clone(), loss.clone()


ref_g, ref_l = replay()
bad = 0
for i in range(replays):
    g, l = replay()
    if not torch.equal(g, ref_g) or not torch.equal(l, ref_l):
        bad += 1
        diff = []
        for n, p in names:
            v = ts.gviews.get(p)
            if v is None:
                continue
            off = (v.data_ptr() - ts.flat_g.data_ptr()) // 4
            a, b = g[off:off + v.numel()], ref_g[off:off + v.numel()]
            if not torch.equal(a, b):
                diff.append((n, int((a != b).sum()), v.numel(), float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))))
        print('replay %d differs: loss equal %s; %d of %d gradient tensors differ' % (i, bool(torch.equal(l, ref_l)), len(diff), len(names)))
        for n, k, tot, rel in diff[:6]:
            print('     first   %-44s %8d of %8d elements, max rel %.2e' % (n, k, tot, rel))
        for n, k, tot, rel in diff[-4:]:
            print('     last    %-44s %8d of %8d elements, max rel %.2e' % (n, k, tot, rel))
        if bad >= 4:
            break
print('resnet%d B=%d pipeline %s: %d of %d replays differ from the first' % (layers, B, pipe, bad, i + 1))
