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
# bisection knobs (environment): LONGRUN_ONLY=50 (layers), LONGRUN_REPEAT=n, LONGRUN_NO_BITS=1, LONGRUN_TWO_PASS_PROXY=1, LONGRUN_TOOLS=1 (tools
# build: STRAPS_BN_TILED etc. are honoured)
if os.environ.get('LONGRUN_TOOLS'):
    from straps_amd import hipabi
    hipabi.use_library(hipabi.build(tools=True))
if os.environ.get('LONGRUN_NO_BITS'):
    from straps_amd import encoder_exec
    encoder_exec._RELU_BITS = False
if os.environ.get('LONGRUN_TWO_PASS_PROXY'):
    from straps_amd import train_step as _ts
    _ts._FUSED_PROXY_NZ = False
dev = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)


TRACES = []


def run(layers, prec, B, graph, pipe):
    torch.manual_seed(6)
    reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
    reg.image_encoder.conv_precision = prec
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs' if prec == 'bf16x3' else 'fp32').to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
                                                                    init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=graph, pipeline_data=pipe)
    if graph and (os.environ.get('LONGRUN_SYNC_BEFORE_ADAM') or os.environ.get('LONGRUN_SYNC_AFTER_ADAM')):
        opt = ts.optimise

        def wrapped():
            if os.environ.get('LONGRUN_SYNC_BEFORE_ADAM'):
                torch.cuda.synchronize()
            opt()
            if os.environ.get('LONGRUN_SYNC_AFTER_ADAM'):
                torch.cuda.synchronize()
        ts.optimise = wrapped
    if os.environ.get('LONGRUN_SYNC') and graph:      # (a device synchronisation after every step: separates what happens inside one graph from what happens between two)
        rec = []
        for _ in range(steps):
            rec.append(ts.step().clone())
            torch.cuda.synchronize()
        losses = torch.stack(rec).cpu()
    elif os.environ.get('LONGRUN_TRACE') == '2':
        # like LONGRUN_TRACE=1, with the intermediate tensors of the data generation kept (make_batch(keep=...)): the captured graphs then hold
        # them at fixed addresses and their checksums say which STAGE differs first.  (Keeping them alive also stops the allocator from
        # re-using their memory inside the capture: if the difference disappears, it lived in that re-use.)
        keeps = {}
        orig = ts.make_batch

        def mb(out=None, keep=None):
            k = keeps.setdefault(id(out), {})
            return orig(out=out, keep=k)
        ts.make_batch = mb
        names = ('uniforms', 'normals', 'joints', 'joints2d_uncropped', 'seg', 'seg_cropped', 'boxes', 'seg_aug', 'joints2d_input')
        rec, extra = [], []
        for _ in range(steps):
            rec.append(ts.step().clone())
            nb = ts._bufs[ts._cur]
            k = keeps.get(id(nb), {})
            extra.append(torch.stack([k[n].double().sum() if n in k and k[n] is not None else torch.zeros((), device=dev, dtype=torch.float64) for n in names]
                                     + [nb['input'].sum(dtype=torch.float64)]))
        losses = torch.stack(rec).cpu()
        TRACES.append(torch.stack(extra).cpu())
    elif os.environ.get('LONGRUN_TRACE'):
        # per step: the loss record, then checksums of the gradient the step produced, of the parameters after its Adam, and of the batch the
        # NEXT step will train on -- which of them differs first says what was corrupted (enqueued on the step's stream, no synchronisation)
        rec, extra = [], []
        for _ in range(steps):
            rec.append(ts.step().clone())
            nb = ts._bufs[ts._cur] if ts.pipeline else None
            extra.append(torch.stack([ts.flat_g.sum(dtype=torch.float64), ts.flat_p.sum(dtype=torch.float64)]
                                     + ([nb['input'].sum(dtype=torch.float64), nb['verts'].sum(dtype=torch.float64), nb['nzmask'].sum(dtype=torch.float64)] if nb else [])))
        losses = torch.stack(rec).cpu()
        TRACES.append(torch.stack(extra).cpu())
    else:
        losses = torch.stack([ts.step().clone() for _ in range(steps)]).cpu()
    torch.cuda.synchronize()
    return losses, ts.flat_p.clone().cpu(), ts.graph is not None


only = os.environ.get('LONGRUN_ONLY')
for layers, prec, B in [c for c in ((18, 'fp32', 4), (18, 'bf16x3', 6), (50, 'bf16x3', 4)) if not only or str(c[0]) == only] * int(os.environ.get('LONGRUN_REPEAT', '1')):
    if os.environ.get('LONGRUN_TRACE'):
        # two runs of the SAME form (graph + pipeline) against each other, with the per-step checksums
        del TRACES[:]
        la, pa, _ = run(layers, prec, B, True, True)
        lb, pb, _ = run(layers, prec, B, True, True)
        ta, tb = TRACES
        first = next((i for i in range(steps) if not torch.equal(la[i], lb[i]) or not torch.equal(ta[i], tb[i])), None)
        print('r%d graph+pipeline twice: %s' % (layers, 'identical' if first is None else 'first difference at step %d' % first), flush=True)
        if first is not None and os.environ.get('LONGRUN_TRACE') == '2':
            names = ('uniforms', 'normals', 'joints', 'joints2d_uncropped', 'seg', 'seg_cropped', 'boxes', 'seg_aug', 'joints2d_input', 'input')
            for i in range(max(0, first - 1), min(steps, first + 2)):
                print('   step %2d: loss equal %s | next batch: %s' % (i, bool(torch.equal(la[i], lb[i])), '  '.join(
                    '%s %s' % (n, 'same' if bool(ta[i][k] == tb[i][k]) else 'DIFF(%.3g)' % float(ta[i][k] - tb[i][k])) for k, n in enumerate(names))), flush=True)
        elif first is not None:
            for i in range(max(0, first - 1), min(steps, first + 2)):
                print('   step %2d: loss equal %s | grad sum %s | params after Adam %s | next batch: input %s verts %s nzmask %s' % (
                    (i, bool(torch.equal(la[i], lb[i]))) + tuple('same' if bool(ta[i][k] == tb[i][k]) else 'DIFF (%.3e)' % float(ta[i][k] - tb[i][k]) for k in range(5))), flush=True)
        continue
    l0, p0, _ = run(layers, prec, B, False, False)
    for graph, pipe in (((False, True),) if os.environ.get('LONGRUN_EAGER_PIPE') else ((True, True), (True, False))):
        l1, p1, captured = run(layers, prec, B, graph, pipe)
        same = bool(torch.equal(l0, l1) and torch.equal(p0, p1))
        first = next((i for i in range(steps) if not torch.equal(l0[i], l1[i])), None)
        if first is not None and os.environ.get('LONGRUN_VERBOSE'):
            print('   eager loss record at step %d: %s' % (first, ' '.join('%.9g' % v for v in l0[first].tolist())))
            print('   other loss record at step %d: %s' % (first, ' '.join('%.9g' % v for v in l1[first].tolist())))
            nd = [i for i in range(first, steps) if not torch.equal(l0[i], l1[i])]
            print('   steps that differ: %s' % nd)
        print('r%d %s B=%d %d steps | graph (pipeline %s, captured %s) == eager: %s%s | loss %.5f -> %.5f' % (
            layers, prec, B, steps, pipe, captured, same, '' if same else ' (first difference at step %s)' % first, float(l0[0, 0]), float(l0[-1, 0])), flush=True)
