#!/usr/bin/env python3
"""tools/two_rank_bisect.py -- tests/test_gpu_two_ranks.py::test_two_ranks_stay_in_sync_and_overlap_changes_nothing with the round's switches
flipped through the environment (TWO_RANK_NO_BITS=1, TWO_RANK_TWO_PASS_PROXY=1, TWO_RANK_STEPS=n): eager vs split-graph + overlapped exchange."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, overlap, use_graph, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import straps_amd
    from straps_amd import encoder_exec, train_step
    from straps_amd.train_step import TrainStep
    if os.environ.get('TWO_RANK_NO_BITS'):
        encoder_exec._RELU_BITS = False
    if os.environ.get('TWO_RANK_TWO_PASS_PROXY'):
        train_step._FUSED_PROXY_NZ = False
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    mp_ = straps_amd.synthetic_mean_params(0)
    torch.manual_seed(1234)
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp_).to(dev).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=8).to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
    ts = TrainStep(reg, smpl, crit, 8, lr=1e-3, rank=rank, world_size=world, seed=77, mean_shape=mp_['shape'], use_graph=use_graph, comm_overlap=overlap)
    names = [n for n, p in list(reg.named_parameters()) + list(crit.named_parameters()) if ts.gviews.get(p) is not None]
    views = [ts.gviews[p] for n, p in list(reg.named_parameters()) + list(crit.named_parameters()) if ts.gviews.get(p) is not None]
    losses, trace = [], []
    for _ in range(int(os.environ.get('TWO_RANK_STEPS', '6'))):
        losses.append(ts.step()[0].clone())
        if os.environ.get('TWO_RANK_TRACE'):      # per-tensor checksums of the (all-reduced) gradient this step applied, enqueued without a synchronisation
            trace.append(torch.stack([v.double().sum() for v in views]))
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    q.put((rank, losses, float(ts.flat_p.double().sum()), [t.cpu().tolist() for t in trace], names))
    dist.destroy_process_group()


def run(overlap, use_graph, port):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, overlap, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    return sorted(q.get(timeout=10) for _ in range(2))


if __name__ == '__main__':
    base = 29900 + os.getpid() % 50
    a = run(False, False, base)
    b = run(True, True, base + 1)
    c = run(False, False, base + 2) if not os.environ.get('TWO_RANK_SKIP_AGAIN') else a
    for name, r in (('eager', a), ('graph+overlap', b), ('eager again', c)):
        print(name, 'rank0 losses', ['%.6f' % v for v in r[0][1]], 'param sum %.6f' % r[0][2], flush=True)
    print('graph == eager:', a[0][1] == b[0][1] and a[0][2] == b[0][2], '| eager == eager again:', a[0][1] == c[0][1] and a[0][2] == c[0][2])
    if os.environ.get('TWO_RANK_TRACE'):
        for name, x, y in (('graph vs eager', a, b), ('eager vs eager again', a, c)):
            ta, tb, names = x[0][3], y[0][3], x[0][4]
            for step in range(len(ta)):
                d = [names[k] for k in range(len(names)) if ta[step][k] != tb[step][k]]
                if d:
                    print('%s: first step whose applied gradient differs: %d; %d of %d tensors differ; first %s ... last %s' % (name, step, len(d), len(names), d[:4], d[-3:]))
                    break
            else:
                print('%s: every gradient tensor of every step equal' % name)
