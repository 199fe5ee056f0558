"""GPU, two ranks on ONE device over gloo: the N > 1 control flow of the training step on real hardware -- split hipGraph
capture, the two-bucket gradient exchange started in the middle of backward, identical updates on both replicas -- as far as
it can be exercised without a second GPU (RCCL itself is only reached by the driver's multi-GPU bench)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what a step leaves behind, in the order the step produces it: the first stage whose digest differs between two launch forms NAMES the kernel
# family at fault (round 4 ended with "digests differ after six steps" and nothing more; round 5: one failing run says where)
STAGES = ('draw-dependent targets (shape, rotations, camera)', 'target vertices (SMPL forward of the data stream)', '2-D joint targets (projection + crop)',
          'network input (rasteriser, crop + resize, augmentation, heat-maps)', 'non-zero map of the input', 'loss record',
          'head bucket of the all-reduced gradient (stem, layer1, layer2)', 'tail bucket of the all-reduced gradient (layer3, layer4, IEF, loss weights)',
          'parameters after Adam')


def _stage_digests(ts, loss):
    """float64 sums of everything step t consumed and produced, enqueued behind the step (no synchronisation): a [len(STAGES), 2] tensor.
    The batch the step trained on sits in the buffer set the data pipeline is NOT writing now (ts._cur flipped at the end of the step)."""
    b = ts._bufs[1 - ts._cur]
    so = ts.tail_offset

    def d(*ts_):
        v = torch.cat([t.reshape(-1).double() for t in ts_])
        ramp = torch.arange(v.numel(), device=v.device, dtype=torch.float64) % 8191.0 + 1.0      # (position-sensitive: a pixel that MOVES changes it)
        return torch.stack([v.sum(), (v * ramp).sum()])
    return torch.stack([d(b['shape'], b['rot'], b['cam_t']), d(b['verts'], b['reposed']), d(b['joints2d'], b['joints3d']), d(b['input']),
                        d(b['nzmask'].to(torch.float64)), d(loss), d(ts.flat_g[:so]), d(ts.flat_g[so:]), d(ts.flat_p)])


def _worker(rank, world, port, overlap, use_graph, q, gm=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import straps_amd
    from straps_amd.train_step import TrainStep
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    mp_ = straps_amd.synthetic_mean_params(0)
    torch.manual_seed(1234)                                     # replicated initial weights
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp_).to(dev).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=8).to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
    ts = TrainStep(reg, smpl, crit, 8, lr=1e-3, rank=rank, world_size=world, seed=77, mean_shape=mp_['shape'], use_graph=use_graph,
                   comm_overlap=overlap, global_masked_mean=gm)
    losses, stages = [], []
    for _ in range(6):
        loss = ts.step()[0:1].clone()      # (a replayed graph returns its own output buffer: the next replay of that parity overwrites it)
        losses.append(loss)
        stages.append(_stage_digests(ts, ts.last['loss']))
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    stages = [s.cpu().tolist() for s in stages]
    digest = torch.stack([ts.flat_p.double().sum(), ts.flat_p.double().abs().sum(), ts.exp_avg.double().abs().sum()]).cpu()
    both = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    q.put((rank, overlap, use_graph, bool(torch.equal(both[0], both[1])), digest.tolist(), losses, ts.graph is not None,
           ts.graph_tail is not None, stages))
    ts.close()
    dist.destroy_process_group()


def _run(overlap, use_graph, gm=False):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    _run.calls = getattr(_run, 'calls', 0) + 1                  # (a fresh rendezvous port per call)
    port = 29700 + (os.getpid() % 1000) + 8 * (_run.calls % 100) + (2 if overlap else 0) + (1 if use_graph else 0) + (4 if gm else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, use_graph, q, gm)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return sorted(q.get(timeout=10) for _ in range(2))


def _first_difference(ref, got):
    """None when the two runs agree in every stage of every step on both ranks, else a sentence naming the first (step, rank, stage) that differs."""
    for step in range(len(ref[0][8])):
        for k, name in enumerate(STAGES):                        # stage-major inside a step: the earliest stage of the step is the origin
            for rank in range(2):
                a, b = ref[rank][8][step][k], got[rank][8][step][k]
                if a != b:
                    later = [STAGES[j] for j in range(k + 1, len(STAGES)) if ref[rank][8][step][j] != got[rank][8][step][j]]
                    return ('step %d, rank %d: the first stage that differs between the eager step and graphs + overlapped exchange is "%s" '
                            '(eager %r, graphs %r); later stages of the same step that differ: %s' % (step, rank, name, a, b, later or 'none'))
    return None


# TWO_RANK_REPEAT=n (environment, default 1): each comparison below is made n times -- the failure this file exists to catch showed up in one of seven
# whole-suite runs of round 5 and never when the file ran alone; a hunt raises n instead of re-running the suite
_REPEAT = int(os.environ.get('TWO_RANK_REPEAT', '1'))


def test_two_ranks_stay_in_sync_and_overlap_changes_nothing():
    for attempt in range(_REPEAT):
        _two_ranks_stay_in_sync_and_overlap_changes_nothing(attempt)


def test_two_ranks_with_the_global_masked_mean_option():
    for attempt in range(_REPEAT):
        _two_ranks_with_the_global_masked_mean_option(attempt)


def _two_ranks_stay_in_sync_and_overlap_changes_nothing(attempt):
    """STRICT (round 5): two-bucket overlapped exchange + split hipGraphs == plain eager step, bit for bit, in EVERY stage of every step on both
    ranks; a failure names the first differing stage.  (Round 4 had loosened this to one-of-three attempts after unexplained failures
    inside whole-suite runs; DESIGN section 1 has the account.)"""
    ref = _run(overlap=False, use_graph=False)
    assert all(r[3] for r in ref)                                # both replicas hold the same parameters and Adam moments
    assert ref[0][5] != ref[1][5]                                # ... although they trained on different data
    got = _run(overlap=True, use_graph=True)
    assert all(r[3] for r in got)
    assert all(r[6] and r[7] for r in got), 'the split hipGraph capture fell back to eager launches'
    why = _first_difference(ref, got)
    assert why is None, 'attempt %d: %s' % (attempt, why)
    assert [r[5] for r in got] == [r[5] for r in ref] and got[0][4] == ref[0][4]


def _two_ranks_with_the_global_masked_mean_option(attempt):
    """TrainStep(global_masked_mean=True): the 1-float count exchange a step ahead of its batch (async, next to the data pipeline) under
    eager launches and under the split hipGraph replay -- replicas stay in sync, graphs == eager bit for bit in every stage (that the
    arithmetic is the global masked mean is tests/test_gpu_backward.py::test_global_masked_mean_loss_two_virtual_ranks_equal_the_global_batch)."""
    ref = _run(overlap=False, use_graph=False, gm=True)
    assert all(r[3] for r in ref)
    got = _run(overlap=True, use_graph=True, gm=True)
    assert all(r[3] for r in got) and all(r[6] and r[7] for r in got)
    why = _first_difference(ref, got)
    assert why is None, 'attempt %d: %s' % (attempt, why)
    assert got[0][4] == ref[0][4] and [r[5] for r in got] == [r[5] for r in ref]


def test_bench_launched_the_way_the_driver_launches_it_two_ranks():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` -- the
    driver's multi-GPU command line -- with the two validation hooks that let it run on a one-GPU box (STRAPS_FORCE_DEVICE=0: both ranks
    on device 0; STRAPS_DIST_BACKEND=gloo: RCCL cannot span two ranks of one device).  Exactly one JSON line, from rank 0, with the
    contract's fields: whole-job bodies/s over both ranks, weak scaling, both ranks seen, per-rank step times and the exposed exchange time."""
    import json
    import subprocess
    env = dict(os.environ, STRAPS_FORCE_DEVICE='0', STRAPS_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2']
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['metric'] == 'bodies/sec' and d['unit'] == 'bodies/s' and d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 2
    assert d['scaling'] == 'weak' and d['higher_is_better'] is True and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert d['config']['global_batch'] == 2 * d['config']['bodies_per_gpu_per_step'] == 128
    assert d['ranks_seen'] == 2
    assert d['value'] == pytest.approx(128 / (d['ms_per_step'] * 1e-3), rel=1e-3)          # whole-job aggregate over the max-over-ranks time
    assert d['cpu_baseline'] is None                                                      # (rank 0 at N = 1 only)
    r = d['ranks']
    assert len(r['per_rank_ms_per_step']) == 2 and r['ms_per_step_min'] <= r['ms_per_step_max'] <= d['ms_per_step'] * 1.02
    assert len(r['per_rank_sclk_mhz']) == 2
    assert 0 <= r['exposed_exchange_ms_mean'] <= r['exposed_exchange_ms_max'] < d['ms_per_step']
    assert r['replicas_in_sync'] is True and r['parameters_finite'] is True                # the digests of both replicas after the timed steps, bit for bit
    assert d['roofline']['kernel'].startswith('conv_igemm') and 0 < d['roofline']['frac'] < 1
