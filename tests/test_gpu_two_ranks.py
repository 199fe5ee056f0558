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
          'loss kernel gradients (d vertices, d joints, d log-variances)', 'SMPL backward (d betas, d rotations)',
          'd estimate after the rot6d backward (input of the IEF backward)', 'd features after the IEF backward (input of the encoder backward)',
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
    w = ts.last['bwd']
    return torch.stack([d(b['shape'], b['rot'], b['cam_t']), d(b['verts'], b['reposed']), d(b['joints2d'], b['joints3d']), d(b['input']),
                        d(b['nzmask'].to(torch.float64)), d(loss), d(w['dverts'], w['djoints'], w['dlv']), d(w['dbetas'], w['drot_smpl']), d(w['dest']),
                        d(w['dfeat']), d(ts.flat_g[:so]), d(ts.flat_g[so:]), d(ts.flat_p)])


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
    ts.keep_bwd = True          # (the stage digests read the backward chain's intermediates: TrainStep.last['bwd'], kept only on request)
    named = [(n, p) for n, p in list(reg.named_parameters()) + [('criterion.' + n, p) for n, p in crit.named_parameters()] if ts.gviews.get(p) is not None]
    losses, stages, pertensor = [], [], []
    for _ in range(6):
        loss = ts.step()[0:1].clone()      # (a replayed graph returns its own output buffer: the next replay of that parity overwrites it)
        losses.append(loss)
        stages.append(_stage_digests(ts, ts.last['loss']))
        pertensor.append(torch.stack([ts.gviews[p].double().sum() for _, p in named]))      # the all-reduced gradient, tensor by tensor
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    stages = [s.cpu().tolist() for s in stages]
    pertensor = [t.cpu().tolist() for t in pertensor]
    digest = torch.stack([ts.flat_p.double().sum(), ts.flat_p.double().abs().sum(), ts.exp_avg.double().abs().sum()]).cpu()
    both = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    q.put((rank, overlap, use_graph, bool(torch.equal(both[0], both[1])), digest.tolist(), losses, ts.graph is not None,
           ts.graph_tail is not None, stages, pertensor, [n for n, _ in named]))
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
                    # the gradient tensor by tensor (registration order: stem, layer1 .. layer4, IEF, loss weights -- backward runs it from the end):
                    # which tensors of this step's all-reduced gradient differ, which do not
                    names = ref[rank][10]
                    bad = [names[t] for t in range(len(names)) if ref[rank][9][step][t] != got[rank][9][step][t]]
                    good = [names[t] for t in range(len(names)) if ref[rank][9][step][t] == got[rank][9][step][t]]
                    return ('step %d, rank %d: the first stage that differs between the eager step and graphs + overlapped exchange is "%s" '
                            '(eager %r, graphs %r); later stages of the same step that differ: %s; gradient tensors that differ: %d of %d -- '
                            'differing: %s; equal: %s' % (step, rank, name, a, b, later or 'none', len(bad), len(names), bad, good))
    return None


# TWO_RANK_REPEAT=n (environment, default 1): each comparison below is made n times -- the difference this file reports shows up in roughly one
# comparison of ten inside a whole-suite run and never when the file runs alone; a hunt raises n instead of re-running the suite
_REPEAT = int(os.environ.get('TWO_RANK_REPEAT', '1'))
_RUNS = {}


def _pair(gm, attempt):
    """(eager step without overlap, split hipGraphs + overlapped exchange) of one attempt; run once, shared by the two tests that read it"""
    key = (gm, attempt)
    if key not in _RUNS:
        _RUNS[key] = (_run(overlap=False, use_graph=False, gm=gm), _run(overlap=True, use_graph=True, gm=gm))
    return _RUNS[key]


@pytest.mark.parametrize('gm', [False, True], ids=['per_rank_mean', 'global_masked_mean'])
def test_two_ranks_replicas_stay_in_sync(gm):
    """the invariants of data parallelism, asserted HARD on every attempt: both replicas hold the same parameters and Adam moments after six steps
    although they trained on different data -- under plain eager steps and under split hipGraphs + the overlapped two-bucket exchange (and with
    TrainStep(global_masked_mean=True): the 1-float count exchange a step ahead of its batch) -- and the split capture did not fall back to eager
    launches."""
    for attempt in range(_REPEAT):
        ref, got = _pair(gm, attempt)
        assert all(r[3] for r in ref) and all(r[3] for r in got), 'attempt %d: the replicas drifted apart' % attempt
        assert ref[0][5] != ref[1][5]                                # ... although they trained on different data
        assert all(r[6] and r[7] for r in got), 'the split hipGraph capture fell back to eager launches'


# Round 5's finding (kept visible instead of retried away), explained and removed.  The STRICT comparison -- every one of thirteen
# per-step stage digests of the graphs + overlap run equal to the eager run's, both ranks -- used to fail in roughly one comparison of fifteen when the
# file ran inside the whole suite.  Every difference caught (TWO_RANK_REPEAT=10, profiles/r05_two_rank_hunt.txt) had one first differing stage, "SMPL
# backward (d betas, d rotations)".  Traced to ONE instruction: the compiler had formed `v_pk_fma_f32 ... op_sel:[0,1,0]` (a packed fp32
# instruction whose LOW result reads the HIGH register of a source) in smpl_pose_bwd_kernel, and on MI355X such an instruction loses the product in lanes
# 48..63 while a bf16x3 convolution workgroup -- here: of the OTHER process -- runs on the same compute unit (a victim of nothing but such instructions,
# checked against plain ones: 0.12 % of the executions fail beside the convolution, none alone).  The kernels that held one are compiled without packed fp32
# instructions and tests/test_packed_fp32_audit.py keeps the library free of them; since then the comparison holds, so it is a plain assertion again.
@pytest.mark.parametrize('gm', [False, True], ids=['per_rank_mean', 'global_masked_mean'])
def test_two_ranks_graphs_and_overlap_equal_the_eager_step_in_every_stage(gm):
    """two-bucket overlapped exchange + split hipGraphs == plain eager step, bit for bit, in EVERY stage of every step on both ranks; a
    difference names the first stage, and the gradient tensors, that differ."""
    for attempt in range(_REPEAT):
        ref, got = _pair(gm, attempt)
        why = _first_difference(ref, got)
        assert why is None, 'attempt %d: %s' % (attempt, why)
        assert [r[5] for r in got] == [r[5] for r in ref] and got[0][4] == ref[0][4]


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
