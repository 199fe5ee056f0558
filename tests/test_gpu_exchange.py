"""GPU: the gradient exchange on RCCL itself, as far as ONE GPU allows.

gloo's blocking CPU collectives (tests/test_distributed_cpu.py, tests/test_gpu_two_ranks.py) prove the arithmetic of the exchange, not
its stream semantics.  Here the real thing runs, with a world of one rank (an all-reduce over one rank is the identity, but every call
is a real RCCL call on RCCL's / the exchange's own stream):
  * the C-ABI exchange of include/straps_hip.h (straps_comm_* / straps_allreduce_grads) on its own;
  * TrainStep(force_exchange=True) with backend 'torch' (torch.distributed `nccl` == RCCL: async work handle on RCCL's stream between the
    two split hipGraphs) and backend 'rccl' (the C-ABI all-reduce on a dedicated stream, ordered with events) -- both bit-identical to
    the step without an exchange over 12 steps, graph capture alive, the exposed-exchange timing populated.
A world of two needs two GPUs (RCCL refuses two ranks on one device): that is the driver's multi-GPU bench."""
import ctypes as C
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exchange_on_one_rank():
    import straps_amd  # noqa: F401
    from straps_amd import hipabi
    L = hipabi.load()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    idbuf = (C.c_char * 128)()
    hipabi.check(L.straps_comm_unique_id(idbuf), 'straps_comm_unique_id')
    assert any(bytes(idbuf))
    comm = C.c_void_p()
    hipabi.check(L.straps_comm_init_rank(idbuf, 1, 0, C.byref(comm)), 'straps_comm_init_rank')
    assert comm.value and L.straps_comm_size(comm) == 1
    assert b'rccl' in L.straps_comm_library()
    g = torch.randn(11_909_794, device=dev)                      # resnet18's flat gradient: 11 909 789 + 5 floats
    want = g.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    # two buckets like the step's: tail first, then head, on a side stream
    split = 683_072
    hipabi.check(L.straps_allreduce_grads(C.c_void_p(g.data_ptr() + 4 * split), g.numel() - split, comm, C.c_void_p(s.cuda_stream)), 'tail')
    hipabi.check(L.straps_allreduce_grads(C.c_void_p(g.data_ptr()), split, comm, C.c_void_p(s.cuda_stream)), 'head')
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(g, want)                                  # sum over one rank
    # argument validation
    assert L.straps_allreduce_grads(None, 4, comm, None) == 1 and b'null pointer' in L.straps_last_error()
    assert L.straps_comm_init_rank(idbuf, 2, 2, C.byref(C.c_void_p())) == 1
    hipabi.check(L.straps_comm_destroy(comm), 'straps_comm_destroy')
    assert L.straps_comm_destroy(None) == 0


def _worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    import straps_amd
    from straps_amd.train_step import TrainStep
    mp_ = straps_amd.synthetic_mean_params(0)
    out = {}
    for name, kw in (('none', dict()), ('torch', dict(force_exchange=True, exchange_backend='torch')),
                     ('rccl', dict(force_exchange=True, exchange_backend='rccl'))):
        torch.manual_seed(1234)
        reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp_).to(dev).train()
        smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=8).to(dev)
        crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
        ts = TrainStep(reg, smpl, crit, 8, lr=1e-3, rank=0, world_size=1, seed=77, mean_shape=mp_['shape'], use_graph=True, **kw)
        ts.time_exchange = True
        losses = [ts.step()[0].clone() for _ in range(12)]
        torch.cuda.synchronize()
        exposed = [a.elapsed_time(b) for a, b in ts.exchange_events]
        import hashlib
        digest = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()      # (plain data only through the queue)
        out[name] = dict(losses=torch.stack(losses).cpu().double().reshape(-1).tolist(), params=digest(ts.flat_p), m=digest(ts.exp_avg),
                         graph=ts.graph is not None, split=ts.graph_tail is not None, overlap=ts.comm_overlap, active=ts.exchange.active,
                         split_off=ts.exchange.split_off, exposed=exposed)
        ts.exchange.close()
        del ts, reg, smpl, crit
    q.put(out)
    dist.destroy_process_group()


def test_train_step_with_the_exchange_forced_on_rccl_world_one():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(29900 + os.getpid() % 1000, q))
    p.start()
    out = q.get(timeout=900)
    p.join(120)
    assert p.exitcode == 0
    ref = out['none']
    assert ref['graph'] and not ref['split'] and not ref['active'] and ref['exposed'] == []
    for name in ('torch', 'rccl'):
        r = out[name]
        assert r['active'] and r['overlap'] and r['split_off'] > 0              # two buckets, tail started mid-backward
        assert r['graph'] and r['split'], '%s: the split hipGraph capture fell back to eager launches' % name
        assert r['losses'] == ref['losses'], name                               # 12 steps, bit for bit
        assert r['params'] == ref['params'] and r['m'] == ref['m'], name        # (sha256 of the flat parameter / first-moment buffers)
        assert len(r['exposed']) == 12 and all(t >= 0.0 for t in r['exposed']), name      # what bench.py reports as exposed_exchange_ms


@pytest.mark.parametrize('backend', ['torch', 'rccl'])
def test_bench_line_with_the_exchange_forced_at_one_rank(backend):
    """`python bench.py --gpus 1 --force-exchange --exchange-backend torch|rccl`: the code path of the first multi-GPU run
    -- split hipGraphs, tail bucket on RCCL under the rest of backward, head bucket, Adam -- driver-visible on a one-GPU box: the bench line carries
    ranks.exposed_exchange_ms_*, ranks.replicas_in_sync and which backend ran."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '3', '--force-exchange', '--exchange-backend', backend,
           '--no-cpu-baseline', '--no-other-configs', '--no-measure-traffic', '--no-stem-ab']
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['metric'] == 'bodies/sec'
    r = d['ranks']
    assert r['exchange']['forced_at_one_rank'] is True and r['exchange']['two_buckets'] is True and r['exchange']['split_graphs'] is True
    assert r['exchange']['backend'].startswith(backend)
    assert r['exchange']['tail_bucket_floats'] > r['exchange']['head_bucket_floats'] > 0
    assert 0 <= r['exposed_exchange_ms_mean'] <= r['exposed_exchange_ms_max'] < d['ms_per_step']
    assert r['replicas_in_sync'] is True and r['parameters_finite'] is True
