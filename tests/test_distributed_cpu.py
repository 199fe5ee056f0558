"""CPU (gloo, world_size 2): the N>1 host logic of the training step -- per-rank data seeds, replicated
weights, ONE sum all-reduce of the flat gradient buffer, identical Adam update on every rank."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import straps_amd
    from straps_amd.train_step import GradientExchange, allreduce_gradients, flatten_parameters
    import straps_oracle as O
    torch.manual_seed(1234)                                     # replicated initial weights
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=straps_amd.synthetic_mean_params(0))
    params = list(reg.parameters())
    flat_p, flat_g, views = flatten_parameters(params, torch.device('cpu'))
    assert flat_p.numel() == 11909789 and len(params) == 66
    assert all(p.data_ptr() >= flat_p.data_ptr() for p in params)           # parameters are views of the flat buffer
    g = torch.Generator().manual_seed(100 + rank)               # per-rank data => per-rank gradients
    for p in params:
        views[p].copy_(torch.randn(p.shape, generator=g) * 1e-2)
    local = flat_g.clone()
    scale = allreduce_gradients(flat_g, world)
    assert scale == 1.0 / world
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    assert torch.allclose(flat_g, sum(gathered), rtol=0, atol=1e-6)
    # the two-bucket exchange the GPU step uses (tail started asynchronously mid-backward, head at the end) gives the same sums
    names = [n for n, _ in reg.named_parameters()]
    split = sum(p.numel() for n, p in zip(names, params) if not n.startswith(('image_encoder.layer3', 'image_encoder.layer4', 'ief_module')))
    assert 0 < split < flat_g.numel()
    for off in (split, 0):
        flat_g.copy_(local)
        ex = GradientExchange(flat_g, off, world)
        ex.start_tail()
        flat_g[:max(off, 1)].mul_(1.0)                          # (the rest of backward keeps writing the head meanwhile)
        assert ex.finish() == 1.0 / world
        assert torch.allclose(flat_g, sum(gathered), rtol=0, atol=1e-6)
    m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    O.adam_step([flat_p], [flat_g * scale], [m], [v], 1)
    digest = torch.stack([flat_p.double().sum(), flat_p.double().abs().sum(), reg.image_encoder.conv1.weight.double().sum()])
    both = [torch.empty_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    q.put((rank, bool(torch.equal(both[0], both[1]))))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(2))
    assert res == {0: True, 1: True}
