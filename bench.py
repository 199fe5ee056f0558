#!/usr/bin/env python3
"""bench.py -- bodies/sec of the STRAPS hot path on MI355X (driver contract in the task prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fwd|train|smpl] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic proxy representations that is
already resident in HBM:
  fwd   (BASELINE configs[1]): [B,18,256,256] -> resnet18 encoder -> 3-iter IEF -> rot6d -> SMPL
        (vertices + 90 joints), B = 64 per GPU.
  smpl  (BASELINE configs[4]): SMPL-only, B bodies of random (theta, beta) per step.
One JSON line is printed by rank 0.  `roofline` is for the dominant kernel of the workload, its
duration measured live with HIP events on the launch stream inside the timed region;
`cpu_baseline` times the CPU oracle on a bounded sample on this host (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import straps_amd  # noqa: E402
from straps_amd import encoder_exec, hipabi  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32-input MFMA peak
HBM_PEAK_GBS = 8000.0


def synthetic_proxy_batch(B, device, seed):
    """seeded silhouette (union of ellipses, ~25 % foreground) + 17 Gaussian joint heatmaps
    (16x16 truncated, sigma 4) -- the 18-channel input of run_train.py:35, NCHW fp32."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing='ij')
    x = torch.zeros(B, 18, 256, 256)
    cen = torch.rand(B, 10, 2, generator=g) * 140 + 58
    rad = torch.rand(B, 10, 2, generator=g) * 30 + 10
    d = ((xs[None, None] - cen[..., 0, None, None]) / rad[..., 0, None, None]) ** 2 + \
        ((ys[None, None] - cen[..., 1, None, None]) / rad[..., 1, None, None]) ** 2
    x[:, 0] = (d < 1).any(dim=1).float()
    j = (torch.rand(B, 17, 2, generator=g) * 216 + 20).floor()
    dx, dy = xs[None, None] - j[..., 0, None, None], ys[None, None] - j[..., 1, None, None]
    hm = torch.exp(-(dx * dx + dy * dy) / 32.0)
    x[:, 1:] = hm * ((dx.abs() <= 8) & (dy.abs() <= 8))
    return x.to(device)


class KernelTimer:
    """HIP-event pairs around selected launches (torch.cuda.Event records on torch's current
    stream, which is the stream every C-ABI call is launched on -- hipabi.stream_ptr())."""

    def __init__(self):
        self.recs = []

    def wrap(self, name, flops, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.recs.append((name, flops, s, e))
        return out

    def summary(self):
        agg = {}
        for name, flops, s, e in self.recs:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
        return agg


def instrument_encoder(timer):
    """time every implicit-GEMM / stem launch of the encoder with its algorithmic FLOPs."""
    L = hipabi.lib()
    orig_conv, orig_stem = L.straps_conv_fwd, L.straps_stem_fwd

    class Proxy:
        def __getattr__(self, k):
            return getattr(L, k)

        def straps_conv_fwd(self, *a):
            B, H, W, Cin, Cout, kh, kw, stride, pad = a[8:17]
            Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
            return timer.wrap('conv_igemm_kernel', 2.0 * B * Ho * Wo * Cout * Cin * kh * kw, lambda: orig_conv(*a))

        def straps_stem_fwd(self, *a):
            B, C, H, W = a[7:11]
            Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            return timer.wrap('stem_kernel', 2.0 * B * Ho * Wo * 64 * C * 49, lambda: orig_stem(*a))
    return Proxy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='fwd', choices=['fwd', 'smpl'])
    ap.add_argument('--batch', type=int, default=0, help='bodies per GPU per step (default 64; smpl: 65536)')
    ap.add_argument('--layers', type=int, default=18)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the hot path has no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)          # nccl == RCCL on ROCm
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    hipabi.load()

    B = args.batch or (64 if args.workload == 'fwd' else 65536)
    mp = straps_amd.synthetic_mean_params(0)
    smpl_model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    timer = KernelTimer()

    if args.workload == 'fwd':
        torch.manual_seed(1234)                                  # identical replicated weights on every rank
        reg = straps_amd.SingleInputRegressor(18, args.layers, 3, mean_params=mp).to(dev).eval()
        x = synthetic_proxy_batch(B, dev, 1234 + rank)           # each rank owns its own shard of bodies
        proxy = instrument_encoder(timer)

        def step(instrumented):
            with torch.no_grad():
                if instrumented:
                    real = hipabi.lib
                    hipabi.lib = lambda: proxy
                    try:
                        cam, pose, shape = reg(x)
                    finally:
                        hipabi.lib = real
                else:
                    cam, pose, shape = reg(x)
                R = straps_amd.rot6d_to_rotmat(pose).view(-1, 24, 3, 3)
                verts, joints = smpl.forward_arrays(shape.contiguous(), R)
            return verts
        workload = 'configs[1]: resnet18 encoder + 3-iter IEF + rot6d + SMPL forward-only, 18x256x256 proxy' \
            if args.layers == 18 else 'resnet50 encoder + IEF + SMPL forward-only'
        dtype, dominant, bound = 'fp32', 'conv_igemm_kernel', 'mfma'
    else:
        g = torch.Generator().manual_seed(rank)
        betas = torch.randn(B, 10, generator=g).to(dev)
        aa = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
        R = straps_amd.batch_rodrigues(aa.view(-1, 3)).view(B, 24, 3, 3).contiguous()
        verts_buf = {}

        def step(instrumented):
            if instrumented:
                return timer.wrap('smpl_fwd', 0.0, lambda: smpl.forward_arrays(betas, R, want_joints=True)[0])
            return smpl.forward_arrays(betas, R, want_joints=True)[0]
        workload = 'configs[4]: SMPL-only forward, %d random (theta,beta) per step -> 6890-vertex meshes + 90 joints' % B
        dtype, dominant, bound = 'fp32', 'smpl_fwd', 'mfma'

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        bodies = B * args.steps * world
        agg = timer.summary()
        roof = None
        if dominant in agg:
            n, flops, secs = agg[dominant]
            if args.workload == 'smpl':
                # algorithmic FLOPs/body of the blend contraction + skinning: 2*218*20670 + 6890*2*(4*12+12)
                flops = n * B * (2.0 * 218 * 20670 + 6890 * 2.0 * 60)
            ach = flops / secs / 1e12
            roof = {'bound': bound, 'kernel': dominant, 'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None,
                    'launches': n, 'avg_launch_us': round(secs / n * 1e6, 2)}
            if args.workload == 'smpl':
                byt = n * B * (6890 * 12 + 90 * 12 + 24 * 36 + 40)
                roof['hbm_side'] = {'achieved': round(byt / secs / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                    'frac': round(byt / secs / 1e9 / HBM_PEAK_GBS, 4)}
        others = {k: {'launches': v[0], 'tflops': round(v[1] / v[2] / 1e12, 2) if v[1] else None,
                      'avg_launch_us': round(v[2] / v[0] * 1e6, 2)} for k, v in agg.items() if k != dominant}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, mp, smpl_model)
        out = {'metric': 'bodies/sec', 'value': round(bodies / elapsed, 1), 'unit': 'bodies/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
               'config': {'workload': workload, 'bodies_per_gpu_per_step': B, 'global_batch': B * world,
                          'input': '18x256x256 fp32 NCHW proxy (silhouette + 17 heatmaps)' if args.workload == 'fwd' else 'theta(24x3x3), beta(10)',
                          'parallelism': 'bodies sharded over %d rank(s), no collective (forward)' % world},
               'roofline': roof, 'other_kernels': others, 'cpu_baseline': cpu}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return out


def cpu_baseline(args, mp, smpl_model):
    """the CPU oracle (port of the reference path) on this host: bounded sample, all cores."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import straps_oracle as O      # checker / baseline only -- never on the product path
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    init = O.ief_init_estimate(mp['pose'], mp['shape'])
    if args.workload == 'fwd':
        torch.manual_seed(1234)
        reg = straps_amd.SingleInputRegressor(18, args.layers, 3, mean_params=mp)
        sd = {k: v.detach() for k, v in reg.state_dict().items()}
        nb = 8
        x = synthetic_proxy_batch(nb, 'cpu', 99)
        with torch.no_grad():
            O.predict_forward(x[:2], sd, init, smpl_model, args.layers, 3)           # warm-up
            t0, it = time.perf_counter(), 0
            while time.perf_counter() - t0 < 12.0 or it < 2:
                O.predict_forward(x, sd, init, smpl_model, args.layers, 3)
                it += 1
            dt = time.perf_counter() - t0
        return {'value': round(nb * it / dt, 2), 'unit': 'bodies/s', 'cores': ncores, 'kind': 'port',
                'sample': '%d passes of a %d-body batch through the torch-CPU oracle (same net, same input generator)' % (it, nb)}
    nb = 64
    g = torch.Generator().manual_seed(0)
    betas = torch.randn(nb, 10, generator=g)
    R = O.batch_rodrigues((torch.randn(nb, 72, generator=g) * 0.3).view(-1, 3)).view(nb, 24, 3, 3)
    with torch.no_grad():
        O.smpl_forward(smpl_model, betas[:4], rotmats=R[:4])
        t0, it = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0 or it < 2:
            O.smpl_forward(smpl_model, betas, rotmats=R)
            it += 1
        dt = time.perf_counter() - t0
    return {'value': round(nb * it / dt, 2), 'unit': 'bodies/s', 'cores': ncores, 'kind': 'port',
            'sample': '%d passes of %d bodies through the torch-CPU SMPL oracle' % (it, nb)}


if __name__ == '__main__':
    main()
