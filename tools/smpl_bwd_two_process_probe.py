#!/usr/bin/env python3
"""tools/smpl_bwd_two_process_probe.py [calls] -- round 5: the two-rank test names straps_smpl_bwd, launched eagerly while a second PROCESS trains on the same
GPU, as the first stage that is occasionally not bit-reproducible.  This probe isolates it: ONE process calls straps_smpl_bwd over and
over on fixed inputs (8 bodies, fresh torch.empty workspace per call, like the eager step) and compares, on the device, every region of the workspace
and both outputs with the first call's --
    F, A        what smpl_pose_kernel recomputes (pose features, joint transforms)
    dF, dA      the per-chunk partials smpl_verts_bwd_kernel hands to smpl_pose_bwd_kernel
    dbetas, drotmats   the results
-- in three situations: alone with GARBAGE in every fresh allocation (a read of unwritten workspace shows at once); beside a second process that runs
eager training steps (8 bodies, resnet18); beside one that replays the step as hipGraphs."""
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def aggressor(use_graph, stop):
    sys.path.insert(0, ROOT)
    import straps_amd
    from straps_amd.train_step import TrainStep
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    mp_ = straps_amd.synthetic_mean_params(0)
    torch.manual_seed(1234)
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp_).to(dev).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=8).to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']).to(dev)
    ts = TrainStep(reg, smpl, crit, 8, lr=1e-4, seed=5, mean_shape=mp_['shape'], use_graph=use_graph)
    n = 0
    while not stop.is_set():
        ts.step()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def victim(label, garbage):
    import ctypes as C
    import straps_amd
    from straps_amd import hipabi
    L = hipabi.use_library(hipabi.build(tools=True)) if os.environ.get('PROBE_TOOLS') else hipabi.load()      # (PROBE_TOOLS: through the tools build of the library)
    dev = torch.device('cuda:0')
    B = 8
    g = torch.Generator().manual_seed(3)
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
    betas = torch.randn(B, 10, generator=g).to(dev)
    R = straps_amd.batch_rodrigues((torch.randn(B, 72, generator=g) * 0.4).to(dev).view(-1, 3)).view(B, 24, 3, 3).contiguous()
    dverts = (torch.randn(B, 6890, 3, generator=g) * 1e-3).to(dev)
    djoints = (torch.randn(B, 90, 3, generator=g) * 1e-3).to(dev)
    nws = L.straps_smpl_bwd_workspace_bytes(B, 0) // 4
    KP, nA = 224, 288
    nch = (nws // B - KP - nA) // (KP + nA)
    regions = {'F': (0, B * KP), 'A': (B * KP, B * (KP + nA)), 'dF partials': (B * (KP + nA), B * (KP + nA) + nch * B * KP),
               'dA partials': (B * (KP + nA) + nch * B * KP, nws)}
    ref = None
    bad = torch.zeros(6, device=dev, dtype=torch.int64)
    worst = torch.zeros(6, device=dev, dtype=torch.float64)
    for it in range(CALLS):
        if garbage:      # fresh allocations of the workspace's size come back full of finite garbage
            junk = [torch.empty(nws, device=dev).uniform_(-3, 3) for _ in range(3)] + [torch.empty(B, 24, 3, 3, device=dev).uniform_(-3, 3), torch.empty(B, 10, device=dev).uniform_(-3, 3)]
            del junk
        ws = torch.empty(nws, device=dev)
        dbetas, drot = torch.empty(B, 10, device=dev), torch.empty(B, 24, 3, 3, device=dev)
        hipabi.check(L.straps_smpl_bwd(C.byref(smpl._model_struct()), hipabi.ptr(betas), hipabi.ptr(R), hipabi.ptr(dverts), hipabi.ptr(djoints), hipabi.ptr(dbetas),
                                       hipabi.ptr(drot), hipabi.ptr(ws), B, 0, hipabi.stream_ptr()), 'straps_smpl_bwd')
        cur = [ws[a:b] for a, b in regions.values()] + [dbetas.reshape(-1), drot.reshape(-1)]
        if ref is None:
            ref = [c.clone() for c in cur]
            continue
        for k, (c, r) in enumerate(zip(cur, ref)):
            d = (c != r)
            bad[k] += d.any().to(torch.int64)
            worst[k] = torch.maximum(worst[k], ((c.double() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-300)))
    torch.cuda.synchronize()
    names = list(regions) + ['dbetas', 'drotmats']
    print('%-46s %d calls: calls that differed from the first -- ' % (label, CALLS - 1) + ', '.join('%s %d (max rel %.1e)' % (n, int(b), float(w)) for n, b, w in zip(names, bad.tolist(), worst.tolist())), flush=True)


if __name__ == '__main__':
    torch.cuda.set_device(0)
    victim('alone, garbage in every fresh allocation:', True)
    victim('alone:', False)
    ctx = mp.get_context('spawn')
    for use_graph, label in ((False, 'beside a process training EAGERLY:'), (True, 'beside a process replaying hipGraphs:'), (False, 'beside TWO processes training eagerly:')):
        stop = ctx.Event()
        procs = [ctx.Process(target=aggressor, args=(use_graph, stop)) for _ in range(2 if 'TWO' in label else 1)]
        for p in procs:
            p.start()
        time.sleep(25)                       # (imports + warm-up of the aggressor)
        victim(label, False)
        stop.set()
        for p in procs:
            p.join(120)
