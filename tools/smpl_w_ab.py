#!/usr/bin/env python3
"""tools/smpl_w_ab.py -- A/B of the SMPL vertex kernels of mode fp16x3_lbs on one box: the 32-body kernel (narrow: smpl_verts_hh_kernel) against
the 64-body kernel (wide: smpl_verts_w_kernel) and the tools-build instantiations of the latter (STRAPS_SMPL_WVAR = 10 * PF + TV, STRAPS_SMPL_WSV = store forms, STRAPS_SMPL_WABL = ablations; one
process per variant: the switch is read once).  Every variant is first checked against the float64 oracle (ragged batch through the
forced kernel), then timed at 65 536 bodies with joints, HIP events around `iters` back-to-back calls.

    python tools/smpl_w_ab.py [--variants 31,30,21,41] [--iters 12] [--batch 65536] [--modes fp16x3_lbs,fp16x3_lbs_p16]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch
    import straps_amd
    from straps_amd import hipabi
    if args.tools_lib:
        hipabi.use_library(hipabi.build(tools=True))
    import straps_oracle as O
    from detgen import det_uniform
    dev = torch.device('cuda:0')
    model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(model, batch_size=1).to(dev)
    tag = 'WVAR=%s WABL=%s WSV=%s %s %s' % (os.environ.get('STRAPS_SMPL_WVAR', '-'), os.environ.get('STRAPS_SMPL_WABL', '-'), os.environ.get('STRAPS_SMPL_WSV', '-'),
                                            args.kernel, args.mode)
    # ---- parity: ragged batch (groups of 64: 70 -> one full + 6 bodies; 2048 + 37), incl. an extreme body, vs float64 ----
    for Bp in (() if os.environ.get('STRAPS_SMPL_WABL') else (70, 2085)):      # (ablations compute wrong results: timing only)
        betas = torch.from_numpy(det_uniform((Bp, 10), 100 + Bp, -2.5, 2.5))
        betas[0] = torch.tensor([10.0, -8.0, 6.0, 4.0, -4.0, 3.0, 3.0, -3.0, 2.0, 2.0])
        aa = torch.from_numpy(det_uniform((Bp, 72), 200 + Bp, -0.9, 0.9))
        R = O.batch_rodrigues(aa.reshape(-1, 3)).view(Bp, 24, 3, 3)
        v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), precision=args.mode, kernel=args.kernel)
        vn, jn = smpl.forward_arrays(betas.to(dev), R.to(dev), precision=args.mode, kernel='narrow')
        sel = list(range(0, 40)) + list(range(Bp - 40, Bp))
        v64, j64 = O.smpl_forward(model, betas[sel].double(), rotmats=R[sel].double(), dtype=torch.float64)
        ev = float((v[sel].cpu().double() - v64).abs().max())
        ej = float((j[sel].cpu().double() - j64).abs().max())
        dn = float((v - vn).abs().max())      # (0: both kernels run the same accumulation chains)
        print('%s  B=%d  |verts - f64| %.2e  |joints - f64| %.2e  |wide - narrow| %.2e  finite %s' % (tag, Bp, ev, ej, dn, bool(torch.isfinite(v).all())))
    # ---- timing ----
    B = args.batch
    g = torch.Generator().manual_seed(0)
    betas = torch.randn(B, 10, generator=g).to(dev)
    R = straps_amd.batch_rodrigues((torch.randn(B, 72, generator=g) * 0.3).to(dev).view(-1, 3)).view(B, 24, 3, 3).contiguous()
    verts = torch.empty(B, 6890, 3, device=dev)
    joints = torch.empty(B, 90, 3, device=dev)
    for _ in range(3):
        smpl.forward_arrays(betas, R, precision=args.mode, kernel=args.kernel, out_verts=verts, out_joints=joints)
    torch.cuda.synchronize()
    L = hipabi.lib()
    clk = torch.zeros(8, dtype=torch.int64, device=dev)      # (shader ticks, wall ticks) of workgroup 0 of every vertex-kernel launch (+ WABL & 128: phase cycles)
    hipabi.check(L.straps_set_clock_accumulator(hipabi.ptr(clk)), 'straps_set_clock_accumulator')
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        smpl.forward_arrays(betas, R, precision=args.mode, kernel=args.kernel, out_verts=verts, out_joints=joints)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / args.iters
    hipabi.check(L.straps_set_clock_accumulator(None), 'straps_set_clock_accumulator')
    cl = [int(v) for v in clk.tolist()]
    c, w = cl[0], cl[1]
    mhz = c / w * L.straps_wall_clock_khz() / 1e3 if w > 0 else 0.0
    print('%s  B=%d  %.3f ms/call  %.2f M bodies/s  HBM frac %.3f  sclk %.0f MHz' % (tag, B, ms, B / ms / 1e3, B * 84664.0 / (ms * 1e-3) / 8e12, mhz))
    if cl[5]:      # (WABL & 128) shader cycles per tile and phase, averaged over the sampled waves: ideal = 252 x 32 = 8064 (blend), 60 x 32 = 1920 per skinning group
        print('%s  cycles per tile: blend %.0f (MFMA floor 8064)  skin g0 %.0f  skin g1 %.0f (floor 1920 each)  total %.0f (floor 11904)'
              % (tag, cl[2] / cl[5], cl[3] / cl[5], cl[4] / cl[5], (cl[2] + cl[3] + cl[4]) / cl[5]))


def sweep(args):
    """narrow vs wide (product library) over batch sizes: where the automatic choice of straps_smpl_fwd should switch"""
    sys.path.insert(0, ROOT)
    import torch
    import straps_amd
    dev = torch.device('cuda:0')
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=1).to(dev)
    for mode in args.modes.split(','):
        for B in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
            g = torch.Generator().manual_seed(0)
            betas = torch.randn(B, 10, generator=g).to(dev)
            R = straps_amd.batch_rodrigues((torch.randn(B, 72, generator=g) * 0.3).to(dev).view(-1, 3)).view(B, 24, 3, 3).contiguous()
            verts, joints = torch.empty(B, 6890, 3, device=dev), torch.empty(B, 90, 3, device=dev)
            row = []
            for kern in ('narrow', 'wide'):
                for _ in range(3):
                    smpl.forward_arrays(betas, R, precision=mode, kernel=kern, out_verts=verts, out_joints=joints)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = max(4, min(200, int(65536 * 12 / B)))
                s.record()
                for _ in range(n):
                    smpl.forward_arrays(betas, R, precision=mode, kernel=kern, out_verts=verts, out_joints=joints)
                e.record()
                torch.cuda.synchronize()
                row.append(s.elapsed_time(e) / n)
            print('%s  B=%6d  narrow %.4f ms  wide %.4f ms  (wide / narrow %.3f)  %.2f M bodies/s best' % (mode, B, row[0], row[1], row[1] / row[0], B / min(row) / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', default='31,30,21,41')
    ap.add_argument('--ablate', default='', help='comma list of STRAPS_SMPL_WABL values (tools build; timing only): 1 no stores, 2 no fragment '
                    'loads, 4 no skinning chains, 8 no blend MFMAs, 16 no fold, 32 no LDS operand reads (3, 12, 15, 31, 63 = sums)')
    ap.add_argument('--sv', default='', help='comma list of STRAPS_SMPL_WSV values (tools build): 1 non-temporal stores, 2 staggered waves, 4 stores delayed into the next blend phase; sums')
    ap.add_argument('--modes', default='fp16x3_lbs')
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--batch', type=int, default=65536)
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--sweep', action='store_true', help='narrow vs wide over batch sizes (product library), then exit')
    ap.add_argument('--kernel', default='wide')
    ap.add_argument('--mode', default='fp16x3_lbs')
    ap.add_argument('--tools-lib', action='store_true')
    args = ap.parse_args()
    if args.child:
        return child(args)
    if args.sweep:
        return sweep(args)
    base = [sys.executable, os.path.abspath(__file__), '--child', '--iters', str(args.iters), '--batch', str(args.batch)]
    for mode in args.modes.split(','):
        # product library: narrow and wide as shipped
        for kern in ('narrow', 'wide'):
            subprocess.run(base + ['--kernel', kern, '--mode', mode], timeout=600)
        if mode == 'fp16x3_lbs':
            for v in [x for x in args.variants.split(',') if x]:
                subprocess.run(base + ['--kernel', 'wide', '--mode', mode, '--tools-lib'], env=dict(os.environ, STRAPS_SMPL_WVAR=v), timeout=600)
            for v in [x for x in args.sv.split(',') if x]:
                subprocess.run(base + ['--kernel', 'wide', '--mode', mode, '--tools-lib'], env=dict(os.environ, STRAPS_SMPL_WSV=v), timeout=600)
            for v in [x for x in args.ablate.split(',') if x]:
                subprocess.run(base + ['--kernel', 'wide', '--mode', mode, '--tools-lib'], env=dict(os.environ, STRAPS_SMPL_WABL=v), timeout=600)


if __name__ == '__main__':
    main()
