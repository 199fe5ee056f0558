#!/usr/bin/env python3
"""Are torch's OWN kernels victims of the packed-operand-select hazard when they run beside this library's convolutions?

tools/audit_neighbours.py finds the form measured to misexecute -- packed fp32 arithmetic with a low-half select on src1 -- in 782 gfx950 functions of
libtorch_hip.so (profiles/r06_neighbour_audit.txt): pow, softplus, mish, log1p, logaddexp, binary_cross_entropy, the _foreach_* kernels of the fused
optimizers, ...  This probe runs a handful of those operators on a SIDE stream while `straps_conv_fwd_x3` launches run on another, and counts the
elements whose bit pattern differs from the same operator's result computed alone.  Controls: the same loop with no aggressor, and beside the exact-fp32
convolution (fp32 MFMAs: no aggressor in round 5's measurements).

The product's step never does this (no torch kernel runs inside a step; RCCL's kernels are clean: same audit) -- the probe answers what a HOST program
that overlaps its own torch work with the step on another stream would see.   python tools/torch_victim_probe.py [iterations]   (needs the GPU)"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import straps_amd  # noqa: F401,E402
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import split3, weight_planes  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device('cuda:0')
    hipabi.load()
    L = hipabi.lib()
    g = torch.Generator(device='cpu').manual_seed(5)
    # the aggressors: layer1 of resnet18 at the bench's size, bf16x3 route and exact-fp32 route
    B, H, C = 64, 64, 64
    x = (torch.rand(B, H, H, C, generator=g) * 2 - 1).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    xp, xps = split3(L, x)
    wp3, wps = weight_planes(L, w, False)
    wp = torch.empty_like(w)
    hipabi.check(L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(wp), C, C, 3, 3, None), 'pack')
    y = torch.empty(B, H, H, C, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def aggress_x3():
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(xp), xps, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y), None, B, H, H, C, C, 3, 3, 1, 1, 0,
                                          sa.cuda_stream), 'conv_fwd_x3')

    def aggress_fp32():
        hipabi.check(L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), None, None, None, 0, hipabi.ptr(y), None, B, H, H, C, C, 3, 3, 1, 1, 0, sa.cuda_stream), 'conv_fwd')

    n = int(os.environ.get('PROBE_N', 1 << 22))
    a = (torch.rand(n, generator=g) * 3 + 0.25).to(dev)
    b = (torch.rand(n, generator=g) * 2 - 1).to(dev)
    p = torch.rand(n, generator=g).clamp(1e-3, 1 - 1e-3).to(dev)
    t = (torch.rand(n, generator=g) > 0.5).float().to(dev)
    lst = [a[i * (n // 16):(i + 1) * (n // 16)].clone() for i in range(16)]
    ops = {
        'pow(a, b)': lambda: torch.pow(a, b),
        'softplus(b)': lambda: F.softplus(b),
        'mish(b)': lambda: F.mish(b),
        'log1p(a)': lambda: torch.log1p(a),
        'logaddexp(a, b)': lambda: torch.logaddexp(a, b),
        'binary_cross_entropy(p, t, reduction=none)': lambda: F.binary_cross_entropy(p, t, reduction='none'),
        'addcmul(a, b, p, value=0.3)': lambda: torch.addcmul(a, b, p, value=0.3),
        'xlogy-free pow(1.3, b)': lambda: torch.pow(1.3, b),
        'atanh(b * 0.9)': lambda: torch.atanh(b * 0.9),
        'log_sigmoid(b)': lambda: F.logsigmoid(b),
        '_foreach_pow(16 tensors, 1.5)': lambda: torch.cat(torch._foreach_pow(lst, 1.5)),
        '_foreach_sqrt(16 tensors)': lambda: torch.cat(torch._foreach_sqrt(lst)),
        '_foreach_mul(16 tensors, 1.7)': lambda: torch.cat(torch._foreach_mul(lst, 1.7)),
        'a * b + p (plain elementwise)': lambda: a * b + p,
        'sum(a * b) (reduce)': lambda: (a * b).view(1024, -1).sum(1),
    }
    # do the two streams overlap at all?  wall time of the convolution launches alone, the operator calls alone, both
    import time

    def timed(with_conv, with_ops):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            if with_conv:
                aggress_x3()
                aggress_x3()
            if with_ops:
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        F.softplus(b)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    timed(True, True)
    tc, to, tb = timed(True, False), timed(False, True), timed(True, True)
    print('overlap check (softplus): convolutions alone %.1f ms, operator calls alone %.1f ms, both %.1f ms (serial would be %.1f)' % (tc, to, tb, tc + to))
    print('%d iterations per operator and aggressor; each iteration: 2 convolution launches on one stream, 4 operator calls on another; %d elements per call'
          % (iters, n))
    print('%-46s %14s %22s %26s' % ('operator (fp32)', 'alone', 'beside exact-fp32 conv', 'beside bf16x3 conv (x3)'))
    total = 0
    for name, op in ops.items():
        ref = op().view(torch.int32).clone()
        torch.cuda.synchronize()
        row = []
        for agg in (None, aggress_fp32, aggress_x3):
            bad = torch.zeros((), device=dev, dtype=torch.int64)
            calls_bad = torch.zeros((), device=dev, dtype=torch.int64)
            sb.wait_stream(torch.cuda.current_stream())
            for _ in range(iters):
                if agg is not None:
                    agg()
                    agg()
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        d = (op().view(torch.int32) != ref).sum()
                        bad += d
                        calls_bad += (d > 0).long()
            torch.cuda.synchronize()
            row.append((int(bad), int(calls_bad)))
        total += row[2][0]
        print('%-46s %14s %22s %26s' % ((name,) + tuple('%d in %d calls' % r for r in row)))
    print('differing elements beside the bf16x3 convolution, all operators: %d' % total)


if __name__ == '__main__':
    main()
