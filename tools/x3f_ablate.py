#!/usr/bin/env python3
"""tools/x3f_ablate.py <Cin> <Cout> [H=64] [B=32] -- one 1x1 layer on the fp32-operand route (forward, raw output + statistics partials) under the tools
build's switches (run through tools/with_tools_lib.py): STRAPS_X3F_ABL (1 no stores, 2 no A loads, 4 no MFMAs, 8 no statistics), STRAPS_X3F_STREAM_WGS;
timed warm (back to back), after a READ flush (cache cold but clean) and after a WRITE flush (cache full of dirty lines), per tile configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402,F401
from straps_amd import hipabi  # noqa: E402
from straps_amd.encoder_exec import weight_planes  # noqa: E402

L = hipabi.lib()
dev = torch.device('cuda:0')
Cin, Cout = int(sys.argv[1]), int(sys.argv[2])
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
flush = torch.empty(1 << 28, device=dev)
x = torch.randn(B, H, H, Cin, device=dev)
w = torch.randn(Cout, Cin, 1, 1, device=dev) * (2.0 / Cin) ** 0.5
wp3, wps = weight_planes(L, w)
y = torch.empty(B, H, H, Cout, device=dev)


def timed(fn, mode, iters=5):
    ts = []
    for _ in range(iters + 1):
        if mode == 'wflush':
            flush.fill_(1.0)
        elif mode == 'rflush':
            torch.sum(flush)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return min(ts[1:])


row = 'abl=%s wgs=%s |' % (os.environ.get('STRAPS_X3F_ABL', '0'), os.environ.get('STRAPS_X3F_STREAM_WGS', '-'))
for cfg in (1, 2, 5):
    nblk = L.straps_conv_x3f_stat_blocks(B, H, H, Cin, Cout, 1, 1, 1, 0, cfg)
    part = torch.empty(max(nblk, 1) * Cout * 2, device=dev)
    fn = lambda: hipabi.check(L.straps_conv_fwd_x3f(hipabi.ptr(x), None, None, 0, hipabi.ptr(wp3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), B, H, H, Cin, Cout,  # noqa: E731
                                                   1, 1, 1, 0, cfg, None), 'fwd_x3f')
    row += ' c%d warm %5.1f rflush %5.1f wflush %5.1f |' % (cfg, timed(fn, 'warm'), timed(fn, 'rflush'), timed(fn, 'wflush'))
print(row, flush=True)
