#!/usr/bin/env python3
"""tools/mfma_dense_probe.py -- what the bf16 matrix pipe sustains on this board (round 6).

Times straps_selftest_mfma_bf16 (rounds 3-5: four accumulators per wave, four waves per SIMD at 1024 workgroups) against
straps_selftest_mfma_bf16_dense (eight independent accumulators per wave, `__launch_bounds__(256, 2)`) at one and two waves per SIMD, with ALL-ZERO
operands (data 0) and operand-like bit patterns (data 1); prints TFLOP/s and the shader clock each ran at (s_memtime / s_memrealtime of workgroup 0).
Run under `rocprofv3 --kernel-trace --pmc MfmaUtil GRBM_GUI_ACTIVE` the same launches give MfmaUtil per case (tools/r06_gpu_1.sh); the kernel
order in the trace is the order printed here."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from straps_amd import hipabi  # noqa: E402

L = hipabi.lib()
dev = torch.device('cuda:0')
khz = L.straps_wall_clock_khz()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(label, launch, flops):
    out = torch.empty(1024 * 256, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    best = None
    for _ in range(reps):
        launch(out, clk)                       # warm-up (clock settles under this load)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch(out, clk)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        c, w = (int(v) for v in clk.tolist())
        mhz = c / w * khz / 1e3 if w > 0 and khz > 0 else float('nan')
        tf = flops / (ms * 1e-3) / 1e12
        best = (tf, mhz, ms) if best is None or tf > best[0] else best
    tf, mhz, ms = best
    # dense issue at the measured clock = 1024 SIMDs x clock / 32 cycles x 32768 flop
    dense_at_clock = 1024 * mhz * 1e6 / 32 * 32768 / 1e12
    print('%-74s %7.1f TFLOP/s  %6.1f MHz  %7.3f ms  = %.3f of a 32-cycle issue stream at that clock (%.0f TF), %.3f of the 2 500 TF spec'
          % (label, tf, mhz, ms, tf / dense_at_clock, dense_at_clock, tf / 2500.0), flush=True)


iters = 1500
run('rounds 3-5 probe: 4 accumulators, 1024 workgroups (4 waves/SIMD), operand-like',
    lambda o, c: hipabi.check(L.straps_selftest_mfma_bf16(hipabi.ptr(o), hipabi.ptr(c), 1024, iters, hipabi.stream_ptr()), 'selftest'), 1024 * 4 * iters * 48 * 32768.0)
for blocks in (256, 512):
    for data in (0, 1):
        it = 6000 * 256 // blocks
        run('dense probe: 8 accumulators, %4d workgroups (%d wave(s)/SIMD), %s' % (blocks, blocks // 256, 'operand-like data' if data else 'ALL-ZERO operands'),
            lambda o, c, b=blocks, d=data, it=it: hipabi.check(L.straps_selftest_mfma_bf16_dense(hipabi.ptr(o), hipabi.ptr(c), b, it, d, hipabi.stream_ptr()), 'dense'),
            blocks * 4 * it * 96 * 32768.0)
