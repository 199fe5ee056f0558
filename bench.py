#!/usr/bin/env python3
"""bench.py -- bodies/sec of the STRAPS hot path on MI355X (driver contract in the task prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|fwd|smpl | --config 1..4] [--batch B] [--layers 18|50]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch; inputs are generated / resident in HBM:
  train (default; BASELINE.json metric "bodies/sec (train step, B=64, 256x256x17)" = configs[2]):
        full synthetic on-the-fly training step, resnet18, B = 64 per GPU -- SMPL/cam augmentation,
        target SMPL x2, proxy construction + augmentation, forward (training-mode BN), heads +
        multi-task loss, backward, [N>1: one RCCL all-reduce of the flat gradient], Adam.
  fwd   (configs[1]): [B,18,256,256] -> resnet18 encoder -> 3-iter IEF -> rot6d -> SMPL, eval mode.
  smpl  (configs[4]): SMPL-only forward, 65 536 bodies of random (theta, beta) per step x 16 steps = the 1 M bodies BASELINE.json
        names; blend contraction in the three-product fp16 split (--smpl-exact: exact-fp32 MFMA kernel, the A/B).
  --config 3 = configs[3]'s per-GPU shape (train, resnet50, 32 bodies per GPU).
Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel of the workload, its launches timed
live with HIP events on the launch stream inside the timed region; `cpu_baseline` = the CPU oracle
on a bounded sample on this host (rank 0, N = 1 only).
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
from straps_amd import hipabi  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32-input MFMA peak (= fp32 vector peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel`, from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_summary.py --json from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload;
    reads carry the gfx950 x2 correction of MI355X_MICROARCH.md).  Counters cannot be collected from inside this
    process, so the figure is the recorded one; {} -> 'traffic' stays null when no pass exists for this configuration."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')
    tag = '%s_r%d_b%d' % (args.workload, args.layers, args.batch or (65536 if args.workload == 'smpl' else 64))
    if args.workload == 'smpl':
        tag += '_' + args.smpl_precision
    if args.workload != 'smpl' and args.conv_precision != 'bf16x3':
        tag += '_' + args.conv_precision + 'conv'
    try:
        rec = json.load(open(path)).get(tag)
    except (OSError, ValueError):
        return {}
    if not rec:
        return {}
    n = byt = 0.0
    for name, k in rec['kernels'].items():
        nm = name.replace('void ', '')
        if kernel == 'smpl_fwd':          # one straps_smpl_fwd call = pose + vertex + joint kernels: traffic of all three per call
            if nm.startswith('smpl_'):
                byt += k['launches'] * (k['hbm_read_bytes'] + k['hbm_write_bytes'])
                if nm.startswith('smpl_verts'):
                    n += k['launches']
        elif nm.startswith(kernel) or (kernel == 'conv_igemm_x3_kernel' and nm.startswith('conv_igemm_x3h_kernel')):
            n += k['launches']
            byt += k['launches'] * (k['hbm_read_bytes'] + k['hbm_write_bytes'])
    if not n:
        return {}
    return {'traffic': round(byt / n), 'traffic_unit': 'HBM bytes per launch (read x2-corrected + write), launch-weighted mean',
            'traffic_source': rec['source'], 'traffic_measured_in_run': False}


def measure_traffic_live(args, kernel):
    """HBM bytes per launch of `kernel`, MEASURED by this run: two short rocprofv3 passes of this same workload (`--kernel-trace --pmc FETCH_SIZE`,
    then `... WRITE_SIZE`: separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes; reads carry its gfx950 x2 correction, units
    KiB), 2 eager steps each, after the timed region.  {} when rocprofv3 is missing or a pass fails -- the committed figure is used then."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return {}
    flags = ['--workload', args.workload, '--layers', str(args.layers), '--conv-precision', args.conv_precision, '--smpl-in-step', args.smpl_in_step,
             '--smpl-precision', args.smpl_precision, '--smpl-kernel', args.smpl_kernel, '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-graph',
             '--no-overlap', '--no-stem-ab', '--no-reduced-ab', '--no-other-configs', '--no-measure-traffic', '--child']
    if args.batch:
        flags += ['--batch', str(args.batch)]
    match = (lambda n: 'smpl_verts' in n) if kernel == 'smpl_fwd' else (lambda n: ('conv_igemm_x3' in n or 'conv1x1_stream' in n) if kernel == 'conv_igemm_x3_kernel'
                                                                         else (kernel + '<' in n or kernel + '(' in n))
    extra = (lambda n: 'smpl_' in n) if kernel == 'smpl_fwd' else match       # (one straps_smpl_fwd call = pose + vertex + joint kernels: bytes of all three per call)
    means = {}
    tmp = tempfile.mkdtemp(prefix='straps_pmc_', dir='/tmp')
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, ctr)
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', ctr, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__)] + flags
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if p.returncode != 0 or not files:
                return {}
            tot, n = 0.0, 0
            for r in csv.DictReader(open(files[0])):
                if r['Counter_Name'] != ctr:
                    continue
                nm = r['Kernel_Name']
                if extra(nm):
                    tot += float(r['Counter_Value'])
                if match(nm):
                    n += 1
            if not n:
                return {}
            means[ctr] = tot / n
    except Exception:           # noqa: BLE001 -- a failed counter pass must not lose the bench line
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    byt = 2.0 * 1024.0 * means['FETCH_SIZE'] + 1024.0 * means['WRITE_SIZE']
    return {'traffic': round(byt), 'traffic_unit': 'HBM bytes per launch (read x2-corrected + write), launch-weighted mean',
            'traffic_source': 'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over 2 eager steps of this workload',
            'traffic_measured_in_run': True}


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
    """HIP-event pairs around selected C-ABI calls.  torch.cuda.Event records on torch's current
    stream, which is the stream every call is launched on (hipabi.stream_ptr())."""

    def __init__(self):
        self.recs = []
        self.on = False

    def wrap(self, name, flops, fn, nbytes=0.0, cls=None):
        if not self.on:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.recs.append((name, flops, s, e, nbytes, cls))
        return out

    def summary(self):
        agg = {}
        for name, flops, s, e, nbytes, cls in self.recs:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
            a[3] += nbytes
        return agg

    def classes(self, name):
        """launches of kernel family `name` by problem class (the geometry string the proxy attached): a box-to-box or commit-to-commit
        difference of the family's average shows up here as the class that moved."""
        agg = {}
        for nm, flops, s, e, nbytes, cls in self.recs:
            if nm != name or cls is None:
                continue
            a = agg.setdefault(cls, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += s.elapsed_time(e) * 1e-3
        return {k: {'launches': v[0], 'avg_launch_us': round(v[2] / v[0] * 1e6, 2), 'fp32_equivalent_tflops': round(v[1] / v[2] / 1e12, 1)}
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}


class ClockProbe:
    """sustained shader clock of the convolution kernels over a region: straps_set_clock_accumulator (csrc/abi.hip) makes workgroup 0 of
    every implicit-GEMM launch add its shader-clock and wall-clock ticks to a device pair; MHz = ratio x wall-clock rate.  Must be
    created BEFORE the step is captured into a hipGraph (the pointer is a kernel argument)."""

    def __init__(self, dev):
        self.acc = torch.zeros(2, dtype=torch.int64, device=dev)
        self.khz = hipabi.lib().straps_wall_clock_khz()
        hipabi.check(hipabi.lib().straps_set_clock_accumulator(hipabi.ptr(self.acc)), 'straps_set_clock_accumulator')
        self.base = (0, 0)

    def close(self):
        """detach the accumulator (launches after this must not add into a buffer that may be freed)"""
        torch.cuda.synchronize()
        hipabi.check(hipabi.lib().straps_set_clock_accumulator(None), 'straps_set_clock_accumulator')

    def start(self):
        torch.cuda.synchronize()
        self.base = tuple(int(v) for v in self.acc.tolist())

    def mhz(self):
        torch.cuda.synchronize()
        c, w = (int(v) - b for v, b in zip(self.acc.tolist(), self.base))
        return round(c / w * self.khz / 1e3, 1) if w > 0 and self.khz > 0 else None


def _time_launch(fn):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3


def sustained_bf16_mfma(dev):
    """what the bf16 matrix pipe sustains on THIS board under its power budget, measured with a DENSE issue stream (round 6; VERDICT r05 weak #9: the
    probe of rounds 3-5 -- straps_selftest_mfma_bf16, four accumulators per wave -- reads MfmaUtil 71-76 %, and a probe with idle issue slots is not a
    ceiling): straps_selftest_mfma_bf16_dense, eight independent accumulators per wave, two waves per SIMD, no memory traffic, timed with HIP
    events; with operand-like bit patterns (what a convolution feeds the pipe: the power-limited rate) and with all-zero operands (the pipe's
    cheapest data: what it reaches when power does not bind).  Returns (TFLOP/s, MHz) for operand-like data and a dict with both cases and the old
    probe's figure."""
    L = hipabi.lib()
    khz = L.straps_wall_clock_khz()
    out = torch.empty(1024 * 256, device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)

    def mhz():
        c, w = (int(v) for v in clk.tolist())
        return round(c / w * khz / 1e3, 1) if w > 0 and khz > 0 else None
    res = {}
    blocks, iters = 512, 3000
    for data, key in ((1, 'operand_like'), (0, 'zero_operands')):
        secs = _time_launch(lambda: hipabi.check(L.straps_selftest_mfma_bf16_dense(hipabi.ptr(out), hipabi.ptr(clk), blocks, iters, data, hipabi.stream_ptr()), 'mfma dense'))
        res[key] = {'tflops': round(blocks * 4 * iters * 96 * 32768.0 / secs / 1e12, 1), 'sclk_mhz': mhz()}
    secs = _time_launch(lambda: hipabi.check(L.straps_selftest_mfma_bf16(hipabi.ptr(out), hipabi.ptr(clk), 1024, 1500, hipabi.stream_ptr()), 'mfma sustained'))
    res['rounds_3_to_5_probe'] = {'tflops': round(1024 * 4 * 1500 * 48 * 32768.0 / secs / 1e12, 1), 'sclk_mhz': mhz(),
                                  'note': 'four accumulators per wave (MfmaUtil 71-76 %): what rounds 3-5 quoted as the sustained rate'}
    return res['operand_like']['tflops'], res['operand_like']['sclk_mhz'], res


def measure_mfma_util_live():
    """MfmaUtil of the dense probe's launches, from a rocprofv3 counter pass of `bench.py --mfma-probe-only` (kernel trace + one counter only).
    {} when rocprofv3 is missing or the pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return {}
    tmp = tempfile.mkdtemp(prefix='straps_pmc_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '--pmc', 'MfmaUtil', '--output-format', 'csv', '-d', tmp, '--', sys.executable, os.path.abspath(__file__), '--mfma-probe-only']
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
        files = glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True)
        if p.returncode != 0 or not files:
            return {}
        rows = [r for r in csv.DictReader(open(files[0])) if 'mfma_bf16_dense' in r['Kernel_Name'] and r['Counter_Name'] == 'MfmaUtil']
        rows.sort(key=lambda r: int(r['Dispatch_Id']))
        if len(rows) < 6:
            return {}
        # (three launches per case -- two warm-ups and the timed one -- operand-like first)
        return {'operand_like': round(float(rows[2]['Counter_Value']) / 100.0, 4), 'zero_operands': round(float(rows[5]['Counter_Value']) / 100.0, 4)}
    except Exception:           # noqa: BLE001
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def step_dtype(args):
    """the arithmetic the train / forward step computes in: fp32 results and fp32 accumulation everywhere; which operands are carried as
    exact multi-term splits on the bf16 / fp16 matrix pipes is named, not hidden."""
    conv = 'encoder convolutions as exact bf16x3 operand splits (six bf16 products per term)' if args.conv_precision == 'bf16x3' \
        else 'encoder convolutions as exact fp32 MFMA chains'
    smpl = {'fp32': 'in-step SMPL forward as exact fp32 MFMA chains',
            'fp16x3': 'in-step SMPL blend contraction as 3-product fp16 splits (fp16x3)',
            'fp16x3_lbs': 'in-step SMPL blend contraction and skinning as 3-product fp16 splits (fp16x3_lbs: 22-bit operands)'}[args.smpl_in_step]
    return 'fp32 (%s; %s; fp32 accumulate throughout)' % (conv, smpl)


OTHER_CONFIGS = (('configs[1]', ['--config', '1', '--steps', '20', '--warmup', '3']),
                 ('configs[3] per-GPU shape', ['--config', '3', '--steps', '10', '--warmup', '3']),
                 ('configs[4]', ['--config', '4', '--steps', '16', '--warmup', '3']))


def run_other_configs():
    """short timed passes of BASELINE.json's other GPU configurations, one child process each (own hipGraph capture, own clock probe:
    nothing of the headline run's state leaks into them), after the headline's timed region.  Returns {label: compact record}."""
    import subprocess
    res = {}
    for label, flags in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--child'] + flags
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if p.returncode != 0 or not line:
                res[label] = {'error': 'rc %d: %s' % (p.returncode, (p.stderr or p.stdout)[-300:])}
                continue
            d = json.loads(line[-1])
        except Exception as e:          # noqa: BLE001 -- a failed side pass must not lose the headline line
            res[label] = {'error': repr(e)[:300]}
            continue
        roof = d.get('roofline') or {}
        rec = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'warmup': d['warmup'],
               'bodies_per_step': d['config']['bodies_per_gpu_per_step'], 'workload': d['config']['workload'], 'dtype': d['dtype'],
               'launch_mode': d.get('launch_mode'), 'sclk_mhz': d.get('sclk_mhz'),
               'roofline': {k: roof.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'launches') if k in roof},
               'wall_s': round(time.perf_counter() - t0, 1), 'cmd': 'bench.py ' + ' '.join(flags)}
        if 'mfma_side' in roof:
            rec['roofline']['mfma_side_frac'] = roof['mfma_side'].get('frac')
        if 'sustained_mfma' in roof:
            rec['roofline']['kernel_frac_of_sustained_mfma'] = roof['sustained_mfma'].get('kernel_frac_of_sustained')
        res[label] = rec
    return res


def instrument(timer):
    """replace hipabi.lib() by a proxy that times the MFMA kernels with their algorithmic FLOPs
    (2*M*N*K of the convolution they implement)."""
    L = hipabi.lib()

    def conv_flops(B, H, W, Cin, Cout, kh, kw, stride, pad):
        return 2.0 * B * _out(H, kh, stride, pad) * _out(W, kw, stride, pad) * Cout * Cin * kh * kw

    def geo(tag, B, H, W, Cin, Cout, kh, kw, stride, pad):
        return '%s %dx%dx%d %d->%d k%d s%d' % (tag, B, H, W, Cin, Cout, kh, stride)

    def conv_bytes(B, H, W, Cin, Cout, kh, kw, stride, pad):
        # algorithmic HBM bytes of one convolution launch: input + packed weights + output, each touched once (fp32)
        return 4.0 * (B * H * W * Cin + Cout * Cin * kh * kw + B * _out(H, kh, stride, pad) * _out(W, kw, stride, pad) * Cout)

    class Proxy:
        def __getattr__(self, k):
            return getattr(L, k)

        def straps_conv_fwd(self, *a):
            return timer.wrap('conv_igemm_kernel', conv_flops(*a[8:17]), lambda: L.straps_conv_fwd(*a), conv_bytes(*a[8:17]), geo('fwd', *a[8:17]))

        def straps_conv_dgrad(self, *a):
            return timer.wrap('conv_igemm_kernel', conv_flops(*a[4:13]), lambda: L.straps_conv_dgrad(*a), conv_bytes(*a[4:13]), geo('dgrad', *a[4:13]))

        def straps_conv_fwd_x3(self, *a):      # (fp32-equivalent flops: the six bf16 products of a term count as one multiply-add)
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[10:19]), lambda: L.straps_conv_fwd_x3(*a), 1.5 * conv_bytes(*a[10:19]), geo('fwd', *a[10:19]))

        def straps_conv_fwd_x3p(self, *a):
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[11:20]), lambda: L.straps_conv_fwd_x3p(*a), 1.5 * conv_bytes(*a[11:20]), geo('fwd', *a[11:20]))

        def straps_conv_dgrad_x3_bn(self, *a):     # (the launch also carries the next BatchNorm backward's sums in its epilogue)
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[6:15]), lambda: L.straps_conv_dgrad_x3_bn(*a), 1.5 * conv_bytes(*a[6:15]), geo('dgrad+bn', *a[6:15]))

        def straps_conv_dgrad_x3(self, *a):
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[6:15]), lambda: L.straps_conv_dgrad_x3(*a), 1.5 * conv_bytes(*a[6:15]), geo('dgrad', *a[6:15]))

        def straps_conv_dgrad_x3_bn_bits(self, *a):     # (the same launches with ReLU decisions read as bits, round 4; same class names)
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[6:15]), lambda: L.straps_conv_dgrad_x3_bn_bits(*a), 1.5 * conv_bytes(*a[6:15]), geo('dgrad+bn', *a[6:15]))

        def straps_conv_dgrad_x3_bits(self, *a):
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[6:15]), lambda: L.straps_conv_dgrad_x3_bits(*a), 1.5 * conv_bytes(*a[6:15]), geo('dgrad', *a[6:15]))

        # the fp32-operand route of the long 1x1 layers (csrc/conv_x3f.hip, round 6): the same matrix work, counted under the same kernel key (class
        # names carry "f32"); algorithmic bytes = the fp32 tensors (the plane route's operands are 1.5x that)
        def straps_conv_fwd_x3f(self, *a):
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[12:21]), lambda: L.straps_conv_fwd_x3f(*a), conv_bytes(*a[12:21]),
                              geo('fwd f32+bn' if a[1] is not None and getattr(a[1], 'value', a[1]) else 'fwd f32', *a[12:21]))

        def straps_conv_dgrad_x3f(self, *a):
            return timer.wrap('conv_igemm_x3_kernel', conv_flops(*a[6:15]), lambda: L.straps_conv_dgrad_x3f(*a), conv_bytes(*a[6:15]),
                              geo('dgrad+bn f32' if a[16] is not None and getattr(a[16], 'value', a[16]) else 'dgrad f32', *a[6:15]))

        def straps_conv_wgrad_x3f(self, *a):
            return timer.wrap('conv_wgrad_x3_kernel', conv_flops(*a[7:16]), lambda: L.straps_conv_wgrad_x3f(*a), 0.0, geo('wgrad f32', *a[7:16]))

        def straps_conv_wgrad(self, *a):
            return timer.wrap('conv_wgrad_kernel', conv_flops(*a[4:13]), lambda: L.straps_conv_wgrad(*a), 0.0, geo('wgrad', *a[4:13]))

        def straps_conv_wgrad_x3(self, *a):
            return timer.wrap('conv_wgrad_x3_kernel', conv_flops(*a[8:17]), lambda: L.straps_conv_wgrad_x3(*a), 0.0, geo('wgrad', *a[8:17]))

        def straps_stem_fwd(self, *a):
            B, C, H, W = a[8:12]
            return timer.wrap('stem_kernel', 2.0 * B * _out(H, 7, 2, 3) * _out(W, 7, 2, 3) * 64 * C * 49, lambda: L.straps_stem_fwd(*a))

        def straps_stem_wgrad(self, *a):
            B, C, H, W = a[5:9]
            return timer.wrap('stem_wgrad_kernel', 2.0 * B * _out(H, 7, 2, 3) * _out(W, 7, 2, 3) * 64 * C * 49, lambda: L.straps_stem_wgrad(*a))

        def straps_smpl_fwd(self, *a):
            return timer.wrap('smpl_fwd', a[6] * (2.0 * 218 * 20670 + 6890 * 120.0), lambda: L.straps_smpl_fwd(*a))

        def straps_smpl_bwd(self, *a):
            return timer.wrap('smpl_bwd', a[8] * (4.0 * 218 * 20670 + 6890 * 240.0), lambda: L.straps_smpl_bwd(*a))
    proxy = Proxy()
    hipabi.lib = lambda: proxy
    return proxy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=0, help='timed steps (default 20; smpl: 16 x 65 536 = the 1 M bodies of configs[4])')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='train', choices=['train', 'fwd', 'smpl'])
    ap.add_argument('--batch', type=int, default=0, help='bodies per GPU per step (default 64; smpl: 65536)')
    ap.add_argument('--layers', type=int, default=18)
    ap.add_argument('--config', type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help='BASELINE.json configs[N] alias: 1 = --workload fwd, 2 = --workload train, 3 = train --layers 50 --batch 32 (per GPU), 4 = --workload smpl')
    ap.add_argument('--smpl-exact', action='store_true', help='smpl workload: exact-fp32 MFMA blend contraction (= --smpl-precision fp32)')
    ap.add_argument('--smpl-precision', default='fp16x3_lbs', choices=['fp32', 'fp16x3', 'fp16x3_lbs', 'fp16x3_lbs_pd16', 'fp16x3_lbs_p16'],
                    help='smpl workload: fp32 = exact fp32 MFMA chain; fp16x3 = blend contraction as a three-product fp16 split; fp16x3_lbs (default: every '
                         'product split three ways, fp32-class accuracy) = skinning on the matrix pipe too; _pd16 / _p16 = pose-corrective blend in plain '
                         'fp16 (narrower than fp32: opt-in A/B only, reported beside the headline under "reduced_precision_modes")')
    ap.add_argument('--smpl-kernel', default='auto', choices=['auto', 'wide', 'narrow'],
                    help='smpl workload, fp16x3_lbs* modes: auto = by batch size (64-body workgroups from 2048 bodies on), wide / narrow = force one (A/B)')
    ap.add_argument('--no-reduced-ab', action='store_true', help='smpl workload: skip the extra timed pass of the reduced-precision p16 mode')
    ap.add_argument('--conv-precision', default='bf16x3', choices=['fp32', 'bf16x3'],
                    help="encoder convolutions (forward + data gradient): 'fp32' = exact-fp32 MFMA chain, 'bf16x3' = three bf16 planes per fp32 "
                         "operand, six products per term, fp32 accumulate (same accuracy class, bf16 matrix pipe)")
    ap.add_argument('--smpl-in-step', default='fp16x3_lbs', choices=['fp32', 'fp16x3', 'fp16x3_lbs'],
                    help='train / fwd workloads: arithmetic of the SMPL forward calls inside the step')
    ap.add_argument('--global-masked-mean', action='store_true',
                    help='train workload, N > 1: the joints2D task as the masked mean over the GLOBAL batch (one extra 1-float all-reduce per step, '
                         'issued a step ahead); default = the average of per-rank masked means')
    ap.add_argument('--force-exchange', action='store_true',
                    help='train workload: run the gradient exchange even at --gpus 1 (an all-reduce over one rank is the identity, but every call is a real '
                         'RCCL call between the two split hipGraphs): the exact code path of a multi-GPU run, with ranks.exposed_exchange_ms_* and '
                         'ranks.replicas_in_sync on the bench line')
    ap.add_argument('--exchange-backend', default='torch', choices=['torch', 'rccl'],
                    help="gradient exchange through torch.distributed ('torch': the nccl backend == RCCL) or through the library's own C ABI "
                         "('rccl': straps_comm_* / straps_allreduce_grads on a dedicated stream)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a captured hipGraph')
    ap.add_argument('--dense-stem', action='store_true', help='A/B: disable the exact zero skipping of the stem (treat every input cell as non-zero)')
    ap.add_argument('--no-overlap', action='store_true', help='(default now) weight-gradient kernels stay on the main stream')
    ap.add_argument('--mfma-probe-only', action='store_true', help='run the dense sustained-MFMA probe (both operand cases) and exit: the counter pass of measure_mfma_util_live')
    ap.add_argument('--x3f-min-rows', type=int, default=-1, help="A/B: pixel rows from which a 1x1 layer takes the fp32-operand route (default: encoder_exec.X3F_MIN_ROWS)")
    ap.add_argument('--no-x3f-operand-bn', action='store_true', help="A/B: the BatchNorm in front of a fp32-operand 1x1 layer as an apply pass instead of in the operand path")
    ap.add_argument('--no-x3f', action='store_true', help="A/B: the long 1x1 layers on the plane route (rounds 2-5) instead of the fp32-operand route (csrc/conv_x3f.hip)")
    ap.add_argument('--no-relu-bits', action='store_true', help="A/B: a residual unit's ReLU decisions reach the backward pass as fp32 tensors (rounds 1-3) instead of bits")
    ap.add_argument('--no-stem-ab', action='store_true', help='skip the dense-stem A/B steps after the timed region (profiling runs: keeps the kernel stats clean)')
    ap.add_argument('--overlap-wgrad', action='store_true', help='A/B: run the weight-gradient kernels on a side stream (0.1 ms slower since the data pipeline)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help="default headline run only: skip the short timed passes of BASELINE.json's other GPU configurations (configs[1], configs[3]'s "
                         "per-GPU shape, configs[4]) that are reported under 'other_configs'")
    ap.add_argument('--measure-traffic', action='store_true',
                    help="measure roofline.traffic in this run (two short rocprofv3 counter passes after the timed region) instead of reading the "
                         "committed profiles/pmc_traffic.json; on by default for the driver-style default call")
    ap.add_argument('--no-measure-traffic', action='store_true')
    ap.add_argument('--child', action='store_true', help=argparse.SUPPRESS)      # (a pass launched by the headline run for 'other_configs')
    args = ap.parse_args()
    # the driver's one call (no --workload / --config / --layers / --batch) also times the other GPU configurations, each in a child process of
    # its own after the headline ("make the driver's one bench call carry every GPU config")
    headline_default = (args.workload == 'train' and not args.config and args.layers == 18 and not args.batch and not args.child
                        and args.conv_precision == 'bf16x3' and not args.dense_stem and not args.no_graph and not args.no_cpu_baseline)
    if args.child:
        args.no_cpu_baseline = args.no_stem_ab = args.no_reduced_ab = True
    if args.no_relu_bits:
        from straps_amd import encoder_exec as _ee
        _ee._RELU_BITS = False
    if args.no_x3f:
        from straps_amd import encoder_exec as _ee2
        _ee2.X3F_MIN_ROWS = 0
    if args.x3f_min_rows >= 0:
        from straps_amd import encoder_exec as _ee3
        _ee3.X3F_MIN_ROWS = args.x3f_min_rows
    if args.no_x3f_operand_bn:
        from straps_amd import encoder_exec as _ee4
        _ee4.X3F_OPERAND_BN = False
    if args.config:
        args.workload = {1: 'fwd', 2: 'train', 3: 'train', 4: 'smpl'}[args.config]
        if args.config == 3:
            args.layers, args.batch = 50, args.batch or 32
    if not args.steps:
        args.steps = 16 if args.workload == 'smpl' else 20
    if args.smpl_exact:
        args.smpl_precision = 'fp32'
    args.smpl_exact = args.smpl_precision == 'fp32'

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the hot path has no CPU fallback)'
    # (validation hooks, unset in normal use: STRAPS_FORCE_DEVICE puts every rank on one GPU and STRAPS_DIST_BACKEND=gloo lets
    #  the N > 1 control flow -- split hipGraph capture, two-bucket exchange -- be exercised on a single-GPU box)
    if os.environ.get('STRAPS_FORCE_DEVICE') is not None:
        local_rank = int(os.environ['STRAPS_FORCE_DEVICE'])
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if args.mfma_probe_only:
        print(json.dumps(sustained_bf16_mfma(dev)[2]))
        return
    dist = None
    if world > 1 or (args.force_exchange and args.exchange_backend == 'torch'):
        # (--force-exchange at one rank: torch's nccl backend needs a process group of one; the C-ABI backend needs none)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('STRAPS_DIST_BACKEND', 'nccl')  # nccl == RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d needs WORLD_SIZE=%d (got %d): launch with python -m torch.distributed.run --nnodes=1 '
                         '--nproc-per-node %d ... bench.py --gpus %d' % (args.gpus, args.gpus, world, args.gpus, args.gpus))
    ranks_seen = 1
    if dist is not None:           # every rank contributes a one: the sum is the number of ranks that really joined the job
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, 'only %d of %d ranks joined the process group' % (ranks_seen, world)
    hipabi.load()
    probe = ClockProbe(dev)        # (before anything is captured into a hipGraph: the accumulator is a kernel argument)

    B = args.batch or (65536 if args.workload == 'smpl' else 64)
    mp = straps_amd.synthetic_mean_params(0)
    smpl_model = straps_amd.synthetic_smpl_model(0)
    # train / forward workloads: the SMPL calls inside them run the fully split kernel (every product three ways: 7e-7 m from float64,
    # closer than fp32 arithmetic -- not the reduced pose-corrective modes of configs[4]); --smpl-in-step fp32 gives the exact-fp32 chain
    smpl = straps_amd.SMPL(smpl_model, batch_size=B, precision=args.smpl_in_step if args.workload != 'smpl' else 'fp32').to(dev)
    timer = KernelTimer()
    instrument(timer)
    net = 'resnet%d' % args.layers

    if args.workload == 'train':
        from straps_amd.train_step import TrainStep
        torch.manual_seed(1234)                                  # identical replicated weights on every rank
        reg = straps_amd.SingleInputRegressor(18, args.layers, 3, mean_params=mp).to(dev).train()
        reg.image_encoder.dense_stem = args.dense_stem
        reg.image_encoder.conv_precision = args.conv_precision
        crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(
            ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
            init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
        ts = TrainStep(reg, smpl, crit, B, lr=1e-4, rank=rank, world_size=world, seed=1234, mean_shape=mp['shape'], use_graph=not args.no_graph, overlap_wgrad=args.overlap_wgrad and not args.no_overlap,
                       global_masked_mean=args.global_masked_mean, force_exchange=args.force_exchange, exchange_backend=args.exchange_backend)
        step = ts.step
        workload = '%s: full synthetic OTF training step (augmentation + proxy construction + forward + multi-task loss ' \
                   '+ backward + Adam), %s, 18x256x256 proxy' % ('configs[3] per-GPU shape' if args.layers == 50 else 'configs[2]', net)
        dominant = 'conv_igemm_x3_kernel' if args.conv_precision == 'bf16x3' else 'conv_igemm_kernel'
        par = 'data parallel: bodies sharded over %d rank(s), replicated weights, one RCCL sum all-reduce of the flat fp32 gradient per step' % world
        if ts.comm_overlap:
            par += ' in two buckets (layer3.. = %.0f %% of the bytes starts while backward runs through layer2/layer1/stem)' % (
                100.0 * (1.0 - ts.exchange.split_off / ts.flat_g.numel()))
    elif args.workload == 'fwd':
        torch.manual_seed(1234)
        reg = straps_amd.SingleInputRegressor(18, args.layers, 3, mean_params=mp).to(dev).eval()
        reg.image_encoder.dense_stem = args.dense_stem
        reg.image_encoder.conv_precision = args.conv_precision
        x = synthetic_proxy_batch(B, dev, 1234 + rank)           # each rank owns its own shard of bodies

        def step():
            with torch.no_grad():
                cam, pose, shape = reg(x)
                R = straps_amd.rot6d_to_rotmat(pose).view(-1, 24, 3, 3)
                return smpl.forward_arrays(shape.contiguous(), R)[0]
        workload = 'configs[1]: %s encoder + 3-iter IEF + rot6d + SMPL forward-only, 18x256x256 proxy' % net
        dominant = 'conv_igemm_x3_kernel' if args.conv_precision == 'bf16x3' else 'conv_igemm_kernel'
        par = 'bodies sharded over %d rank(s), no collective (forward)' % world
    else:
        g = torch.Generator().manual_seed(rank)
        betas = torch.randn(B, 10, generator=g).to(dev)
        aa = (torch.randn(B, 72, generator=g) * 0.3).to(dev)
        R = straps_amd.batch_rodrigues(aa.view(-1, 3)).view(B, 24, 3, 3).contiguous()

        smpl_precision = args.smpl_precision

        def step():
            return smpl.forward_arrays(betas, R, want_joints=True, precision=smpl_precision, kernel=args.smpl_kernel)[0]
        workload = 'configs[4]: SMPL-only forward, %d random (theta,beta) per step x %d steps = %d bodies -> 6890-vertex meshes + 90 joints' % (
            B, args.steps, B * args.steps)
        dominant = 'smpl_fwd'
        par = 'bodies sharded over %d rank(s), no collective (forward)' % world

    graph_mode = not args.no_graph
    fwd_graph = None
    for _ in range(max(args.warmup, 3 if graph_mode else 0)):
        step()
    torch.cuda.synchronize()
    if graph_mode and args.workload != 'train':
        # static inputs: capture the whole forward once, replay it per step (no host launch cost in the timed region)
        try:
            fwd_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(fwd_graph, capture_error_mode='thread_local'):
                step()
            run = fwd_graph.replay
        except Exception as e:                 # noqa: BLE001 -- eager launches of the same kernels
            sys.stderr.write('hipGraph capture failed (%s); timing eager launches\n' % (e,))
            torch.cuda.synchronize()
            graph_mode, run = False, step
    else:
        run = step
    torch.cuda.synchronize()
    if args.workload == 'train':
        ts.time_exchange, ts.exchange_events = True, []
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timer.on = not graph_mode          # a replayed graph makes no Python-side launches: kernels are timed in the pass below
    probe.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.on = False
    sclk_mhz = probe.mhz()
    exposed_exchange_ms = None
    if args.workload == 'train':
        ts.time_exchange = False
        if ts.exchange_events:
            exposed_exchange_ms = sum(a.elapsed_time(b) for a, b in ts.exchange_events) / len(ts.exchange_events)
    graph_captured = args.workload != 'train' or ts.graph is not None
    eager_ms, eager_sclk_mhz = None, None
    if graph_mode:
        # same K steps launched eagerly with HIP-event pairs around the MFMA kernels (roofline section)
        if args.workload == 'train':
            ts.use_graph = False
            ts.side_stream = None          # one stream: kernels run back to back, so each event pair times ONE kernel alone
            ts.pipeline = False            # (and no next-batch generation running beside the timed kernels)
        step()
        torch.cuda.synchronize()
        timer.on = True
        probe.start()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
        timer.on = False
        eager_sclk_mhz = probe.mhz()
    # A/B inside the same run: the stem kernels with their exact zero skipping defeated (every input cell marked non-zero = the
    # plain dense convolution), 3 eager steps on every rank (the step's all-reduce is collective)
    stem_ab = None
    if args.workload == 'train' and not args.dense_stem and not args.no_stem_ab:
        if ts.use_graph:
            ts.use_graph, ts.side_stream, ts.pipeline = False, None, False
        saved, timer.recs = timer.recs, []
        reg.image_encoder.dense_stem = True
        step()
        torch.cuda.synchronize()
        timer.on = True
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        timer.on = False
        dense = timer.summary()
        timer.recs = saved
        reg.image_encoder.dense_stem = False
        stem_ab = {k: dense[k][2] / 3 * 1e3 for k in ('stem_kernel', 'stem_wgrad_kernel') if k in dense}
    reduced = None
    if args.workload == 'smpl' and smpl_precision == 'fp16x3_lbs' and not args.no_reduced_ab:
        # the pose-corrective blend in plain fp16 (narrower than the reference's fp32; inside north_star's 1e-4 m): NOT the headline --
        # the same K steps timed again and reported beside it
        def step_p16():
            return smpl.forward_arrays(betas, R, want_joints=True, precision='fp16x3_lbs_p16', kernel=args.smpl_kernel)[0]
        for _ in range(3):
            step_p16()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step_p16()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        reduced = {'fp16x3_lbs_p16': {'value': round(B * args.steps / dt2, 1), 'unit': 'bodies/s (this rank)', 'ms_per_step': round(dt2 / args.steps * 1e3, 4),
                                      'launch_mode': 'eager', 'note': 'pose-corrective blend as ONE plain-fp16 product per term: reduced precision, opt-in'}}
    rank_ms = None
    if dist is not None or (args.workload == 'train' and args.force_exchange):
        if dist is not None:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # per-rank view: each rank's own time to finish its K steps (before the closing barrier) and the part of the gradient
        # exchange its backward did not hide
        # (+ a digest of this replica's parameters after the timed steps: data parallelism keeps the replicas bit-identical, and a corrupted
        #  gradient exchange is the one failure that a throughput number would not show)
        dig = [0.0, 0.0, 1.0]
        if args.workload == 'train':
            dig = [float(ts.flat_p.double().sum()), float(ts.flat_p.double().abs().sum()), 1.0 if bool(torch.isfinite(ts.flat_p).all()) else 0.0]
        mine = torch.tensor([local_elapsed / args.steps * 1e3, -1.0 if exposed_exchange_ms is None else exposed_exchange_ms,
                             -1.0 if sclk_mhz is None else sclk_mhz] + dig, device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if dist is not None:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]                  # (--force-exchange --exchange-backend rccl at one rank: no torch process group exists)
        tab = torch.stack(allr).cpu()
        rank_ms = {'ms_per_step_min': round(float(tab[:, 0].min()), 4), 'ms_per_step_max': round(float(tab[:, 0].max()), 4),
                   'per_rank_ms_per_step': [round(float(v), 4) for v in tab[:, 0]]}
        if float(tab[:, 1].max()) >= 0:
            rank_ms.update({'exposed_exchange_ms_mean': round(float(tab[:, 1].mean()), 4), 'exposed_exchange_ms_max': round(float(tab[:, 1].max()), 4),
                            'exposed_exchange_note': 'HIP-event pair on the step stream around GradientExchange.finish(): wait for the tail bucket '
                                                     '(started mid-backward) + the head bucket all-reduce = what backward did not hide'})
        if float(tab[:, 2].max()) > 0:
            rank_ms['per_rank_sclk_mhz'] = [round(float(v), 1) for v in tab[:, 2]]
        if args.workload == 'train':
            rank_ms['replicas_in_sync'] = bool((tab[:, 3] == tab[0, 3]).all() and (tab[:, 4] == tab[0, 4]).all())
            rank_ms['parameters_finite'] = bool((tab[:, 5] == 1.0).all())
            rank_ms['replicas_note'] = 'parameter digests (sum, sum of magnitudes, in float64) of every rank after the timed steps, compared bit for bit'
            rank_ms['exchange'] = {'backend': args.exchange_backend + (' (torch.distributed nccl == RCCL)' if args.exchange_backend == 'torch' else ' (C ABI: straps_allreduce_grads)'),
                                   'forced_at_one_rank': bool(args.force_exchange and world == 1), 'two_buckets': bool(ts.comm_overlap),
                                   'split_graphs': ts.graph_tail is not None, 'tail_bucket_floats': int(ts.flat_g.numel() - ts.exchange.split_off),
                                   'head_bucket_floats': int(ts.exchange.split_off)}
            # what the first multi-GPU run's exposed_exchange_ms_* will be read against: the time a sum all-reduce of the TAIL bucket
            # needs over xGMI (point-to-point, 7 links x 153 GB/s per GPU: MI355X guide), at this run's world size, or at 8 ranks when this run has one
            nw = world if world > 1 else 8
            tail_bytes = 4.0 * (ts.flat_g.numel() - ts.exchange.split_off)
            per_rank = 2.0 * tail_bytes * (nw - 1) / nw                 # bytes every rank sends (reduce-scatter + all-gather)
            rank_ms['exchange']['predicted_tail_allreduce_ms'] = {
                'world_size': nw, 'tail_bucket_bytes': int(tail_bytes), 'bytes_sent_per_rank': int(per_rank),
                'ring_over_one_link': round(per_rank / 153e9 * 1e3, 4), 'all_links_in_parallel': round(per_rank / (153e9 * min(nw - 1, 7)) * 1e3, 4),
                'backward_left_when_it_starts_ms': None,
                'note': 'bytes_sent_per_rank / (153 GB/s x links used): a ring keeps one link per direction busy, a direct exchange all min(N - 1, 7); the tail '
                        'bucket starts when layer3\'s backward is done -- it is hidden if the smaller figure is below the backward time still to run'}

    out = None
    if rank == 0:
        bodies = B * args.steps * world
        agg = timer.summary()
        roof = None
        if dominant in agg:
            n, flops, secs, abytes = agg[dominant]
            ach = flops / secs / 1e12
            roof = {'bound': 'mfma', 'kernel': dominant, 'achieved': round(ach, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None, 'traffic_measured_in_run': False, 'launches': n,
                    'avg_launch_us': round(secs / n * 1e6, 2), 'flops_per_launch': round(flops / n)}
            if abytes:
                roof['algorithmic_bytes_per_launch'] = round(abytes / n)
            if args.workload == 'smpl':
                # algorithmic bytes per body = 6890 x 12 (vertices) + 90 x 12 (joints) + 24 x 36 (rotation matrices) + 40 (betas)
                per_body = 6890 * 12 + 90 * 12 + 24 * 36 + 40
                byt = n * B * per_body
                hbm = {'achieved': round(byt / secs / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(byt / secs / 1e9 / HBM_PEAK_GBS, 4)}
                if args.smpl_exact:
                    roof['hbm_side'] = hbm                # exact-fp32 blend: the fp32 matrix pipe binds long before HBM does
                else:
                    # three-product fp16 split: the contraction issues 3 x 2 x 224 x (tiles x 96) flops per body on the fp16 pipe,
                    # the matrix-pipe skinning another 3 x 2 x 32 x (tiles x 32 x 12)
                    # products per term: 3 everywhere; the pd16 / p16 modes use 2 / 1 from the second 16-column k step on (13 of 14 steps)
                    npose = {'fp16x3_lbs_pd16': 2.0, 'fp16x3_lbs_p16': 1.0}.get(args.smpl_precision, 3.0)
                    issued = n * B * 2.0 * smpl.n_tiles * (96 * 16 * (3.0 + 13.0 * npose) + (3.0 * 32 * 32 * 12 if args.smpl_precision.startswith('fp16x3_lbs') else 0))
                    roof = {'bound': 'hbm', 'kernel': dominant, 'achieved': hbm['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm['frac'],
                            'traffic': None, 'traffic_measured_in_run': False, 'launches': n, 'avg_launch_us': round(secs / n * 1e6, 2),
                            'algorithmic_bytes_per_launch': B * per_body,
                            # (`frac` divides by the kernels' own HIP-event time, `value` by the step time -- both here)
                            'frac_by_step_time': round(B * per_body / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                            'achieved_by_step_time': round(B * per_body / (elapsed / args.steps) / 1e9, 1),
                            'mfma_side': {'pipe': 'fp16 MFMA, fp32 accumulate (split products)', 'issued': round(issued / secs / 1e12, 1), 'peak': 2500.0,
                                          'unit': 'TFLOP/s', 'frac': round(issued / secs / 1e12 / 2500.0, 4),
                                          'fp32_equivalent_tflops': round(ach, 2)}}
            if dominant == 'conv_igemm_x3_kernel':
                # bf16x3 route: every multiply-add of the convolution is issued as six bf16 products on the bf16 pipe (2.5 PFLOP/s dense)
                roof.update({'pipe': 'bf16 MFMA, fp32 accumulate (six products per term of the three-plane split)', 'peak': MFMA_BF16_PEAK_TFLOPS,
                             'achieved': round(6.0 * ach, 2), 'frac': round(6.0 * ach / MFMA_BF16_PEAK_TFLOPS, 4),
                             'fp32_equivalent_tflops': round(ach, 2), 'fp32_pipe_peak': MFMA_F32_PEAK_TFLOPS,
                             'fp32_equivalent_over_fp32_peak': round(ach / MFMA_F32_PEAK_TFLOPS, 4)})
            roof.update(pmc_traffic(args, dominant))
            if (args.measure_traffic or headline_default) and world == 1 and not args.no_measure_traffic and not args.child:
                live = measure_traffic_live(args, dominant)
                if live:
                    roof['traffic_committed_profile'] = roof.get('traffic')      # (the figure of profiles/pmc_traffic.json, for comparison)
                    roof.update(live)
            if dominant == 'conv_igemm_x3_kernel' or (args.workload == 'smpl' and not args.smpl_exact):
                # the spec peak assumes 2.4 GHz; under its power budget the board runs a pure bf16 / fp16 MFMA stream on real data at ~1.5 GHz
                # (tools/mfma_lds_probe.hip).  Measured here, in this process, on this board:
                sus, sus_mhz, sus_all = sustained_bf16_mfma(dev)
                issued = roof['achieved'] if roof.get('bound') == 'mfma' else roof['mfma_side']['issued']
                roof['sustained_mfma'] = {'tflops': sus, 'sclk_mhz': sus_mhz, 'frac_of_spec_peak': round(sus / MFMA_BF16_PEAK_TFLOPS, 4),
                                          'kernel_frac_of_sustained': round(issued / sus, 4), 'cases': sus_all,
                                          'note': 'DENSE register-resident v_mfma_f32_32x32x16_bf16 stream (eight independent accumulators per wave, two waves per '
                                                  'SIMD, no memory traffic: straps_selftest_mfma_bf16_dense) on operand-like data = the power-limited ceiling of the '
                                                  'matrix pipe on this board; cases.zero_operands = the same stream on all-zero operands (power does not bind: the '
                                                  'spec rate); cases.rounds_3_to_5_probe = the four-accumulator probe earlier rounds quoted (75 % dense)'}
                if (args.measure_traffic or headline_default) and world == 1 and not args.no_measure_traffic and not args.child:
                    mu = measure_mfma_util_live()
                    if mu:
                        roof['sustained_mfma']['mfma_util'] = mu['operand_like']
                        roof['sustained_mfma']['mfma_util_zero_operands'] = mu['zero_operands']
                        roof['sustained_mfma']['mfma_util_source'] = 'rocprofv3 --kernel-trace --pmc MfmaUtil over `bench.py --mfma-probe-only`, in this run'
            cls = timer.classes(dominant)
            if cls:
                roof['classes'] = cls
            if eager_sclk_mhz is not None or not graph_mode:
                roof['sclk_mhz_during_measurement'] = eager_sclk_mhz if eager_sclk_mhz is not None else sclk_mhz
        others = {k: {'launches': v[0], 'tflops': round(v[1] / v[2] / 1e12, 2), 'avg_launch_us': round(v[2] / v[0] * 1e6, 2),
                      'ms_per_step': round(v[2] / args.steps * 1e3, 3)} for k, v in agg.items()}
        for k in others:
            if k.startswith('stem'):
                # the stem kernels skip the all-zero strips of the proxy input (exact): their rate is the DENSE-EQUIVALENT one,
                # i.e. the dense conv's flops over the measured time -- not a fraction of the MFMA peak
                others[k]['tflops_dense_equivalent'] = others[k].pop('tflops')
                others[k]['zero_skipping'] = True
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, mp, smpl_model, (smpl, smpl_precision) if args.workload == 'smpl' else None)
        out = {'metric': 'bodies/sec', 'value': round(bodies / elapsed, 1), 'unit': 'bodies/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': step_dtype(args),
               'data': 'synthetic',
               'config': {'workload': workload, 'bodies_per_gpu_per_step': B, 'global_batch': B * world,
                          'input': 'theta(24x3x3), beta(10)' if args.workload == 'smpl' else '18x256x256 fp32 NCHW proxy built on the device by the step itself (rendered part silhouette + 17 joint heat-maps, ~98 % exact zeros as in the reference pipeline)',
                          'parallelism': par},
               'roofline': roof, 'kernels': others, 'cpu_baseline': cpu,
               'sclk_mhz': sclk_mhz, 'sclk_note': 'shader clock the implicit-GEMM convolution kernels ran at over the timed region (s_memtime / s_memrealtime '
                                                  'ticks of workgroup 0 of every launch; spec 2400; null for workloads without convolutions): the chip clocks '
                                                  'to its power budget, boards differ by several per cent'}
        if rank_ms is not None:
            out['ranks'] = rank_ms
        if reduced is not None:
            out['reduced_precision_modes'] = reduced
        if args.workload == 'train':
            out['final_loss'] = round(float(ts.last['loss'][0]), 5)
            captured = graph_mode and graph_captured            # TrainStep falls back to eager launches if capture fails
            out['launch_mode'] = ('hipGraph replay of data-gen (next batch, second stream) + forward + loss + backward (all-reduce and Adam eager)'
                                  if captured else 'eager')
            out['ranks_seen'] = ranks_seen
            out['smpl_in_step'] = args.smpl_in_step
            if stem_ab:
                out['stem_dense_ms_per_step'] = round(sum(stem_ab.values()), 4)
                out['stem_sparse_ms_per_step'] = round(sum(v[2] for k, v in agg.items() if k in stem_ab) / args.steps * 1e3, 4)
        else:
            out['launch_mode'] = 'hipGraph replay of the whole forward' if graph_mode else 'eager'
        if args.workload != 'smpl':
            out['stem_zero_skipping'] = not args.dense_stem
        else:
            out['smpl_blend_precision'] = smpl_precision
            out['dtype'] = {'fp32': 'fp32', 'fp16x3': 'fp32 (blend contraction: 3-product fp16 split, fp32 accumulate)',
                            'fp16x3_lbs': 'fp32 (blend contraction and skinning transforms: 3-product fp16 splits, fp32 accumulate)',
                            'fp16x3_lbs_p16': 'fp32 results; template / shape blend and skinning as 3-product fp16 splits, pose-corrective blend in plain fp16 (1 product)',
                            'fp16x3_lbs_pd16': 'fp32 results; template / shape blend and skinning as 3-product fp16 splits, pose-corrective directions as plain fp16 (2 products)'}[smpl_precision]
        if eager_ms is not None:
            out['eager_ms_per_step'] = round(eager_ms, 4)
            if roof is not None:
                roof['measured_in'] = 'a second pass of the same %d steps launched eagerly (HIP events cannot bracket kernels inside a replayed graph)' % args.steps
        if headline_default and world == 1 and not args.no_other_configs:
            probe.close()
            out['other_configs'] = run_other_configs()
            out['other_configs_note'] = ("short timed passes of BASELINE.json's other GPU configurations, each in a child process started by this "
                                         "run AFTER the headline's timed region (same protocol: warm-up, hipGraph replay, barrier + synchronise; no "
                                         "CPU baseline); 'value' above is configs[2] alone")
        print(json.dumps(out))
    probe.close()
    if args.workload == 'train':
        ts.close()                         # (the C-ABI communicator of --exchange-backend rccl)
    if dist is not None:
        dist.destroy_process_group()
    return out


def cpu_baseline(args, mp, smpl_model, gpu_smpl=None):
    """the CPU oracle (torch-CPU port of the reference path, oracle/straps_oracle.py) on this host:
    bounded sample, `cores` = the torch thread count actually used."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import straps_oracle as O      # checker / baseline only -- never on the product path
    ncores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(ncores)
    init = O.ief_init_estimate(mp['pose'], mp['shape'])
    budget = 12.0
    if args.workload in ('train', 'fwd'):
        torch.manual_seed(1234)
        reg = straps_amd.SingleInputRegressor(18, args.layers, 3, mean_params=mp)
        sd = {k: v.detach().clone() for k, v in reg.state_dict().items()}
        nb = args.batch or 64                      # the GPU run's batch (same B)
        x = synthetic_proxy_batch(nb, 'cpu', 99)
        if args.workload == 'fwd':
            def once():
                with torch.no_grad():
                    O.predict_forward(x, sd, init, smpl_model, args.layers, 3)
            what = 'forward passes of a %d-body batch'
        else:
            names = [n for n, _ in reg.named_parameters()]
            ps = [sd[n].requires_grad_(True) for n in names]
            m_, v_ = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
            tv, _ = O.smpl_forward(smpl_model, torch.zeros(nb, 10), rotmats=torch.eye(3).expand(nb, 24, 3, 3))
            lab = {'verts': tv, 'joints2D': torch.rand(nb, 17, 2) * 256, 'joints3D': torch.zeros(nb, 14, 3), 'shape_params': torch.zeros(nb, 10),
                   'pose_params_rot_matrices': torch.eye(3).expand(nb, 24, 3, 3).contiguous()}
            lab['vis'] = O.check_joints2d_visibility(lab['joints2D'])
            lv = {k: torch.tensor(v, requires_grad=True) for k, v in O.init_log_vars(
                {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).items()}
            stepno = [0]

            def once():
                cam, pose, shape, _ = O.regressor_forward(x, sd, init, args.layers, 3, training=True)
                R = O.rot6d_to_rotmat(pose.contiguous()).view(-1, 24, 3, 3)
                verts, joints = O.smpl_forward(smpl_model, shape, rotmats=R)
                with torch.no_grad():
                    O.smpl_forward(smpl_model, shape.detach(), rotmats=torch.eye(3).expand(nb, 24, 3, 3))      # reposed (metrics)
                pred = {'verts': verts, 'joints2D': O.orthographic_project(joints[:, O.ALL_JOINTS_TO_COCO_MAP], cam),
                        'joints3D': joints[:, O.ALL_JOINTS_TO_H36M_MAP][:, O.H36M_TO_J14], 'shape_params': shape, 'pose_params_rot_matrices': R}
                total, _ = O.multi_task_loss(lab, pred, lv)
                for p in ps:
                    p.grad = None
                total.backward()
                stepno[0] += 1
                with torch.no_grad():
                    O.adam_step([p for p in ps], [p.grad for p in ps], m_, v_, stepno[0])
            what = 'training steps (forward + loss + backward + Adam; data generation excluded) of a %d-body batch'
        once()                                     # 1 warm-up + 3 timed iterations (more while the time budget lasts)
        t0, it = time.perf_counter(), 0
        while it < 3 or (time.perf_counter() - t0 < budget and it < 50):
            once()
            it += 1
        dt = time.perf_counter() - t0
        return {'value': round(nb * it / dt, 2), 'unit': 'bodies/s', 'cores': ncores, 'kind': 'port', 'batch': nb,
                'sample': ('1 warm-up + %d ' + what + ' through the torch-CPU oracle (same net, synthetic input)') % (it, nb)}
    nb = 64
    g = torch.Generator().manual_seed(0)
    betas = torch.randn(nb, 10, generator=g)
    R = O.batch_rodrigues((torch.randn(nb, 72, generator=g) * 0.3).view(-1, 3)).view(nb, 24, 3, 3)
    with torch.no_grad():
        O.smpl_forward(smpl_model, betas[:4], rotmats=R[:4])
        t0, it = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget or it < 2:
            O.smpl_forward(smpl_model, betas, rotmats=R)
            it += 1
        dt = time.perf_counter() - t0
    out = {'value': round(nb * it / dt, 2), 'unit': 'bodies/s', 'cores': ncores, 'kind': 'port',
           'sample': '%d passes of %d bodies through the torch-CPU SMPL oracle' % (it, nb)}
    if gpu_smpl is not None:
        # the checker's other job in this leg: the timed kernel's vertices / joints against the float64 oracle on the same 64 bodies
        # (north_star: <= 1e-4 m)
        smpl_mod, prec = gpu_smpl
        dev = next(smpl_mod.buffers()).device
        with torch.no_grad():
            v, j = smpl_mod.forward_arrays(betas.to(dev), R.to(dev), precision=prec)
            v64, j64 = O.smpl_forward(smpl_model, betas.double(), rotmats=R.double(), dtype=torch.float64)
        out['gpu_max_abs_err_vs_float64_m'] = {'vertices': float('%.3g' % float((v.cpu().double() - v64).abs().max())),
                                               'joints': float('%.3g' % float((j.cpu().double() - j64).abs().max())),
                                               'bodies': nb, 'precision': prec, 'north_star_tolerance_m': 1e-4}
    return out


if __name__ == '__main__':
    main()
