"""GPU parity tests of the bf16x3 convolution route (csrc/conv_x3.hip): fp32 operands carried as three bf16 planes, six bf16
products per term, fp32 accumulate -- the replacement of the exact-fp32 MFMA chain for the forward convolution and its data
gradient (models/resnet.py:61-77, 101-121 and their backward).

Bars (written where used): the split is EXACT (bit-for-bit reconstruction); the convolutions meet the same bars against the
float64 convolution as the exact-fp32 kernels do in test_gpu_forward.py / test_gpu_backward.py (2e-5 abs + 2e-5 rel; data gradient
2e-5 of the maximum), and their error is compared with the fp32 kernels' on the same inputs; fused plane outputs equal a split
pass over the fp32 output bit for bit; the training step on this route stays inside the bars of the fp32 route against the
float64 oracle (tests/test_gpu_train_step.py, parametrised there).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import straps_amd
from detgen import det_uniform
from straps_amd import hipabi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    hipabi.load()
    return torch.device('cuda:0')


def _split_linear(t):
    """planes in the order of t (straps_split3_bf16: the exactness tests)"""
    L = hipabi.lib()
    n = t.numel()
    ps = (n + 7) // 8 * 8
    planes = torch.zeros(3, ps, device=t.device, dtype=torch.int16)
    hipabi.check(L.straps_split3_bf16(hipabi.ptr(t), hipabi.ptr(planes), n, ps, None), 'split3')
    return planes, ps


def _split(t):
    """chunk-major planes of an NHWC activation / gradient tensor [..., C] (what the convolution kernels read)"""
    from straps_amd.encoder_exec import split3
    return split3(hipabi.lib(), t)


def _wsplit(dev, w, dgrad=False):
    """chunk-major planes of a convolution weight (OIHW), forward or data-gradient layout"""
    from straps_amd.encoder_exec import weight_planes
    return weight_planes(hipabi.lib(), w.float().contiguous().to(dev), dgrad)


def _cm(t):
    """torch restatement of the chunk-major order (csrc/common.h cm_index) of an NHWC tensor [..., C]: [C/32][rows][32] flattened"""
    C = t.shape[-1]
    return t.reshape(-1, C // 32, 32).permute(1, 0, 2).contiguous().reshape(-1)


def _planes_to_f64(planes, n):
    """bf16 bit patterns [3][ps] -> float64 sum of the three planes (exact: 24 significant bits)."""
    bits = planes[:, :n].cpu().numpy().view(np.uint16).astype(np.uint32) << 16
    return bits.view(np.float32).astype(np.float64).sum(0)


def test_split_is_exact(dev):
    """x == p1 + p2 + p3 bit for bit (|x| >= 2^-110): normal range, tiny / huge magnitudes, zeros, exact powers of two, an odd element count."""
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.standard_normal(100003).astype(np.float32),
                        (rng.standard_normal(4096) * 1e-25).astype(np.float32), (rng.standard_normal(4096) * 1e30).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -126, 2.0 ** 100, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0, 255.0, 257.0], np.float32)])
    t = torch.from_numpy(x).to(dev)
    planes, ps = _split_linear(t)
    torch.cuda.synchronize()
    got = _planes_to_f64(planes, x.size)
    assert np.array_equal(got, x.astype(np.float64)), 'three-plane split is not exact'
    # the planes are ordered by magnitude: |p2| <= 2^-8 |p1|, |p3| <= 2^-16 |p1| (up to rounding of the leading plane)
    b = (planes[:, :x.size].cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    nz = np.abs(b[0]) > 0
    assert np.all(np.abs(b[1][nz]) <= np.abs(b[0][nz]) * 2.0 ** -8 * 1.01)
    assert np.all(np.abs(b[2][nz]) <= np.abs(b[0][nz]) * 2.0 ** -16 * 1.01)
    # below 2^-110 the last bits of an fp32 value lie under bf16's smallest subnormal (2^-133): the split is then within 2^-133
    tiny = (rng.standard_normal(4096) * 1e-36).astype(np.float32)
    planes, ps = _split_linear(torch.from_numpy(tiny).to(dev))
    torch.cuda.synchronize()
    assert np.abs(_planes_to_f64(planes, tiny.size) - tiny.astype(np.float64)).max() <= 2.0 ** -133


def test_split_planes_equal_the_cpu_model_bit_for_bit(dev):
    """integer work: the kernel's three planes are the bit patterns of oracle/bf16x3_emul.split3 (round to nearest even on the fp32 bit
    pattern, exact residues) -- the bar is equality.  Values in the normal range of every residue (1e-20 .. 1e20, zeros, powers of two)."""
    import bf16x3_emul as em
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.standard_normal(65536 + 5).astype(np.float32), (rng.standard_normal(4096) * 1e-20).astype(np.float32),
                        (rng.standard_normal(4096) * 1e20).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -60, 2.0 ** 100, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0, 255.0, 257.0, np.inf, -np.inf], np.float32)])
    planes, ps = _split_linear(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    got = planes[:, :x.size].cpu().numpy().view(np.uint16)
    want, _ = em.split3(x)
    assert np.array_equal(got, want)


def test_chunk_major_split_is_the_permuted_plain_split(dev):
    """straps_split3_bf16_cm == straps_split3_bf16 followed by the chunk-major permutation (csrc/common.h cm_index: element (row, c) of a
    [rows][C] tensor at ((c / 32) * rows + row) * 32 + c % 32), restated here with torch views; odd row counts, C = 32 .. 512."""
    for rows, C in ((37, 32), (1000, 64), (129, 128), (77, 512), (4096, 256)):
        x = torch.from_numpy(det_uniform((rows, C), 70 + C, -3, 3)).to(dev)
        lin, ps = _split_linear(x)
        cm, ps2 = _split(x)
        torch.cuda.synchronize()
        assert ps == ps2
        for pl in range(3):
            want = _cm(lin[pl, :rows * C].view(rows, C))
            assert torch.equal(cm[pl, :rows * C], want), (rows, C, pl)


@pytest.mark.parametrize('with_fp32', [False, True])
def test_batched_weight_pack_writes_the_planes_of_both_layouts(dev, with_fp32):
    """straps_pack_conv_weights_batched_x3: the fp32 outputs equal straps_pack_conv_weights_batched's, and the planes are the split of
    the packed weights in CHUNK-MAJOR order -- forward planes: element (cout, tap, cin) at ((tap * Cin/32 + cin/32) * Cout + cout) * 32 +
    cin % 32; data-gradient planes: (cin, flipped tap, cout) at ((tap * Cout/32 + cout/32) * Cin + cin) * 32 + cout % 32 -- restated here
    with torch views, bit for bit (1x1 / 3x3 taps).  Layers whose reduction extent is not a multiple of 32 (they cannot run on the bf16x3
    route) get no planes: their slots stay untouched; with NULL fp32 destinations only the planes are written."""
    L = hipabi.lib()
    shapes = [(64, 64, 3, 3), (128, 64, 1, 1), (96, 160, 3, 3), (33, 70, 5, 5), (256, 128, 3, 3), (8, 8, 1, 1), (64, 96, 3, 3)]
    ws = [torch.from_numpy(det_uniform(sh, 140 + i, -1, 1)).to(dev) for i, sh in enumerate(shapes)]
    total = sum(w.numel() for w in ws)
    ps = (total + 7) // 8 * 8 + 8

    def table(krsc, crsk):
        descs = (hipabi.PackDesc * len(ws))()
        off = 0
        for d, w in zip(descs, ws):
            d.src = w.data_ptr()
            d.dst_krsc = krsc.data_ptr() + 4 * off if krsc is not None else None
            d.dst_crsk = crsk.data_ptr() + 4 * off if crsk is not None else None
            d.o, d.c, d.r, d.s, d.first = w.shape[0], w.shape[1], w.shape[2], w.shape[3], off
            off += w.numel()
        return torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)

    krsc, crsk = torch.empty(total, device=dev), torch.empty(total, device=dev)
    hipabi.check(L.straps_pack_conv_weights_batched(hipabi.ptr(table(krsc, crsk)), len(ws), total, None), 'batched pack')
    # expected planes: per layer the chunk-major permutation of the packed fp32 block, split; zero where no planes are written
    want_k, want_c = torch.zeros(3, ps, device=dev, dtype=torch.int16), torch.zeros(3, ps, device=dev, dtype=torch.int16)
    off = 0
    for w in ws:
        O, C, R, S = w.shape
        n, RS = w.numel(), R * S
        if C % 32 == 0:
            blk = krsc[off:off + n].view(O, RS, C // 32, 32).permute(1, 2, 0, 3).contiguous().reshape(-1)
            pl, _ = _split_linear(blk)
            want_k[:, off:off + n] = pl[:, :n]
        if O % 32 == 0:
            blk = crsk[off:off + n].view(C, RS, O // 32, 32).permute(1, 2, 0, 3).contiguous().reshape(-1)
            pl, _ = _split_linear(blk)
            want_c[:, off:off + n] = pl[:, :n]
        off += n
    k2 = torch.full((total,), float('nan'), device=dev) if with_fp32 else None
    c2 = torch.full((total,), float('nan'), device=dev) if with_fp32 else None
    k3 = torch.zeros(3, ps, device=dev, dtype=torch.int16)
    c3 = torch.zeros(3, ps, device=dev, dtype=torch.int16)
    hipabi.check(L.straps_pack_conv_weights_batched_x3(hipabi.ptr(table(k2, c2)), len(ws), total, hipabi.ptr(k3), hipabi.ptr(c3), ps, None), 'batched pack x3')
    torch.cuda.synchronize()
    assert torch.equal(k3, want_k) and torch.equal(c3, want_c)
    if with_fp32:
        assert torch.equal(k2, krsc) and torch.equal(c2, crsk)
    # forward-only form: no data-gradient planes
    k3b = torch.zeros(3, ps, device=dev, dtype=torch.int16)
    hipabi.check(L.straps_pack_conv_weights_batched_x3(hipabi.ptr(table(None, None)), len(ws), total, hipabi.ptr(k3b), None, ps, None), 'batched pack x3 fwd')
    torch.cuda.synchronize()
    assert torch.equal(k3b, want_k)
    assert L.straps_pack_conv_weights_batched_x3(hipabi.ptr(table(None, None)), len(ws), total, None, None, ps, None) != 0


def _pack(dev, w, dgrad=False):
    L = hipabi.lib()
    Cout, Cin, k, _ = w.shape
    wd = w.float().contiguous().to(dev)
    wp = torch.empty_like(wd)
    fn = L.straps_pack_conv_weight_dgrad if dgrad else L.straps_pack_conv_weight
    hipabi.check(fn(hipabi.ptr(wd), hipabi.ptr(wp), Cout, Cin, k, k, None), 'pack')
    return wp


def _fwd(dev, x_nchw, w, stride, pad, cfg, x3=True, scale=None, shift=None, res_nchw=None, relu=False, stats=False):
    L = hipabi.lib()
    B, Cin, H, W = x_nchw.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = x_nchw.float().permute(0, 2, 3, 1).contiguous().to(dev)
    wp = _pack(dev, w)
    y = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
    res = res_nchw.float().permute(0, 2, 3, 1).contiguous().to(dev) if res_nchw is not None else None
    sc = scale.float().to(dev) if scale is not None else None
    sh = shift.float().to(dev) if shift is not None else None
    part = None
    if x3:
        if stats:
            part = torch.empty(L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, k, k, stride, pad, cfg), Cout, 2, device=dev)
        xp, xps = _split(x)
        wp3, wps = _wsplit(dev, w)
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(xp), xps, hipabi.ptr(wp3), wps, hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), int(relu),
                                          hipabi.ptr(y), hipabi.ptr(part), B, H, W, Cin, Cout, k, k, stride, pad, cfg, None), 'conv_fwd_x3')
    else:
        hipabi.check(L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), int(relu), hipabi.ptr(y),
                                       None, B, H, W, Cin, Cout, k, k, stride, pad, 0, None), 'conv_fwd')
    torch.cuda.synchronize()
    return y.permute(0, 3, 1, 2).cpu().double(), part


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,cfg', [
    (2, 64, 64, 16, 16, 3, 1, 0), (2, 64, 64, 16, 16, 3, 1, 3), (2, 64, 64, 16, 16, 3, 1, 7), (2, 64, 128, 16, 16, 3, 2, 1),
    (3, 128, 128, 8, 8, 3, 1, 5), (5, 128, 128, 24, 24, 3, 1, 4), (5, 128, 128, 24, 24, 3, 1, 6), (2, 64, 128, 16, 16, 1, 2, 0),
    (1, 256, 512, 8, 8, 3, 2, 0), (1, 512, 512, 8, 8, 3, 1, 0), (5, 64, 64, 7, 7, 3, 1, 2), (2, 256, 64, 16, 16, 1, 1, 0),
    (2, 64, 64, 10, 24, 3, 1, 0), (3, 96, 128, 7, 13, 3, 2, 0), (1, 32, 64, 3, 3, 3, 1, 0), (2, 2048, 512, 4, 4, 1, 1, 0),
    # shapes the halo-patch kernel takes (cfg 0): two rows of 64, four rows of 32, eight rows of 16, two whole 8x8 images, 128-row
    # tiles inside taller maps, 64 / 128 output channels (cfg 512 = halo kernel wherever it applies) -- the same through the im2col
    # kernel (cfg 256) -- and a full-size layer3 launch, which the auto rule gives to the halo kernel
    (1, 64, 64, 64, 64, 3, 1, 512), (1, 128, 128, 32, 32, 3, 1, 512), (2, 256, 256, 16, 16, 3, 1, 512), (4, 512, 512, 8, 8, 3, 1, 512),
    (2, 64, 128, 8, 16, 3, 1, 512), (1, 64, 64, 8, 32, 3, 1, 512), (2, 64, 64, 16, 16, 3, 1, 256), (4, 512, 512, 8, 8, 3, 1, 256),
    (64, 256, 256, 16, 16, 3, 1, 0),
    # the single-patch-buffer halo kernel (the auto rule for 64-channel outputs; cfg 1024 = explicitly): four channel chunks = three
    # patch reloads at chunk boundaries, whole-image tiles
    (2, 128, 64, 16, 16, 3, 1, 0), (4, 64, 64, 8, 8, 3, 1, 1024),
    # software-pipelined chunk loop (barrier between the two k steps, register double buffer): every such configuration
    (3, 128, 128, 8, 8, 3, 1, 8), (5, 128, 128, 24, 24, 3, 1, 9), (2, 64, 64, 16, 16, 3, 1, 10), (5, 64, 64, 7, 7, 3, 1, 11),
    (5, 128, 128, 24, 24, 3, 1, 12), (2, 64, 128, 16, 16, 1, 2, 8), (1, 32, 64, 3, 3, 3, 1, 11), (2, 64, 128, 16, 16, 3, 2, 9),
    # odd-sized maps through the chunk-major addressing: rows that are not a multiple of anything, stride 2 (64-byte pieces at a 128-byte stride)
    (3, 64, 64, 5, 9, 3, 1, 0), (1, 160, 64, 11, 7, 3, 2, 0)])
def test_conv_fwd_x3_vs_float64(dev, B, Cin, Cout, H, W, k, stride, cfg):
    """all tile configurations (4- and 8-wave, 2- and 3-stage rings), ragged M, stride 1 / 2, 3x3 and 1x1, non-square maps, the
    fused epilogue and the training-mode statistics; the bar is the exact-fp32 kernel's (test_gpu_forward.py)."""
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 1, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 2, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    y, part = _fwd(dev, x, w, stride, pad, cfg, stats=True)
    err = (y - ref).abs()
    assert float((err - (2e-5 + 2e-5 * ref.abs())).max()) <= 0, 'raw conv: max abs err %.3e' % float(err.max())
    y32, _ = _fwd(dev, x, w, stride, pad, 0, x3=False)
    e32 = float((y32 - ref).abs().max())
    assert float(err.max()) <= 3.0 * e32 + 1e-7, 'bf16x3 error %.3e vs exact-fp32 chain %.3e' % (float(err.max()), e32)
    s = part.double().sum(0).cpu()
    np.testing.assert_allclose(s[:, 0].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    sc = torch.from_numpy(det_uniform((Cout,), 3, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((Cout,), 4, -0.5, 0.5))
    res = torch.from_numpy(det_uniform(tuple(ref.shape), 5, -1, 1))
    want = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res.double())
    y2, _ = _fwd(dev, x, w, stride, pad, cfg, scale=sc, shift=sh, res_nchw=res, relu=True)
    err2 = (y2 - want).abs()
    assert float((err2 - (3e-5 + 3e-5 * want.abs())).max()) <= 0, 'fused epilogue: max abs err %.3e' % float(err2.max())


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,cfg', [
    (2, 64, 64, 16, 16, 3, 1, 0), (2, 64, 128, 16, 16, 3, 2, 0), (3, 128, 64, 9, 9, 3, 1, 0), (2, 64, 128, 16, 16, 1, 2, 0),
    (1, 256, 512, 8, 8, 3, 2, 0), (2, 256, 64, 8, 8, 1, 1, 0), (2, 128, 128, 15, 15, 3, 2, 1), (4, 128, 128, 20, 12, 3, 1, 4),
    (2, 128, 256, 9, 14, 3, 2, 5), (2, 64, 64, 10, 24, 3, 1, 3),
    # the stride-2 rule of round 3 (tiles per parity class): >= 512 -> 256x128; 256..511 -> 128x128 8-wave (1x1) / 256x128 (3x3); the 8x8 case above -> 128x64
    (2, 256, 64, 256, 256, 3, 2, 0), (2, 512, 64, 128, 128, 1, 2, 0), (2, 512, 64, 128, 128, 3, 2, 0)])
def test_conv_dgrad_x3_vs_float64(dev, B, Cin, Cout, H, W, k, stride, cfg):
    """data gradient on the bf16x3 route (stride-2 parity classes, odd sizes, the skip-gradient addend) vs float64 autograd."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 1, -1, 1)).double().requires_grad_()
    w = (torch.from_numpy(det_uniform((Cout, Cin, k, k), 2, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5).double()
    y = F.conv2d(x, w, None, stride, pad)
    dy = torch.from_numpy(det_uniform(tuple(y.shape), 3, -1, 1)).double() * 1e-3          # gradient-sized values
    y.backward(dy)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    g3, gps = _split(dyd)
    w3, wps = _wsplit(dev, w, dgrad=True)
    add = torch.from_numpy(det_uniform((B, H, W, Cin), 4, -1, 1)).to(dev) * 1e-3
    dx = torch.full((B, H, W, Cin), float('nan'), device=dev)
    hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(add), hipabi.ptr(dx), B, H, W, Cin, Cout, k, k,
                                        stride, pad, cfg, None), 'dgrad_x3')
    want = x.grad.permute(0, 2, 3, 1) + add.cpu().double()
    err = float((dx.cpu().double() - want).abs().max() / want.abs().max())
    assert err < 2e-5, 'dgrad_x3 relative-to-max error %.3e' % err


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride', [
    (2, 64, 64, 16, 16, 3, 1), (3, 128, 64, 8, 8, 3, 1), (2, 64, 128, 4, 32, 3, 1), (1, 128, 128, 6, 64, 3, 1), (5, 64, 64, 12, 8, 3, 1),
    # shapes outside the halo plan (odd row count, non-power-of-two width), stride 2, 1x1 (stride 1 and 2), the 128x128 channel tile
    # of the big 1x1 layers (M >= 8192), ragged last step, zero-padded borders: the per-tap kernel on the planes
    (2, 64, 64, 9, 16, 3, 1), (2, 64, 64, 10, 24, 3, 1), (2, 64, 128, 16, 16, 3, 2), (2, 128, 128, 15, 15, 3, 2), (2, 64, 128, 16, 16, 1, 2),
    (2, 256, 64, 8, 8, 1, 1), (3, 128, 256, 56, 56, 1, 1), (1, 256, 128, 96, 96, 1, 1), (1, 256, 512, 8, 8, 3, 2), (3, 64, 192, 7, 5, 1, 1),
    # the rectangular channel blocks of round 4 (csrc/backward.hip wgrad_x3_block): 256 x 128 (1x1 with >= 32 768 pixels; 3x3 / stride 2 above),
    # 64 x 128 and 128 x 64 (1x1 with <= 8 192 pixels, by the larger side / at stride 2), 128 x 64 (3x3 / stride 2 with 128 output channels above)
    (2, 128, 256, 128, 128, 1, 1), (2, 128, 256, 32, 32, 1, 1), (2, 256, 128, 32, 32, 1, 1), (2, 128, 256, 64, 64, 1, 2), (3, 384, 256, 20, 20, 1, 1)])
def test_conv_wgrad_x3_vs_float64(dev, B, Cin, Cout, H, W, k, stride):
    """weight gradient on the planes: the halo-patch kernel (3x3 / stride 1, every chunk geometry the plan produces: W = 8 .. 64) and
    the per-tap kernel (everything else), transpose-read operand gathers, several splits, accumulate.  Bar: 2e-5 of the maximum,
    the fp32 kernels' bar in test_gpu_backward.py."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 31, -1, 1)).double()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.from_numpy(det_uniform((B, Cout, Ho, Wo), 32, -1, 1)).double() * 1e-3
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, stride=stride, padding=pad)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
    gd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
    x3, xps = _split(xd)
    g3, gps = _split(gd)
    ws = torch.empty(L.straps_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, k, k, stride, pad) // 4, device=dev)
    dw = torch.full((Cout, Cin, k, k), float('nan'), device=dev)
    args = (hipabi.ptr(xd), hipabi.ptr(gd), hipabi.ptr(x3), xps, hipabi.ptr(g3), gps, hipabi.ptr(dw), hipabi.ptr(ws), B, H, W, Cin, Cout, k, k, stride, pad)
    hipabi.check(L.straps_conv_wgrad_x3(*args, 0, None), 'wgrad_x3')
    err = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, 'wgrad_x3 relative-to-max error %.3e' % err
    hipabi.check(L.straps_conv_wgrad_x3(*args, 1, None), 'wgrad_x3 accumulate')
    err = float((dw.cpu().double() - 2 * ref).abs().max() / ref.abs().max())
    assert err < 4e-5, 'wgrad_x3 accumulate: %.3e' % err
    # without planes the entry point is the fp32 weight gradient
    hipabi.check(L.straps_conv_wgrad_x3(hipabi.ptr(xd), hipabi.ptr(gd), None, 0, None, 0, hipabi.ptr(dw), hipabi.ptr(ws), B, H, W, Cin, Cout, k, k,
                                        stride, pad, 0, None), 'wgrad_x3 (no planes)')
    err = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,cfg,from_out', [
    (2, 64, 64, 16, 16, 3, 1, 0, False), (2, 64, 128, 16, 16, 3, 2, 0, True), (64, 256, 256, 16, 16, 3, 1, 0, False), (3, 128, 128, 8, 8, 3, 1, 5, True),
    (2, 64, 128, 16, 16, 1, 2, 0, False), (5, 128, 128, 24, 24, 3, 1, 12, True), (2, 128, 256, 9, 14, 3, 2, 1, False), (4, 128, 64, 32, 32, 3, 1, 512, False),
    # ragged last tiles of a stride-2 gradient's parity classes on the AUTOMATIC tiles (the lean data-gradient epilogue, round 6), odd maps
    (2, 128, 256, 9, 14, 3, 2, 0, False), (3, 64, 128, 15, 11, 3, 2, 0, False), (2, 64, 128, 30, 22, 3, 2, 0, False), (5, 64, 64, 7, 7, 3, 1, 0, False)])
def test_dgrad_x3_with_fused_batchnorm_sums(dev, B, Cin, Cout, H, W, k, stride, cfg, from_out):
    """straps_conv_dgrad_x3_bn: dx as straps_conv_dgrad_x3 writes it (bit for bit), and the per-tile partials of the next BatchNorm
    backward's sums S1 = sum mask*dy, S2 = invstd * sum mask*dy*(raw - mean) (double) -- every tile configuration in use, the four parity
    classes of a stride-2 gradient, the halo-patch kernel, both mask sources -- against a float64 evaluation on the kernel's own dx; and
    straps_bn_bwd_finish_x3 on those partials against straps_bn_bwd_x3 (which sums in its own pass)."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 2, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    g = torch.from_numpy(det_uniform((B, Ho, Wo, Cout), 3, -1, 1)).to(dev) * 1e-3
    g3, gps = _split(g)
    w3, wps = _wsplit(dev, w, dgrad=True)
    # (the addend and raw sit in the middle of NaN-filled buffers: an operand read past a tensor's end -- a ragged tile's rows behind the last pixel -- shows
    #  up as a NaN in the sums instead of depending on what the allocator left there)
    n_el = B * H * W * Cin
    poison_a, poison_r = torch.full((3 * n_el,), float('nan'), device=dev), torch.full((3 * n_el,), float('nan'), device=dev)
    add, raw = poison_a[n_el:2 * n_el].view(B, H, W, Cin), poison_r[n_el:2 * n_el].view(B, H, W, Cin)
    add.copy_(torch.from_numpy(det_uniform((B, H, W, Cin), 4, -1, 1)).to(dev) * 1e-3)
    raw.copy_(torch.from_numpy(det_uniform((B, H, W, Cin), 5, -2, 2)).to(dev))
    mean = raw.mean(dim=(0, 1, 2)).contiguous()
    invstd = (raw.var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt().contiguous()
    msc = torch.from_numpy(det_uniform((Cin,), 6, 0.5, 1.5)).to(dev)
    msh = torch.from_numpy(det_uniform((Cin,), 7, -0.5, 0.5)).to(dev)
    out = torch.relu(raw * msc + msh + torch.from_numpy(det_uniform((B, H, W, Cin), 8, -1, 1)).to(dev)).contiguous()      # (a residual unit's output)
    dx0 = torch.full((B, H, W, Cin), float('nan'), device=dev)
    dx1 = torch.full((B, H, W, Cin), float('nan'), device=dev)
    geo = (B, H, W, Cin, Cout, k, k, stride, pad, cfg)
    hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(add), hipabi.ptr(dx0), *geo, None), 'dgrad_x3')
    nblk = L.straps_conv_dgrad_x3_bn_blocks(*geo)
    assert nblk > 0
    part = torch.full((nblk, Cin, 2), float('nan'), device=dev, dtype=torch.float64)
    hipabi.check(L.straps_conv_dgrad_x3_bn(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(add), hipabi.ptr(dx1), *geo, hipabi.ptr(raw),
                                           hipabi.ptr(out if from_out else None), hipabi.ptr(None if from_out else msc), hipabi.ptr(None if from_out else msh),
                                           hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(part), None), 'dgrad_x3_bn')
    assert torch.equal(dx0, dx1)
    # mask = fma(raw, scale, shift) > 0 in the kernel; evaluated here in float64 and rounded once (= the fused fp32 operation except
    # for values within half an ulp of a tie -- none with these inputs)
    mask = (out > 0) if from_out else ((raw.double() * msc.double() + msh.double()).float() > 0)
    gd = torch.where(mask, dx1, torch.zeros_like(dx1)).double()
    s1 = gd.sum(dim=(0, 1, 2))
    s2 = (gd * (raw.double() - mean.double())).sum(dim=(0, 1, 2)) * invstd.double()
    got = part.sum(0)
    scale1, scale2 = float(gd.abs().sum(dim=(0, 1, 2)).max()), float((gd * (raw.double() - mean.double())).abs().sum(dim=(0, 1, 2)).max() * invstd.max())
    assert float((got[:, 0] - s1).abs().max()) <= 1e-9 * scale1 + 1e-30, 'S1'
    assert float((got[:, 1] - s2).abs().max()) <= 1e-9 * scale2 + 1e-30, 'S2'
    # finish = finalize + apply on the partials == the three-pass entry point on the same dy
    rows = B * H * W
    gamma = torch.from_numpy(det_uniform((Cin,), 9, 0.5, 1.5)).to(dev)
    ws = torch.empty(L.straps_bn_bwd_workspace_bytes(rows, Cin) // 4, device=dev)
    outs = []
    for fused in (False, True):
        dg, db, draw, dz = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev), torch.empty_like(raw), torch.empty_like(raw)
        common = (hipabi.ptr(dx1), hipabi.ptr(out if from_out else None), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma),
                  hipabi.ptr(None if from_out else msc), hipabi.ptr(None if from_out else msh), hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(dz),
                  None, 0)
        if fused:
            hipabi.check(L.straps_bn_bwd_finish_x3(*common, hipabi.ptr(part), nblk, hipabi.ptr(ws), rows, Cin, 0, None), 'bn_bwd_finish_x3')
        else:
            hipabi.check(L.straps_bn_bwd_x3(*common, hipabi.ptr(ws), rows, Cin, 0, None), 'bn_bwd_x3')
        outs.append((dg, db, draw, dz))
    for a, b, name in zip(outs[0], outs[1], ('dgamma', 'dbeta', 'draw', 'dz')):
        err = float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
        assert err < 2e-6, (name, err)          # (the same double sums in another order, rounded once to fp32)


def _pack_relu_bits(y):
    """[rows..., C] activation -> int32 words [rows][C / 32], bit (c & 31) of word [row][c / 32] = (y > 0) (include/straps_hip.h, ABI 8)"""
    C = y.shape[-1]
    b = (y.reshape(-1, C // 32, 32) > 0).to(torch.int64)
    wd = (b << torch.arange(32, device=y.device, dtype=torch.int64)).sum(-1)
    return torch.where(wd >= 2 ** 31, wd - 2 ** 32, wd).to(torch.int32).contiguous()


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,cfg', [
    (4, 64, 64, 16, 16, 3, 1, 0), (3, 128, 64, 12, 20, 1, 1, 1), (2, 256, 64, 16, 16, 1, 1, 0), (2, 64, 128, 16, 16, 1, 2, 0), (2, 128, 256, 9, 14, 3, 2, 1),
    (4, 128, 64, 32, 32, 3, 1, 512), (2, 512, 128, 8, 8, 1, 1, 3), (1, 2048, 512, 8, 8, 1, 1, 0),
    # round 5: the bit forms (and every data gradient with an addend) run the LOOK-AHEAD epilogue of csrc/conv_igemm.h, the fp32-mask form of the
    # fused sums still runs the row-by-row epilogue of rounds 1-4 -- this test is their bit-for-bit comparison, so every tile configuration the
    # rule can choose is listed: 256x128 and 128x128 eight-wave tiles (two of four / two units in flight), four-wave 128x128 (four units), 128x64
    # pipelined and plain, 64x64, both halo-patch kernels, ragged last tiles (row-by-row inside a look-ahead launch), stride-2 parity classes
    (5, 128, 128, 24, 24, 3, 1, 12), (4, 128, 128, 16, 16, 3, 1, 12), (3, 128, 128, 8, 8, 3, 1, 5), (5, 128, 128, 24, 24, 3, 1, 9), (5, 64, 64, 7, 7, 3, 1, 11),
    (4, 64, 64, 16, 16, 3, 1, 11), (2, 64, 64, 16, 16, 3, 1, 7), (4, 64, 64, 8, 8, 3, 1, 1024), (2, 128, 64, 16, 16, 3, 2, 12), (2, 256, 256, 16, 16, 3, 1, 512),
    (2, 128, 128, 16, 16, 3, 1, 4), (3, 128, 192, 12, 12, 1, 1, 8), (2, 64, 128, 32, 32, 3, 2, 0)])
def test_relu_bits_forms_equal_the_fp32_mask_forms_bit_for_bit(dev, B, Cin, Cout, H, W, k, stride, cfg):
    """ABI 8: a residual unit's ReLU decisions as bits.  straps_bn_apply_bits_x3 writes y / planes as straps_bn_apply_x3 and the words
    bit (c & 31) of [row][c / 32] = (y > 0); every *_bits backward form -- data gradient with a masked addend (plain and with the fused
    BatchNorm sums masked by bits), BatchNorm backward in three passes and on fused partials -- equals, bit for bit, the fp32 form fed with the
    activation itself and with the masked gradient dz the fp32 route materialises.  Ragged M tiles, stride-2 parity classes, halo-patch
    kernel, 64 ... 2048 channels (both loop forms of the streaming kernels)."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    rows = B * H * W
    raw = torch.from_numpy(det_uniform((B, H, W, Cin), 5, -2, 2)).to(dev)
    res = torch.from_numpy(det_uniform((B, H, W, Cin), 8, -1, 1)).to(dev)
    res[0, 0, :, : Cin // 2] = 0.0
    raw[0, 0, :, : Cin // 2] = 0.0          # exact zeros after the ReLU's input: y == 0 must read as "off"
    sc = torch.from_numpy(det_uniform((Cin,), 6, 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(det_uniform((Cin,), 7, -0.5, 0.5)).to(dev)
    sh[: Cin // 4] = 0.0
    ps = (rows * Cin + 7) // 8 * 8
    # ---- forward: y, planes, bits
    y0, y1 = torch.empty_like(raw), torch.empty_like(raw)
    p0, p1 = torch.zeros(3, ps, device=dev, dtype=torch.int16), torch.zeros(3, ps, device=dev, dtype=torch.int16)
    bits = torch.full((rows, Cin // 32), 0x5a5a5a5a, device=dev, dtype=torch.int32)
    hipabi.check(L.straps_bn_apply_x3(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, hipabi.ptr(y0), hipabi.ptr(p0), ps, rows, Cin, None),
                 'bn_apply_x3')
    hipabi.check(L.straps_bn_apply_bits_x3(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), hipabi.ptr(y1), hipabi.ptr(p1), ps, hipabi.ptr(bits),
                                           rows, Cin, None), 'bn_apply_bits_x3')
    assert torch.equal(y0, y1) and torch.equal(p0, p1)
    assert torch.equal(bits, _pack_relu_bits(y0))
    assert 0.2 < float((y0 > 0).float().mean()) < 0.8 and bool((y0[0, 0, :, : Cin // 4] == 0).all())
    # ---- the gradient arriving at the unit's output, its masked copy (what the fp32 route writes as dz), the convolution's operands
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 2, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    g = torch.from_numpy(det_uniform((B, Ho, Wo, Cout), 3, -1, 1)).to(dev) * 1e-3
    g3, gps = _split(g)
    w3, wps = _wsplit(dev, w, dgrad=True)
    dy_next = torch.from_numpy(det_uniform((B, H, W, Cin), 4, -1, 1)).to(dev) * 1e-3      # unmasked gradient of the LATER unit's output ...
    y_next = torch.relu(torch.from_numpy(det_uniform((B, H, W, Cin), 9, -1, 1)).to(dev)).contiguous()      # ... and that unit's activation
    bits_next = _pack_relu_bits(y_next)
    dz_next = torch.where(y_next > 0, dy_next, torch.zeros_like(dy_next)).contiguous()
    mean = raw.mean(dim=(0, 1, 2)).contiguous()
    invstd = (raw.var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt().contiguous()
    geo = (B, H, W, Cin, Cout, k, k, stride, pad, cfg)
    # ---- plain data gradient with a masked addend
    dxa, dxb = torch.full_like(raw, float('nan')), torch.full_like(raw, float('nan'))
    hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(dz_next), hipabi.ptr(dxa), *geo, None), 'dgrad_x3')
    hipabi.check(L.straps_conv_dgrad_x3_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(dy_next), hipabi.ptr(dxb), *geo, hipabi.ptr(bits_next), None),
                 'dgrad_x3_bits')
    assert torch.equal(dxa, dxb)
    # ---- data gradient + the sums of THIS unit's last BatchNorm: mask = y0 > 0 as fp32 tensor / as bits; addend masked / unmasked + bits
    nblk = L.straps_conv_dgrad_x3_bn_blocks(*geo)
    pa = torch.full((nblk, Cin, 2), float('nan'), device=dev, dtype=torch.float64)
    pb = torch.full_like(pa, float('nan'))
    dxa.fill_(float('nan')); dxb.fill_(float('nan'))
    hipabi.check(L.straps_conv_dgrad_x3_bn(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(dz_next), hipabi.ptr(dxa), *geo, hipabi.ptr(raw), hipabi.ptr(y0),
                                           None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(pa), None), 'dgrad_x3_bn')
    hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(dy_next), hipabi.ptr(dxb), *geo, hipabi.ptr(raw), None,
                                                None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(pb), hipabi.ptr(bits_next), hipabi.ptr(bits), None),
                 'dgrad_x3_bn_bits')
    assert torch.equal(dxa, dxb) and torch.equal(pa, pb)
    # (one operand as bits only: the other as in the plain form)
    dxc, pc = torch.full_like(raw, float('nan')), torch.full_like(pa, float('nan'))
    hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(dz_next), hipabi.ptr(dxc), *geo, hipabi.ptr(raw), None,
                                                None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(pc), None, hipabi.ptr(bits), None), 'dgrad_x3_bn_bits')
    assert torch.equal(dxa, dxc) and torch.equal(pa, pc)
    # ---- BatchNorm backward: three passes and finish-on-partials, planes + fp32
    gamma = torch.from_numpy(det_uniform((Cin,), 10, 0.5, 1.5)).to(dev)
    ws = torch.empty(L.straps_bn_bwd_workspace_bytes(rows, Cin) // 4, device=dev)
    res_ = {}
    for form in ('fp32', 'bits', 'fp32_finish', 'bits_finish'):
        dg, db, draw = torch.full((Cin,), float('nan'), device=dev), torch.full((Cin,), float('nan'), device=dev), torch.full_like(raw, float('nan'))
        dz = torch.full_like(raw, float('nan'))
        pl = torch.zeros(3, ps, device=dev, dtype=torch.int16)
        if form == 'fp32':
            hipabi.check(L.straps_bn_bwd_x3(hipabi.ptr(dxa), hipabi.ptr(y0), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma), None, None,
                                            hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(dz), hipabi.ptr(pl), ps, hipabi.ptr(ws), rows, Cin, 0, None), form)
        elif form == 'bits':
            hipabi.check(L.straps_bn_bwd_bits_x3(hipabi.ptr(dxa), hipabi.ptr(bits), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma),
                                                 hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(pl), ps, hipabi.ptr(ws), rows, Cin, 0, None), form)
        elif form == 'fp32_finish':
            hipabi.check(L.straps_bn_bwd_finish_x3(hipabi.ptr(dxa), hipabi.ptr(y0), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma), None,
                                                   None, hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(dz), hipabi.ptr(pl), ps, hipabi.ptr(pa), nblk,
                                                   hipabi.ptr(ws), rows, Cin, 0, None), form)
        else:
            hipabi.check(L.straps_bn_bwd_finish_bits_x3(hipabi.ptr(dxa), hipabi.ptr(bits), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma),
                                                        hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(pl), ps, hipabi.ptr(pa), nblk, hipabi.ptr(ws),
                                                        rows, Cin, 0, None), form)
        res_[form] = (dg, db, draw, pl)
        if form.startswith('fp32'):      # the masked gradient the fp32 forms write is what the bits forms' consumers derive from (dy, bits)
            assert torch.equal(dz, torch.where(y0 > 0, dxa, torch.zeros_like(dxa)))
    for a, b in (('fp32', 'bits'), ('fp32_finish', 'bits_finish')):
        for t0, t1, name in zip(res_[a], res_[b], ('dgamma', 'dbeta', 'draw', 'planes')):
            assert torch.equal(t0, t1), (a, b, name)
    # the *_bits forms refuse what they cannot index
    assert L.straps_bn_bwd_bits_x3(hipabi.ptr(dxa), None, hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma), hipabi.ptr(dg), hipabi.ptr(db),
                                   hipabi.ptr(draw), None, 0, hipabi.ptr(ws), rows, Cin, 0, None) == 1      # STRAPS_EINVAL
    assert L.straps_conv_dgrad_x3_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, None, hipabi.ptr(dxb), *geo, hipabi.ptr(bits_next), None) == 1


@pytest.mark.parametrize('B,Cin,Cout', [(64, 64, 64), (16, 128, 64)])
def test_single_patch_buffer_halo_kernel_at_layer1_size_forward_dgrad_and_repeats(dev, B, Cin, Cout):
    """conv_igemm_x3h_kernel<PBUF = 1> (the auto rule of every 64-channel 3x3 / stride-1 layer: one patch buffer re-copied at the channel-chunk
    boundary while the co-resident workgroup computes) at the REAL layer1 map size 64 x 64 -- 2 048 / 512 tiles, two workgroups per CU, one and
    three patch reloads per tile: forward and data gradient against float64, and 50 repeated launches on one input bit-identical (a stale
    patch read -- the failure of an earlier form of this kernel -- would show as a launch-to-launch difference or a column of wrong values)."""
    L = hipabi.lib()
    H = W = 64
    k, pad = 3, 1
    assert L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, k, k, 1, pad, 0) == B * H * W // 128
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 51, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 52, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    torch.set_num_threads(min(32, __import__('os').cpu_count() or 8))
    ref = F.conv2d(x.double(), w.double(), None, 1, pad)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    x3, xps = _split(xd)
    w3, wps = _wsplit(dev, w)
    first = None
    for rep in range(50):
        y = torch.full((B, H, W, Cout), float('nan'), device=dev)
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y), None, B, H, W, Cin, Cout, k, k, 1, pad,
                                          0, None), 'conv_fwd_x3')
        if first is None:
            first = y
        else:
            assert torch.equal(first, y), 'forward launch %d differs from launch 0' % rep
    err = (first.permute(0, 3, 1, 2).cpu().double() - ref).abs()
    assert float((err - (2e-5 + 2e-5 * ref.abs())).max()) <= 0, 'forward: max abs err %.3e' % float(err.max())
    # the im2col kernel on the same operands (tile_cfg bit 8) agrees to rounding: the two kernels sum the same products in another order
    y_i = torch.empty_like(first)
    hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y_i), None, B, H, W, Cin, Cout, k, k, 1, pad,
                                      256, None), 'conv_fwd_x3 im2col')
    assert float((first - y_i).abs().max()) < 2e-5
    del ref, err
    # data gradient of a Cout -> Cin convolution whose input has `Cout` channels: the GEMM has Cout output columns again
    # (dy [B,H,W,Cin_of_gemm = Cin] -> dx [B,H,W,Cout]), i.e. the same kernel instantiation with the flipped taps
    wd_full = torch.from_numpy(det_uniform((Cin, Cout, k, k), 53, -1, 1)) * (2.0 / (Cout * k * k)) ** 0.5      # conv: Cout channels in, Cin out
    dy = torch.from_numpy(det_uniform((B, Cin, H, W), 54, -1, 1)) * 1e-3
    want = F.conv_transpose2d(dy.double(), wd_full.double(), None, 1, pad)                                        # [B,Cout,H,W]
    g3, gps = _split(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    wd3, wdps = _wsplit(dev, wd_full, dgrad=True)
    add = torch.from_numpy(det_uniform((B, H, W, Cout), 55, -1, 1)).to(dev) * 1e-3
    first = None
    for rep in range(50):
        dx = torch.full((B, H, W, Cout), float('nan'), device=dev)
        hipabi.check(L.straps_conv_dgrad_x3(hipabi.ptr(g3), gps, hipabi.ptr(wd3), wdps, hipabi.ptr(add), hipabi.ptr(dx), B, H, W, Cout, Cin, k, k, 1, pad, 0,
                                            None), 'dgrad_x3')
        if first is None:
            first = dx
        else:
            assert torch.equal(first, dx), 'data-gradient launch %d differs from launch 0' % rep
    wantn = want.permute(0, 2, 3, 1) + add.cpu().double()
    e = float((first.cpu().double() - wantn).abs().max() / wantn.abs().max())
    assert e < 2e-5, 'dgrad relative-to-max error %.3e' % e


def test_error_budget_of_the_six_products(dev):
    """long reductions (K = 4608, layer4) of same-sign terms -- where a systematic bias of the dropped low-order products would
    show -- stay at the exact-fp32 chain's error against float64, and so do operands spread over 12 orders of magnitude."""
    B, Cin, Cout, H, k = 2, 512, 512, 8, 3
    x = torch.from_numpy(det_uniform((B, Cin, H, H), 11, 0.0, 1.0))                       # all positive: errors cannot cancel
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 12, 0.0, 1.0)) * (2.0 / (Cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    y3, _ = _fwd(dev, x, w, 1, 1, 0)
    y32, _ = _fwd(dev, x, w, 1, 1, 0, x3=False)
    r3 = float(((y3 - ref).abs() / ref.abs()).max())
    r32 = float(((y32 - ref).abs() / ref.abs()).max())
    # (fp32 accumulation of 4608 same-sign terms: the exact-fp32 chain itself sits at ~4e-6 here)
    assert r3 < 1.5 * r32 + 1e-7 and r3 < 1e-5, 'relative error %.3e of a same-sign K=4608 reduction (exact-fp32 chain: %.3e)' % (r3, r32)
    scale = torch.from_numpy(10.0 ** det_uniform((1, Cin, 1, 1), 13, -6, 6)).float()     # per-channel magnitudes 1e-6 .. 1e6
    xs, ws = x * scale, w / scale
    refs = F.conv2d(xs.double(), ws.double(), None, 1, 1)
    ys, _ = _fwd(dev, xs, ws, 1, 1, 0)
    ys32, _ = _fwd(dev, xs, ws, 1, 1, 0, x3=False)
    rs = float(((ys - refs).abs() / refs.abs()).max())
    rs32 = float(((ys32 - refs).abs() / refs.abs()).max())
    assert rs < 1.5 * rs32 + 1e-7 and rs < 1e-5, ('relative error %.3e with operand magnitudes over 12 decades (exact-fp32 chain: %.3e; no scaling is '
                                                   'involved: bf16 has the fp32 exponent)' % (rs, rs32))
    print('same-sign K=4608: bf16x3 %.2e  fp32 chain %.2e | 12 decades: bf16x3 %.2e  fp32 chain %.2e' % (r3, r32, rs, rs32))


@pytest.mark.parametrize('B,Cin,Cout,H,k,stride,cfg', [(2, 64, 64, 16, 3, 1, 0), (3, 128, 128, 8, 3, 1, 5), (2, 64, 128, 16, 3, 2, 12), (64, 256, 256, 16, 3, 1, 0), (5, 64, 64, 7, 3, 1, 2)])
def test_conv_fwd_x3p_planes_equal_a_split_pass(dev, B, Cin, Cout, H, k, stride, cfg):
    """straps_conv_fwd_x3p (inference chains): the fp32 result equals straps_conv_fwd_x3's bit for bit and the planes written by the epilogue
    equal a straps_split3_bf16 pass over it; with y = NULL the planes alone."""
    L = hipabi.lib()
    pad = 1
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.from_numpy(det_uniform((B, H, H, Cin), 41, -1, 1)).to(dev)
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 42, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    sc = torch.from_numpy(det_uniform((Cout,), 43, 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(det_uniform((Cout,), 44, -0.5, 0.5)).to(dev)
    res = torch.from_numpy(det_uniform((B, Ho, Ho, Cout), 45, -1, 1)).to(dev)
    x3, xps = _split(x)
    w3, wps = _wsplit(dev, w)
    y0 = torch.full((B, Ho, Ho, Cout), float('nan'), device=dev)
    y1 = torch.full((B, Ho, Ho, Cout), float('nan'), device=dev)
    hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, hipabi.ptr(y0), None,
                                      B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), 'fwd_x3')
    want, ps = _split(y0)
    for y in (y1, None):
        pl = torch.zeros(3, ps, device=dev, dtype=torch.int16)
        hipabi.check(L.straps_conv_fwd_x3p(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, hipabi.ptr(y),
                                           hipabi.ptr(pl), ps, B, H, H, Cin, Cout, k, k, stride, pad, cfg, None), 'fwd_x3p')
        assert torch.equal(pl, want)
        if y is not None:
            assert torch.equal(y, y0)


@pytest.mark.parametrize('B,H,W,C', [(3, 10, 14, 64), (2, 8, 8, 256), (1, 4, 12, 512), (1, 3, 5, 256), (2, 4, 4, 2048), (3, 10, 14, 128), (5, 6, 8, 1024)])
def test_fused_plane_outputs_equal_a_split_pass(dev, B, H, W, C):
    """straps_bn_apply_x3 / straps_bn_relu_maxpool_fwd_x3 / straps_bn_bwd_x3 write the same fp32 outputs as their plain forms and the
    planes a straps_split3_bf16 pass over that output would (bit for bit).  Shapes with rows % 4 == 0 and C % 256 == 0 run the tiled form
    of the streaming kernels (the plain straps_bn_apply never does: an independent index mapping), the others the linear forms."""
    L = hipabi.lib()
    rows = B * H * W
    raw = torch.from_numpy(det_uniform((B, H, W, C), 21, -2, 2)).to(dev)
    sc = torch.from_numpy(det_uniform((C,), 22, 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(det_uniform((C,), 23, -0.5, 0.5)).to(dev)
    res = torch.from_numpy(det_uniform((B, H, W, C), 24, -1, 1)).to(dev)
    y0, y1 = torch.empty_like(raw), torch.empty_like(raw)
    ps = (raw.numel() + 7) // 8 * 8
    pl = torch.zeros(3, ps, device=dev, dtype=torch.int16)
    hipabi.check(L.straps_bn_apply(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, hipabi.ptr(y0), rows, C, None), 'bn_apply')
    hipabi.check(L.straps_bn_apply_x3(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, hipabi.ptr(y1), hipabi.ptr(pl), ps, rows, C, None), 'bn_apply_x3')
    want, _ = _split(y0)
    assert torch.equal(y0, y1) and torch.equal(pl, want)
    pl.zero_()          # y = NULL: the planes alone (the fp32 copy is dead when only bf16x3 kernels consume it)
    hipabi.check(L.straps_bn_apply_x3(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), 1, None, hipabi.ptr(pl), ps, rows, C, None), 'bn_apply_x3')
    assert torch.equal(pl, want)
    # stem tail
    Hp, Wp = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    p0, p1 = torch.empty(B, Hp, Wp, C, device=dev), torch.empty(B, Hp, Wp, C, device=dev)
    i0, i1 = torch.empty(B, Hp, Wp, C, device=dev, dtype=torch.uint8), torch.empty(B, Hp, Wp, C, device=dev, dtype=torch.uint8)
    pps = (p0.numel() + 7) // 8 * 8
    ppl = torch.zeros(3, pps, device=dev, dtype=torch.int16)
    hipabi.check(L.straps_bn_relu_maxpool_fwd(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(p0), hipabi.ptr(i0), B, H, W, C, None), 'pool')
    hipabi.check(L.straps_bn_relu_maxpool_fwd_x3(hipabi.ptr(raw), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(p1), hipabi.ptr(i1), hipabi.ptr(ppl), pps,
                                                 B, H, W, C, None), 'pool_x3')
    want, _ = _split(p0)
    assert torch.equal(p0, p1) and torch.equal(i0, i1) and torch.equal(ppl, want)
    # BatchNorm backward
    dy = torch.from_numpy(det_uniform((B, H, W, C), 25, -1, 1)).to(dev) * 1e-3
    mean = raw.mean(dim=(0, 1, 2)).contiguous()
    invstd = (raw.var(dim=(0, 1, 2), unbiased=False) + 1e-5).rsqrt().contiguous()
    gamma = torch.from_numpy(det_uniform((C,), 26, 0.5, 1.5)).to(dev)
    ws = torch.empty(L.straps_bn_bwd_workspace_bytes(rows, C) // 4, device=dev)
    outs = []
    for planes in (None, torch.zeros(3, ps, device=dev, dtype=torch.int16)):
        dg, db, draw, dz = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty_like(raw), torch.empty_like(raw)
        hipabi.check(L.straps_bn_bwd_x3(hipabi.ptr(dy), hipabi.ptr(y0), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma), None, None,
                                        hipabi.ptr(dg), hipabi.ptr(db), hipabi.ptr(draw), hipabi.ptr(dz), hipabi.ptr(planes), ps if planes is not None else 0,
                                        hipabi.ptr(ws), rows, C, 0, None), 'bn_bwd_x3')
        outs.append((dg, db, draw, dz, planes))
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    want, _ = _split(outs[0][2])
    assert torch.equal(outs[1][4], want)
    planes = torch.zeros(3, ps, device=dev, dtype=torch.int16)      # draw = NULL: planes only
    dg, db, dz = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty_like(raw)
    hipabi.check(L.straps_bn_bwd_x3(hipabi.ptr(dy), hipabi.ptr(y0), hipabi.ptr(raw), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(gamma), None, None,
                                    hipabi.ptr(dg), hipabi.ptr(db), None, hipabi.ptr(dz), hipabi.ptr(planes), ps, hipabi.ptr(ws), rows, C, 0, None), 'bn_bwd_x3')
    assert torch.equal(planes, want) and torch.equal(dg, outs[0][0]) and torch.equal(dz, outs[0][3])


@pytest.mark.parametrize('layers', [18, 50])
def test_regressor_forward_on_the_x3_route_vs_fp32_route(dev, layers):
    """eval- and train-mode forward of the whole regressor with conv_precision = 'bf16x3' against the exact-fp32 route: the 85 outputs
    agree to 2e-5 (the bar of the fp32 route against the reference golden is 2e-4)."""
    torch.manual_seed(3)
    reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=straps_amd.synthetic_mean_params(0)).to(dev)
    x = torch.zeros(4, 18, 256, 256, device=dev)
    x[:, 0, 60:200, 80:180] = 1.0
    g = torch.Generator(device='cpu').manual_seed(5)
    for b in range(4):
        for j in range(17):
            cy, cx = (int(v) for v in torch.randint(20, 236, (2,), generator=g))
            x[b, 1 + j, cy - 3:cy + 4, cx - 3:cx + 4] = 0.8
    for mode in ('eval', 'train'):
        getattr(reg, mode)()
        outs = {}
        for prec in ('fp32', 'bf16x3'):
            reg.image_encoder.conv_precision = prec
            reg.image_encoder._cache.clear()
            with torch.no_grad():
                outs[prec] = torch.cat([t.reshape(4, -1) for t in reg(x)], 1).double().cpu()
        err = float((outs['fp32'] - outs['bf16x3']).abs().max())
        assert err < 2e-5, '%s-mode forward, resnet%d: routes differ by %.3e' % (mode, layers, err)
