"""GPU parity tests of the fp32-operand 1x1 convolution route (csrc/conv_x3f.hip, round 6): the A operand is read from the fp32 tensor, split into
its three bf16 parts in registers (hardware conversion), optionally behind the producer's BatchNorm + ReLU applied in the operand path
(models/resnet.py:34-36 conv1x1 inside the bottleneck units :80-121).

Bars (written where used):
  * against the float64 convolution: the bars of the plane route (2e-5 abs + 2e-5 rel forward; 2e-5 of the maximum for the data gradient);
  * against the PLANE route on the same values (straps_conv_fwd_x3 / straps_conv_dgrad_x3_bn_bits on straps_split3_bf16_cm planes, same tile
    shape => same reduction order): BIT FOR BIT -- the in-register split produces the planes the split kernels write;
  * operand-path BatchNorm: bit for bit against straps_bn_apply_x3 (+ ReLU) followed by the plane route.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import straps_amd  # noqa: F401
from detgen import det_uniform
from straps_amd import hipabi
from straps_amd.encoder_exec import split3, weight_planes

pytestmark = pytest.mark.gpu

# (x3f tile_cfg, plane-route tile_cfg of the same BM x BN shape and an unpipelined two- or three-stage loop: the same per-element reduction order)
TWINS = {1: 2, 2: 3}
# tile_cfg 5 = the persistent streaming kernel (other wave layout: its outputs equal the plane route's bit for bit -- a per-element reduction order does not
# depend on the tile -- its partial sums only to rounding)
STREAM = 5


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    hipabi.load()
    return torch.device('cuda:0')


def _nhwc(t, dev):
    return t.float().permute(0, 2, 3, 1).contiguous().to(dev)


def _fwd_x3f(dev, x, w, stride, cfg, a_scale=None, a_shift=None, a_relu=0, scale=None, shift=None, res=None, relu=0, stats=False):
    """x NHWC on the device; returns (y NHWC, stats partials)"""
    L = hipabi.lib()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
    w3, wps = weight_planes(L, w.to(dev))
    part = None
    if stats:
        nb = L.straps_conv_x3f_stat_blocks(B, H, W, Cin, Cout, 1, 1, stride, 0, cfg)
        assert nb > 0
        part = torch.full((nb, Cout, 2), float('nan'), device=dev)
    hipabi.check(L.straps_conv_fwd_x3f(hipabi.ptr(x), hipabi.ptr(a_scale), hipabi.ptr(a_shift), int(a_relu), hipabi.ptr(w3), wps, hipabi.ptr(scale),
                                       hipabi.ptr(shift), hipabi.ptr(res), int(relu), hipabi.ptr(y), hipabi.ptr(part), B, H, W, Cin, Cout, 1, 1, stride, 0, cfg,
                                       None), 'conv_fwd_x3f')
    torch.cuda.synchronize()
    return y, part


def _fwd_planes(dev, x, w, stride, cfg, scale=None, shift=None, res=None, relu=0, stats=False):
    L = hipabi.lib()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.full((B, Ho, Wo, Cout), float('nan'), device=dev)
    w3, wps = weight_planes(L, w.to(dev))
    x3, xps = split3(L, x)
    part = None
    if stats:
        part = torch.full((L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, 1, 1, stride, 0, cfg), Cout, 2), float('nan'), device=dev)
    hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, hipabi.ptr(scale), hipabi.ptr(shift), hipabi.ptr(res), int(relu),
                                      hipabi.ptr(y), hipabi.ptr(part), B, H, W, Cin, Cout, 1, 1, stride, 0, cfg, None), 'conv_fwd_x3')
    torch.cuda.synchronize()
    return y, part


FWD_CASES = [
    # B, Cin, Cout, H, W, stride, x3f tile cfg
    (2, 64, 256, 16, 16, 1, 0), (2, 64, 256, 16, 16, 1, 1), (2, 64, 256, 16, 16, 1, 2), (2, 256, 64, 16, 16, 1, 1), (2, 256, 64, 16, 16, 1, 2),
    (3, 128, 512, 8, 8, 1, 1), (2, 512, 128, 16, 16, 1, 2), (2, 256, 512, 16, 16, 2, 1), (1, 1024, 256, 8, 8, 1, 2), (2, 64, 64, 16, 16, 1, 0),
    (5, 64, 128, 7, 9, 1, 1), (3, 96, 128, 7, 13, 2, 2), (1, 32, 64, 3, 3, 1, 0), (2, 2048, 512, 4, 4, 1, 1), (2, 1024, 2048, 8, 8, 2, 2),
    # the streaming kernel: 256-wide resident weights (K = 64), the 128- and 64-wide ring forms, several tiles per workgroup (M = 36 992 rows > 256 tiles),
    # a ragged last tile, stride 2, K = 512
    (2, 64, 256, 16, 16, 1, 5), (17, 64, 256, 48, 48, 1, 5), (2, 256, 64, 16, 16, 1, 5), (9, 256, 64, 64, 64, 1, 5), (3, 128, 512, 24, 24, 1, 5), (2, 512, 128, 16, 16, 1, 5),
    (5, 64, 128, 7, 9, 1, 5), (2, 256, 512, 16, 16, 2, 5), (9, 64, 64, 64, 64, 1, 5),
]


@pytest.mark.parametrize('B,Cin,Cout,H,W,stride,cfg', FWD_CASES)
def test_conv_fwd_x3f_vs_float64_and_the_plane_route(dev, B, Cin, Cout, H, W, stride, cfg):
    """bars: 2e-5 abs + 2e-5 rel against the float64 convolution (the plane route's bar, tests/test_gpu_conv_x3.py); bit equality with the plane route
    at the twin tile shape, statistics partials included"""
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 1201, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, 1, 1), 1202, -1, 1)) * (2.0 / Cin) ** 0.5
    xd = _nhwc(x, dev)
    y, part = _fwd_x3f(dev, xd, w, stride, cfg, stats=True)
    ref = F.conv2d(x.double(), w.double(), stride=stride).permute(0, 2, 3, 1)
    got = y.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs()
    assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), 'max err %.3e' % err.max().item()
    # statistics partials: their sum over the blocks = (sum, sum of squares) of the output per channel (fp32 partials: 1e-4 relative)
    s = part.double().sum(0).cpu()
    r2 = ref.reshape(-1, Cout)
    assert torch.allclose(s[:, 0], r2.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(s[:, 1], (r2 * r2).sum(0), rtol=1e-4, atol=1e-2)
    if (cfg in TWINS or cfg == STREAM) and Cin % 32 == 0:
        yp, pp = _fwd_planes(dev, xd, w, stride, TWINS.get(cfg, 0), stats=True)
        assert torch.equal(y, yp), 'differs from the plane route: max %.3e' % (y - yp).abs().max().item()
        if cfg in TWINS:
            assert torch.equal(part, pp)


@pytest.mark.parametrize('B,Cin,Cout,H,W,cfg,relu', [(2, 64, 256, 16, 16, 1, 1), (2, 64, 256, 16, 16, 2, 1), (3, 128, 512, 8, 8, 2, 1), (2, 512, 128, 16, 16, 1, 1),
                                                      (2, 256, 64, 16, 16, 1, 0), (5, 64, 128, 7, 9, 0, 1), (17, 64, 256, 48, 48, 5, 1), (9, 256, 64, 64, 64, 5, 1),
                                                      (3, 128, 512, 24, 24, 5, 1), (2, 512, 128, 16, 16, 5, 0)])
def test_operand_path_batchnorm_equals_an_apply_pass_bit_for_bit(dev, B, Cin, Cout, H, W, cfg, relu):
    """conv1x1(relu(raw * scale + shift)) with the BatchNorm in the operand path == straps_bn_apply_x3 (planes of the activation) followed by the plane
    convolution: bit for bit (same fmaf, same split, same reduction order at the twin tile shape); and within the float64 bar"""
    L = hipabi.lib()
    raw = torch.from_numpy(det_uniform((B, Cin, H, W), 1301, -1.5, 1.5))
    w = torch.from_numpy(det_uniform((Cout, Cin, 1, 1), 1302, -1, 1)) * (2.0 / Cin) ** 0.5
    sc = torch.from_numpy(det_uniform((Cin,), 1303, 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(det_uniform((Cin,), 1304, -0.5, 0.5)).to(dev)
    rd = _nhwc(raw, dev)
    y, part = _fwd_x3f(dev, rd, w, 1, cfg, a_scale=sc, a_shift=sh, a_relu=relu, stats=True)
    act = raw.double() * sc.cpu().double().view(1, -1, 1, 1) + sh.cpu().double().view(1, -1, 1, 1)
    if relu:
        act = act.clamp_min(0)
    ref = F.conv2d(act, w.double()).permute(0, 2, 3, 1)
    err = (y.cpu().double() - ref).abs()
    assert (err <= 4e-5 + 2e-5 * ref.abs()).all(), 'max err %.3e' % err.max().item()      # (the fp32 fmaf of the BatchNorm adds its rounding: 4e-5 abs)
    if cfg in TWINS or cfg == STREAM:
        rows = B * H * W
        ps = (rows * Cin + 7) // 8 * 8
        planes = torch.empty(3, ps, device=dev, dtype=torch.int16)
        hipabi.check(L.straps_bn_apply_x3(hipabi.ptr(rd), hipabi.ptr(sc), hipabi.ptr(sh), None, int(relu), None, hipabi.ptr(planes), ps, rows, Cin, None), 'bn_apply_x3')
        w3, wps = weight_planes(L, w.to(dev))
        yp = torch.full_like(y, float('nan'))
        tw = TWINS.get(cfg, 0)
        pp = torch.full((L.straps_conv_x3_stat_blocks(B, H, W, Cin, Cout, 1, 1, 1, 0, tw), Cout, 2), float('nan'), device=dev)
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(planes), ps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(yp), hipabi.ptr(pp), B, H, W, Cin, Cout,
                                          1, 1, 1, 0, tw, None), 'conv_fwd_x3')
        torch.cuda.synchronize()
        assert torch.equal(y, yp)
        if cfg in TWINS:
            assert torch.equal(part, pp)
        else:
            assert torch.allclose(part.double().sum(0), pp.double().sum(0), rtol=1e-5, atol=1e-3)


def test_fwd_x3f_eval_epilogue_forms(dev):
    """folded scale / shift, residual and ReLU in the epilogue (the eval-mode forms of the plane kernel) behind the fp32 operand path"""
    B, Cin, Cout, H = 2, 128, 256, 8
    x = torch.from_numpy(det_uniform((B, Cin, H, H), 1401, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, 1, 1), 1402, -1, 1)) * (2.0 / Cin) ** 0.5
    sc = torch.from_numpy(det_uniform((Cout,), 1403, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((Cout,), 1404, -0.5, 0.5))
    res = torch.from_numpy(det_uniform((B, Cout, H, H), 1405, -1, 1))
    y, _ = _fwd_x3f(dev, _nhwc(x, dev), w, 1, 0, scale=sc.to(dev), shift=sh.to(dev), res=_nhwc(res, dev), relu=1)
    ref = (F.conv2d(x.double(), w.double()) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double()).clamp_min(0).permute(0, 2, 3, 1)
    err = (y.cpu().double() - ref).abs()
    assert (err <= 4e-5 + 2e-5 * ref.abs()).all(), 'max err %.3e' % err.max().item()


def _pack_relu_bits(y):
    """[rows][C] fp32 -> [rows][C / 32] int32 words, bit c & 31 = (y[row][c] > 0)  (what straps_bn_apply_bits_x3 writes)"""
    rows, C = y.shape
    b = (y > 0).view(rows, C // 32, 32).to(torch.int64)
    w = (b << torch.arange(32, device=y.device, dtype=torch.int64)).sum(-1)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
    return w.to(torch.int32).contiguous()


DGRAD_CASES = [
    # B, Cin, Cout, H, W, stride, cfg    (conv Cin -> Cout; the data gradient is [.., Cout] -> [.., Cin])
    (2, 256, 64, 16, 16, 1, 0), (2, 256, 64, 16, 16, 1, 1), (2, 256, 64, 16, 16, 1, 2), (2, 64, 256, 16, 16, 1, 1), (3, 512, 128, 8, 8, 1, 2),
    (2, 128, 512, 16, 16, 1, 1), (2, 256, 512, 16, 16, 2, 1), (2, 256, 512, 16, 16, 2, 2), (1, 1024, 2048, 8, 8, 2, 0), (5, 128, 64, 7, 9, 1, 1),
    # the streaming kernel (stride 1: one class): 64 -> 256 gradient (K = 256, 64 wide), 256 -> 64 gradient (K = 64, 256 wide, resident weights), ragged
    (9, 64, 256, 64, 64, 1, 5), (17, 256, 64, 48, 48, 1, 5), (3, 512, 128, 24, 24, 1, 5), (5, 128, 64, 7, 9, 1, 5), (2, 256, 512, 16, 16, 2, 5),
]


@pytest.mark.parametrize('B,Cin,Cout,H,W,stride,cfg', DGRAD_CASES)
def test_conv_dgrad_x3f_vs_float64_and_the_plane_route(dev, B, Cin, Cout, H, W, stride, cfg):
    """every optional operand at once: addend masked by ReLU bits, fused BatchNorm-backward sums masked by bits.  Bars: 2e-5 of the maximum against
    float64 autograd (the plane route's bar); sums 1e-5 relative of their scale; bit equality with straps_conv_dgrad_x3_bn_bits on split planes at the
    twin tile shape (dx AND partials)"""
    L = hipabi.lib()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = torch.from_numpy(det_uniform((B, Cout, Ho, Wo), 1501, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, 1, 1), 1502, -1, 1)) * (2.0 / Cout) ** 0.5
    add = torch.from_numpy(det_uniform((B, Cin, H, W), 1503, -1, 1))
    unit_out = torch.from_numpy(det_uniform((B, Cin, H, W), 1504, -1, 1))            # the later unit's output: its sign masks the addend
    raw = torch.from_numpy(det_uniform((B, Cin, H, W), 1505, -2, 2))                 # input of the BatchNorm whose output the convolution read
    bn_out = torch.from_numpy(det_uniform((B, Cin, H, W), 1506, -1, 1))              # that BatchNorm unit's output (residual unit: sign from bits)
    mean = torch.from_numpy(det_uniform((Cin,), 1507, -0.5, 0.5)).to(dev)
    invstd = torch.from_numpy(det_uniform((Cin,), 1508, 0.5, 2.0)).to(dev)
    dyd, addd = _nhwc(dy, dev), _nhwc(add, dev)
    rows = B * H * W
    abits = _pack_relu_bits(_nhwc(unit_out, dev).view(rows, Cin))
    obits = _pack_relu_bits(_nhwc(bn_out, dev).view(rows, Cin))
    rawd = _nhwc(raw, dev)
    w3, wps = weight_planes(L, w.to(dev), dgrad=True)

    def run_f(cfg_):
        dx = torch.full((B, H, W, Cin), float('nan'), device=dev)
        nb = L.straps_conv_dgrad_x3f_bn_blocks(B, H, W, Cin, Cout, 1, 1, stride, 0, cfg_)
        part = torch.full((nb, Cin, 2), float('nan'), device=dev, dtype=torch.float64)
        hipabi.check(L.straps_conv_dgrad_x3f(hipabi.ptr(dyd), hipabi.ptr(w3), wps, hipabi.ptr(addd), hipabi.ptr(abits), hipabi.ptr(dx), B, H, W, Cin, Cout, 1, 1,
                                             stride, 0, cfg_, hipabi.ptr(rawd), hipabi.ptr(obits), None, None, hipabi.ptr(mean), hipabi.ptr(invstd),
                                             hipabi.ptr(part), None), 'conv_dgrad_x3f')
        torch.cuda.synchronize()
        return dx, part

    dx, part = run_f(cfg)
    # float64 reference
    xx = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xx, w.double(), stride=stride).backward(dy.double())
    ref = (xx.grad + add.double() * (unit_out > 0).double()).permute(0, 2, 3, 1)
    got = dx.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)
    g = ref * (bn_out > 0).double().permute(0, 2, 3, 1)
    s1 = g.reshape(-1, Cin).sum(0)
    s2 = (g * (raw.double().permute(0, 2, 3, 1) - mean.cpu().double())).reshape(-1, Cin).sum(0) * invstd.cpu().double()
    ps = part.sum(0).cpu()
    scale = g.abs().reshape(-1, Cin).sum(0).clamp_min(1.0)
    assert ((ps[:, 0] - s1).abs() <= 1e-5 * scale).all() and ((ps[:, 1] - s2).abs() <= 1e-4 * scale).all()
    if cfg in TWINS or cfg == STREAM:
        g3, gps = split3(L, dyd)
        dxp = torch.full_like(dx, float('nan'))
        nbp = L.straps_conv_dgrad_x3_bn_blocks(B, H, W, Cin, Cout, 1, 1, stride, 0, TWINS.get(cfg, 0))
        pp = torch.full((nbp, Cin, 2), float('nan'), device=dev, dtype=torch.float64)
        hipabi.check(L.straps_conv_dgrad_x3_bn_bits(hipabi.ptr(g3), gps, hipabi.ptr(w3), wps, hipabi.ptr(addd), hipabi.ptr(dxp), B, H, W, Cin, Cout, 1, 1, stride,
                                                    0, TWINS.get(cfg, 0), hipabi.ptr(rawd), None, None, None, hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(pp),
                                                    hipabi.ptr(abits), hipabi.ptr(obits), None), 'conv_dgrad_x3_bn_bits')
        torch.cuda.synchronize()
        assert torch.equal(dx, dxp), 'dx differs from the plane route: max %.3e' % (dx - dxp).abs().max().item()
        if cfg in TWINS:
            # (the lean data-gradient epilogue pre-sums a unit's 16 values in fp32 before the double accumulation: partials equal to rounding, stride 2 --
            #  the shared epilogue -- bit for bit)
            assert nbp == part.shape[0]
            if stride == 2:
                assert torch.equal(part, pp)
            else:
                sc_ = pp.abs().sum(0).clamp_min(1.0)
                assert ((part.sum(0) - pp.sum(0)).abs() <= 1e-6 * sc_).all()


def test_conv_dgrad_x3f_plain_and_mask_from_raw(dev):
    """no addend / no sums; and the sums with the ReLU mask re-derived from raw (bn_mask_scale / bn_mask_shift) -- against float64"""
    L = hipabi.lib()
    B, Cin, Cout, H = 2, 128, 256, 8
    dy = torch.from_numpy(det_uniform((B, Cout, H, H), 1601, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, 1, 1), 1602, -1, 1)) * (2.0 / Cout) ** 0.5
    raw = torch.from_numpy(det_uniform((B, Cin, H, H), 1603, -2, 2))
    msc = torch.from_numpy(det_uniform((Cin,), 1604, 0.5, 1.5)).to(dev)
    msh = torch.from_numpy(det_uniform((Cin,), 1605, -0.5, 0.5)).to(dev)
    mean = torch.from_numpy(det_uniform((Cin,), 1606, -0.5, 0.5)).to(dev)
    invstd = torch.from_numpy(det_uniform((Cin,), 1607, 0.5, 2.0)).to(dev)
    dyd, rawd = _nhwc(dy, dev), _nhwc(raw, dev)
    w3, wps = weight_planes(L, w.to(dev), dgrad=True)
    xx = torch.zeros(B, Cin, H, H, dtype=torch.float64, requires_grad=True)
    F.conv2d(xx, w.double()).backward(dy.double())
    ref = xx.grad.permute(0, 2, 3, 1)
    dx = torch.full((B, H, H, Cin), float('nan'), device=dev)
    hipabi.check(L.straps_conv_dgrad_x3f(hipabi.ptr(dyd), hipabi.ptr(w3), wps, None, None, hipabi.ptr(dx), B, H, H, Cin, Cout, 1, 1, 1, 0, 0, None, None, None, None,
                                         None, None, None, None), 'conv_dgrad_x3f')
    torch.cuda.synchronize()
    assert (dx.cpu().double() - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)
    nb = L.straps_conv_dgrad_x3f_bn_blocks(B, H, H, Cin, Cout, 1, 1, 1, 0, 0)
    part = torch.full((nb, Cin, 2), float('nan'), device=dev, dtype=torch.float64)
    dx2 = torch.full_like(dx, float('nan'))
    hipabi.check(L.straps_conv_dgrad_x3f(hipabi.ptr(dyd), hipabi.ptr(w3), wps, None, None, hipabi.ptr(dx2), B, H, H, Cin, Cout, 1, 1, 1, 0, 0, hipabi.ptr(rawd), None,
                                         hipabi.ptr(msc), hipabi.ptr(msh), hipabi.ptr(mean), hipabi.ptr(invstd), hipabi.ptr(part), None), 'conv_dgrad_x3f + sums')
    torch.cuda.synchronize()
    assert torch.equal(dx, dx2)
    on = (rawd.double() * msc.double() + msh.double()) > 0            # (fp64 restatement of the mask; ties are measure-zero on this data)
    g = ref.to(dev) * on
    s1 = g.reshape(-1, Cin).sum(0)
    s2 = (g * (rawd.double() - mean.double())).reshape(-1, Cin).sum(0) * invstd.double()
    ps = part.sum(0)
    scale = g.abs().reshape(-1, Cin).sum(0).clamp_min(1.0)
    assert ((ps[:, 0] - s1).abs() <= 1e-5 * scale).all() and ((ps[:, 1] - s2).abs() <= 1e-4 * scale).all()


def test_x3f_refuses_what_it_does_not_cover(dev):
    L = hipabi.lib()
    assert L.straps_conv_x3f_supported(64, 256, 1, 1, 1, 0) == 1 and L.straps_conv_x3f_supported(64, 64, 3, 3, 1, 1) == 0
    x = torch.zeros(1, 8, 8, 64, device=dev)
    y = torch.zeros(1, 8, 8, 64, device=dev)
    w3 = torch.zeros(3, 64 * 64 * 9 + 8, device=dev, dtype=torch.int16)
    assert L.straps_conv_fwd_x3f(hipabi.ptr(x), None, None, 0, hipabi.ptr(w3), 64 * 64 * 9 + 8, None, None, None, 0, hipabi.ptr(y), None, 1, 8, 8, 64, 64, 3, 3, 1, 1,
                                 0, None) != 0
    assert b'1x1' in L.straps_last_error()
    assert L.straps_conv_x3f_stat_blocks(1, 8, 8, 64, 64, 3, 3, 1, 1, 0) == -1


WGRAD_CASES = [
    # B, Cin, Cout, H, W, stride, operand-path BatchNorm     (every channel block of csrc/conv_wgrad_x3f.hip; ragged pixel counts; stride 2)
    (2, 64, 256, 16, 16, 1, 1), (2, 256, 64, 16, 16, 1, 0), (3, 128, 512, 8, 8, 1, 1), (2, 512, 128, 16, 16, 1, 0), (2, 64, 64, 16, 16, 1, 1),
    (2, 128, 128, 8, 8, 1, 0), (5, 64, 256, 7, 9, 1, 1), (3, 256, 512, 9, 7, 2, 0), (1, 1024, 256, 8, 8, 1, 0), (2, 256, 1024, 8, 8, 1, 1),
    (17, 64, 256, 48, 48, 1, 1), (2, 1024, 2048, 8, 8, 2, 0), (2, 512, 2048, 4, 4, 1, 1),
]


@pytest.mark.parametrize('B,Cin,Cout,H,W,stride,bn', WGRAD_CASES)
def test_conv_wgrad_x3f_vs_float64(dev, B, Cin, Cout, H, W, stride, bn):
    """dW of a 1x1 convolution from the fp32 tensors, optionally with the producer's BatchNorm + ReLU in the operand path.  Bar: 2e-5 of the maximum
    against float64 autograd -- the plane kernel's bar (tests/test_gpu_conv_x3.py::test_conv_wgrad_x3_vs_float64); accumulate adds"""
    L = hipabi.lib()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 1701, -1, 1))
    dy = torch.from_numpy(det_uniform((B, Cout, Ho, Wo), 1702, -1, 1))
    sc = torch.from_numpy(det_uniform((Cin,), 1703, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((Cin,), 1704, -0.5, 0.5))
    xd, dyd = _nhwc(x, dev), _nhwc(dy, dev)
    scd, shd = sc.to(dev), sh.to(dev)
    act = x.double()
    if bn:
        act = (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).double()      # (the fp32 fmaf's rounding is inside the bar)
        act = act.clamp_min(0)
    w = torch.zeros(Cout, Cin, 1, 1, dtype=torch.float64, requires_grad=True)
    F.conv2d(act, w, stride=stride).backward(dy.double())
    ref = w.grad
    ws = torch.empty(max(L.straps_conv_wgrad_x3f_workspace_bytes(B, H, W, Cin, Cout, 1, 1, stride, 0) // 4, 1), device=dev)
    dw = torch.full((Cout, Cin, 1, 1), float('nan'), device=dev)
    hipabi.check(L.straps_conv_wgrad_x3f(hipabi.ptr(xd), hipabi.ptr(scd if bn else None), hipabi.ptr(shd if bn else None), int(bn), hipabi.ptr(dyd), hipabi.ptr(dw), hipabi.ptr(ws),
                                         B, H, W, Cin, Cout, 1, 1, stride, 0, 0, None), 'conv_wgrad_x3f')
    torch.cuda.synchronize()
    got = dw.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0), 'max err %.3e of %.3e' % ((got - ref).abs().max().item(), ref.abs().max().item())
    hipabi.check(L.straps_conv_wgrad_x3f(hipabi.ptr(xd), hipabi.ptr(scd if bn else None), hipabi.ptr(shd if bn else None), int(bn), hipabi.ptr(dyd), hipabi.ptr(dw), hipabi.ptr(ws),
                                         B, H, W, Cin, Cout, 1, 1, stride, 0, 1, None), 'conv_wgrad_x3f accumulate')
    torch.cuda.synchronize()
    assert (dw.cpu().double() - 2 * ref).abs().max().item() <= 4e-5 * max(ref.abs().max().item(), 1.0)
