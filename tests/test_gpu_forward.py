"""GPU parity tests of the forward hot path: every HIP kernel against the CPU oracle (and the
committed golden vectors), called through the C ABI exactly as the product does.

Tolerances (written where used):
  * SMPL vertices / joints: <= 1e-4 abs (BASELINE.json north_star) -- measured ~1e-6;
  * encoder / IEF outputs: fp32 with a different summation order than the oracle's CPU kernels
    -> <= 2e-4 abs + 2e-4 rel on O(1) activations (measured ~1e-5).
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import straps_amd
import straps_oracle as O
from detgen import det_uniform, det_state_dict
from straps_amd import hipabi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MP = straps_amd.synthetic_mean_params(0)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    hipabi.load()                       # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def smpl_model():
    return straps_amd.synthetic_smpl_model(0)


def _close(a, b, atol, rtol, what):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    worst = float((err - bound).max())
    assert worst <= 0, '%s: max abs err %.3e (bound violated by %.3e)' % (what, float(err.max()), worst)
    return float(err.max())


# ------------------------------------------------------------------------------------------ SMPL
@pytest.mark.parametrize('B,chunks', [(1, 0), (5, 0), (37, 8), (70, 54), (64, 3)])
def test_smpl_forward_vs_oracle(dev, smpl_model, B, chunks):
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 100 + B, -2.5, 2.5))
    aa = torch.from_numpy(det_uniform((B, 72), 200 + B, -0.9, 0.9))
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), chunks=chunks)
    v64, j64 = O.smpl_forward(smpl_model, betas.double(), rotmats=R.double(), dtype=torch.float64)
    ev = _close(v, v64, 1e-4, 0, 'vertices')       # north_star: <= 1e-4 abs
    ej = _close(j, j64, 1e-4, 0, 'joints')
    assert ev < 2e-5 and ej < 2e-5                  # in practice fp32 round-off only
    # vertices only (config 5 microbench form)
    v2, none = smpl.forward_arrays(betas.to(dev), R.to(dev), want_joints=False, chunks=chunks)
    assert none is None and torch.equal(v2, v)


@pytest.mark.parametrize('mode', ['fp16x3', 'fp16x3_lbs'])
@pytest.mark.parametrize('B', [1, 33, 70, 4096])
def test_smpl_split_precision_vs_oracle(dev, smpl_model, B, mode):
    """STRAPS_SMPL_SPLIT_F16 (blend contraction as three fp16-MFMA products of two-term splits, fp32 accumulate) and
    STRAPS_SMPL_SPLIT_F16_LBS (the skinning transforms as split products too) against the float64 oracle on identical
    (theta, beta): north_star's bar is 1e-4 m; the test asserts 2e-5 and that the split kernels are as close to float64 as
    the exact-fp32 kernel (both errors printed)."""
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 100 + B, -2.5, 2.5))
    betas[0] = torch.tensor([10.0, -8.0, 6.0, 4.0, -4.0, 3.0, 3.0, -3.0, 2.0, 2.0])          # an extreme body
    aa = torch.from_numpy(det_uniform((B, 72), 200 + B, -0.9, 0.9))
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), precision=mode)
    v32, j32 = smpl.forward_arrays(betas.to(dev), R.to(dev), precision='fp32')
    n = min(B, 96)                                                                         # the float64 oracle on a slice
    v64, j64 = O.smpl_forward(smpl_model, betas[:n].double(), rotmats=R[:n].double(), dtype=torch.float64)
    ev, ej = float((v[:n].cpu().double() - v64).abs().max()), float((j[:n].cpu().double() - j64).abs().max())
    ev32, ej32 = float((v32[:n].cpu().double() - v64).abs().max()), float((j32[:n].cpu().double() - j64).abs().max())
    print('SMPL B=%d max |err| vs float64: %s verts %.2e joints %.2e | exact-fp32 verts %.2e joints %.2e' % (B, mode, ev, ej, ev32, ej32))
    assert ev < 2e-5 and ej < 2e-5
    assert ev <= 3 * ev32 + 1e-6 and ej <= 3 * ej32 + 1e-6
    assert float((v - v32).abs().max()) < 1e-5 and float((j - j32).abs().max()) < 1e-5
    assert torch.isfinite(v).all() and torch.isfinite(j).all()
    # deterministic, vertices-only form identical, batch rows independent of the batch they ride in
    v2, none = smpl.forward_arrays(betas.to(dev), R.to(dev), want_joints=False, precision=mode)
    assert none is None and torch.equal(v2, v)
    if B > 40:
        vs, js = smpl.forward_arrays(betas[30:37].contiguous().to(dev), R[30:37].contiguous().to(dev), precision=mode)
        assert torch.equal(vs, v[30:37]) and torch.equal(js, j[30:37])
    # module-level switch
    fast = straps_amd.SMPL(smpl_model, batch_size=B, precision=mode).to(dev)
    with torch.no_grad():
        o = fast(body_pose=R[:, 1:].to(dev), global_orient=R[:, 0:1].to(dev), betas=betas.to(dev), pose2rot=False)
    assert torch.equal(o.vertices, v)
    with pytest.raises(ValueError):
        straps_amd.SMPL(smpl_model, batch_size=1, precision='bf16')


@pytest.mark.parametrize('mode', ['fp16x3_lbs_pd16', 'fp16x3_lbs_p16'])
@pytest.mark.parametrize('B', [1, 70, 4096])
def test_smpl_pd16_mode_vs_oracle(dev, smpl_model, B, mode):
    """STRAPS_SMPL_SPLIT_F16_LBS_PD16: pose-corrective directions as plain fp16 (two products), template / shape / skinning as the
    three-product splits.  Not as close to float64 as fp32 arithmetic -- the point of the mode is throughput inside north_star's
    1e-4 m: asserted < 2e-5 m (the bar of every SMPL test here), measured error printed; everything else as the other modes."""
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 100 + B, -2.5, 2.5))
    betas[0] = torch.tensor([10.0, -8.0, 6.0, 4.0, -4.0, 3.0, 3.0, -3.0, 2.0, 2.0])
    aa = torch.from_numpy(det_uniform((B, 72), 200 + B, -0.9, 0.9))
    aa[-1] = torch.from_numpy(det_uniform((72,), 7, -3.0, 3.0))                             # rotations up to pi per axis component
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), precision=mode)
    n = min(B, 96)
    idx = list(range(n - 1)) + [B - 1]
    v64, j64 = O.smpl_forward(smpl_model, betas[idx].double(), rotmats=R[idx].double(), dtype=torch.float64)
    ev, ej = float((v[idx].cpu().double() - v64).abs().max()), float((j[idx].cpu().double() - j64).abs().max())
    print('SMPL B=%d %s max |err| vs float64: verts %.2e joints %.2e' % (B, mode, ev, ej))
    assert ev < 2e-5 and ej < 2e-5
    assert torch.isfinite(v).all() and torch.isfinite(j).all()
    v2, none = smpl.forward_arrays(betas.to(dev), R.to(dev), want_joints=False, precision=mode)
    assert none is None and torch.equal(v2, v)
    if B > 40:
        vs, js = smpl.forward_arrays(betas[30:37].contiguous().to(dev), R[30:37].contiguous().to(dev), precision=mode)
        assert torch.equal(vs, v[30:37]) and torch.equal(js, j[30:37])


def _real_magnitude_model(model):
    """the synthetic SMPL model with its blend directions scaled to the real model's magnitudes (missing #3): the synthetic
    posedirs are uniform +-1e-3 where SMPL_NEUTRAL's largest entries are one to two orders larger, and the shape directions about three times
    smaller than the real ones -- every 'm from float64' figure of the split-precision modes above is a statement about the SMALL directions."""
    big = dict(model)
    big['posedirs'] = (np.asarray(model['posedirs'], dtype=np.float64) * 50.0).astype(np.float32)
    big['shapedirs'] = (np.asarray(model['shapedirs'], dtype=np.float64) * 3.0).astype(np.float32)
    return big


# mode -> bar in metres at the real model's magnitudes.  The three parity-grade modes keep the 2e-5 of every SMPL test here (north_star: 1e-4).
# The two opt-in throughput modes round the pose-corrective directions to plain fp16: their error scales with those directions, and at the
# real model's magnitudes it EXCEEDS north_star's 1e-4 (round 5, first measurement: 2-3e-4 m where the synthetic model showed 5-6e-6) --
# they are NOT parity modes, nothing in the product selects them, and the test pins that finding from both sides.
_REAL_MAGNITUDE_BARS = {'fp32': 2e-5, 'fp16x3': 2e-5, 'fp16x3_lbs': 2e-5, 'fp16x3_lbs_pd16': 1e-3, 'fp16x3_lbs_p16': 1e-3}
_OUTSIDE_NORTH_STAR = ('fp16x3_lbs_pd16', 'fp16x3_lbs_p16')


@pytest.mark.parametrize('kernel', ['narrow', 'wide'])
@pytest.mark.parametrize('mode', ['fp32', 'fp16x3', 'fp16x3_lbs', 'fp16x3_lbs_pd16', 'fp16x3_lbs_p16'])
def test_smpl_precision_modes_at_real_model_magnitudes(dev, smpl_model, mode, kernel):
    """every SMPL precision mode against the float64 oracle on a model whose pose-corrective directions are 50x and whose shape directions are
    3x the synthetic model's (the real model's scale), with an extreme body (|beta| up to 10) and a body whose joint rotations reach pi --
    the only check that stands in for models/smpl_official.py:27-41 on real data while the licensed model file is absent."""
    if mode in ('fp32', 'fp16x3') and kernel == 'wide':
        pytest.skip('the 64-body kernel exists for the fp16x3_lbs* modes only')
    big = _real_magnitude_model(smpl_model)
    B = 70
    smpl = straps_amd.SMPL(big, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 100 + B, -2.5, 2.5))
    betas[0] = torch.tensor([10.0, -8.0, 6.0, 4.0, -4.0, 3.0, 3.0, -3.0, 2.0, 2.0])
    aa = torch.from_numpy(det_uniform((B, 72), 200 + B, -0.9, 0.9))
    aa[-1] = torch.from_numpy(det_uniform((72,), 7, -3.0, 3.0))                             # rotations up to pi per axis component
    aa[-2] = torch.from_numpy(det_uniform((72,), 8, -1.8, 1.8))
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    kw = {} if mode in ('fp32', 'fp16x3') else {'kernel': kernel}
    v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), precision=mode, **kw)
    v64, j64 = O.smpl_forward(big, betas.double(), rotmats=R.double(), dtype=torch.float64)
    ev, ej = float((v.cpu().double() - v64).abs().max()), float((j.cpu().double() - j64).abs().max())
    span = float((v64 - O.smpl_forward(smpl_model, betas.double(), rotmats=R.double(), dtype=torch.float64)[0]).abs().max())
    print('SMPL at real magnitudes, %s / %s kernel: max |err| vs float64 verts %.2e joints %.2e (the scaled directions move vertices by up to %.2f m)'
          % (mode, kernel, ev, ej, span))
    assert span > 0.05                                   # the scaled directions do matter
    bar = _REAL_MAGNITUDE_BARS[mode]
    assert ev < bar and ej < bar, 'mode %s: %.2e / %.2e m from float64 at the real model\'s magnitudes (bar %.0e)' % (mode, ev, ej, bar)
    if mode in _OUTSIDE_NORTH_STAR:
        # (should a future change bring these modes inside 1e-4 at these magnitudes, this line fails and the documentation is corrected)
        assert ev > 1e-4, 'mode %s is now %.2e m from float64 at the real magnitudes: inside north_star -- update DESIGN / README' % (mode, ev)
    assert torch.isfinite(v).all() and torch.isfinite(j).all()


@pytest.mark.parametrize('mode', ['fp16x3', 'fp16x3_lbs', 'fp16x3_lbs_p16'])
def test_smpl_split_modes_saturate_out_of_range_operands(dev, smpl_model, mode):
    """a diverging regressor can predict betas in the thousands: the split kernels' fp16 operands then SATURATE (finite output for that
    body, clipped) instead of overflowing to inf / NaN -- and the other bodies of the batch are bit-identical to a run without the outlier
    (bodies are independent).  The exact-fp32 kernel has no such range."""
    B = 40
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 300, -2.5, 2.5))
    aa = torch.from_numpy(det_uniform((B, 72), 301, -0.9, 0.9))
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3).to(dev)
    v0, j0 = smpl.forward_arrays(betas.to(dev), R, precision=mode)
    wild = betas.clone()
    wild[5] = 5000.0
    wild[17, 3] = -2.0e6
    v1, j1 = smpl.forward_arrays(wild.to(dev), R, precision=mode)
    assert torch.isfinite(v1).all() and torch.isfinite(j1).all()
    keep = [b for b in range(B) if b not in (5, 17)]
    assert torch.equal(v1[keep], v0[keep]) and torch.equal(j1[keep], j0[keep])
    vf, _ = smpl.forward_arrays(wild.to(dev), R, precision='fp32')
    assert torch.isfinite(vf).all() and float(vf[5].abs().max()) > float(v1[5].abs().max())      # fp32 follows the huge betas, the split modes clip


def test_smpl_module_call_forms(dev, smpl_model):
    """the three call forms of the reference (train loop :132, :144, :258)."""
    B = 4
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 7, -2, 2))
    aa = torch.from_numpy(det_uniform((B, 72), 8, -0.7, 0.7))
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    with torch.no_grad():
        o1 = smpl(body_pose=R[:, 1:].to(dev), global_orient=R[:, 0:1].to(dev), betas=betas.to(dev), pose2rot=False)
        o2 = smpl(body_pose=aa[:, 3:].to(dev), global_orient=aa[:, :3].to(dev), betas=betas.to(dev))
        o3 = smpl(betas=betas.to(dev))
    v1, j1 = O.smpl_forward(smpl_model, betas, rotmats=R)
    v3, j3 = O.smpl_forward(smpl_model, betas, rotmats=torch.eye(3).expand(B, 24, 3, 3))
    _close(o1.vertices, v1, 1e-4, 0, 'rotmat form')
    _close(o2.vertices, v1, 1e-4, 0, 'axis-angle form')
    _close(o2.joints, j1, 1e-4, 0, 'axis-angle form joints')
    _close(o3.vertices, v3, 1e-4, 0, 'betas-only form')
    assert o1.vertices.shape == (B, 6890, 3) and o1.joints.shape == (B, 90, 3)
    assert o2.full_pose.shape == (B, 72) and o1.betas.shape == (B, 10)


def test_smpl_large_batch_properties(dev, smpl_model):
    """size-independent properties at a batch the oracle cannot do in seconds (4096 bodies, 8 chunks):
    zero pose == shaped template; a global rotation moves vertices rigidly about the root joint."""
    B = 4096
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    g = torch.Generator().manual_seed(3)
    betas = (torch.rand(B, 10, generator=g) * 4 - 2).to(dev)
    eye = torch.eye(3, device=dev).expand(B, 24, 3, 3).contiguous()
    v0, j0 = smpl.forward_arrays(betas, eye)
    vt = torch.tensor(smpl_model['v_template'], device=dev)
    sdirs = torch.tensor(smpl_model['shapedirs'], device=dev)
    v_shaped = vt[None] + torch.einsum('bl,vcl->bvc', betas, sdirs)
    assert float((v0 - v_shaped).abs().max()) < 5e-6
    aa = (torch.rand(B, 72, generator=g) - 0.5)
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3).to(dev)
    v1, j1 = smpl.forward_arrays(betas, R)
    Rg = O.batch_rodrigues(torch.tensor([[0.4, -0.9, 0.2]]))[0].to(dev)
    R2 = R.clone()
    R2[:, 0] = Rg @ R[:, 0]
    v2, j2 = smpl.forward_arrays(betas, R2)
    root = j1[:, 0:1]
    assert float((j2[:, 0:1] - root).abs().max()) < 1e-6
    assert float((v2 - ((v1 - root) @ Rg.T + root)).abs().max()) < 5e-6
    assert float((j2 - ((j1 - root) @ Rg.T + root)).abs().max()) < 5e-6
    # picked-vertex joints are exact copies
    assert torch.equal(j1[:, 24:45], v1[:, smpl_model['extra_vertex_ids'].tolist()])
    # batch slices are independent of the batch they ride in
    v_s, j_s = smpl.forward_arrays(betas[1000:1037].contiguous(), R[1000:1037].contiguous())
    assert torch.equal(v_s, v1[1000:1037]) and float((j_s - j1[1000:1037]).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------------ pose
def test_rot6d_and_rodrigues(dev):
    gold = np.load(os.path.join(GOLD, 'small_golden.npz'))
    x6 = torch.from_numpy(det_uniform((4, 144), 32, -1.5, 1.5))
    R = straps_amd.rot6d_to_rotmat(x6.to(dev))
    assert R.shape == (96, 3, 3)
    _close(R, torch.from_numpy(gold['rot6d_out']), 2e-6, 0, 'rot6d vs reference golden')
    # strided view (the IEF estimate's pose slice) gives the same result
    buf = torch.zeros(4, 160)
    buf[:, 3:147] = x6
    R2 = straps_amd.rot6d_to_rotmat(buf.to(dev)[:, 3:147])
    assert torch.equal(R2, R)
    z = torch.zeros(2, 6, device=dev)               # degenerate input: F.normalize eps path, no NaN
    assert torch.isfinite(straps_amd.rot6d_to_rotmat(z)).all()
    aa = torch.from_numpy(det_uniform((300, 3), 9, -3, 3))
    aa[0] = 0
    _close(straps_amd.batch_rodrigues(aa.to(dev)), O.batch_rodrigues(aa.double()), 2e-6, 0, 'rodrigues')


# ------------------------------------------------------------------------------------------ conv pieces
def _run_conv(dev, x_nchw, w, stride, pad, scale=None, shift=None, res_nchw=None, relu=False, cfg=0, stats=False):
    L = hipabi.lib()
    B, Cin, H, W = x_nchw.shape
    Cout, _, k, _ = w.shape
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.contiguous().to(dev)
    wp = torch.empty_like(wd)
    hipabi.check(L.straps_pack_conv_weight(hipabi.ptr(wd), hipabi.ptr(wp), Cout, Cin, k, k, None), 'pack')
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, device=dev)
    res = res_nchw.permute(0, 2, 3, 1).contiguous().to(dev) if res_nchw is not None else None
    sc = scale.to(dev) if scale is not None else None
    sh = shift.to(dev) if shift is not None else None
    part = None
    if stats:
        part = torch.empty(L.straps_conv_stat_blocks(B, Ho, Wo, Cout, k * k * Cin, cfg), Cout, 2, device=dev)
    hipabi.check(L.straps_conv_fwd(hipabi.ptr(x), hipabi.ptr(wp), hipabi.ptr(sc), hipabi.ptr(sh), hipabi.ptr(res), int(relu),
                                   hipabi.ptr(y), hipabi.ptr(part), B, H, W, Cin, Cout, k, k, stride, pad, cfg, None), 'conv')
    torch.cuda.synchronize()
    return y.permute(0, 3, 1, 2).cpu(), part


@pytest.mark.parametrize('B,Cin,Cout,H,k,stride,cfg', [
    (2, 64, 64, 16, 3, 1, 0), (2, 64, 64, 16, 3, 1, 1), (2, 64, 128, 16, 3, 2, 2), (3, 128, 128, 8, 3, 1, 3),
    (2, 64, 128, 16, 1, 2, 0), (1, 256, 512, 8, 3, 2, 0), (1, 512, 512, 8, 3, 1, 0), (5, 64, 64, 7, 3, 1, 1),
    (2, 256, 64, 16, 1, 1, 0)])
def test_conv_igemm_vs_cpu(dev, B, Cin, Cout, H, k, stride, cfg):
    """implicit-GEMM conv (all tile configs, ragged M, stride 1/2, 3x3 and 1x1) vs F.conv2d fp64."""
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, H), 1, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 2, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    y, part = _run_conv(dev, x, w, stride, pad, cfg=cfg, stats=True)
    _close(y, ref, 2e-5, 2e-5, 'raw conv')
    # training-mode statistics partials: per-channel sum / sum of squares of the raw output
    s = part.double().sum(0).cpu()
    np.testing.assert_allclose(s[:, 0].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # fused epilogue: BN scale/shift + residual + ReLU
    sc = torch.from_numpy(det_uniform((Cout,), 3, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((Cout,), 4, -0.5, 0.5))
    res = torch.from_numpy(det_uniform(tuple(ref.shape), 5, -1, 1))
    want = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res.double())
    y2, _ = _run_conv(dev, x, w, stride, pad, sc, sh, res, True, cfg)
    _close(y2, want, 3e-5, 3e-5, 'fused epilogue')


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride', [(2, 64, 64, 10, 24, 3, 1), (3, 96, 128, 7, 13, 3, 2), (2, 64, 64, 5, 40, 1, 1), (1, 32, 64, 3, 3, 3, 1)])
def test_conv_igemm_non_square(dev, B, Cin, Cout, H, W, k, stride):
    """implicit GEMM forward + data gradient on non-square maps, channel counts of 32 / 96, a 3x3 map smaller than one tile."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, W), 51, -1, 1)).double().requires_grad_()
    w = (torch.from_numpy(det_uniform((Cout, Cin, k, k), 52, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5).double().requires_grad_()
    ref = F.conv2d(x, w, None, stride, pad)
    y, _ = _run_conv(dev, x.detach().float(), w.detach().float(), stride, pad, stats=True)
    _close(y, ref.detach(), 2e-5, 2e-5, 'raw conv')
    if Cin % 64 == 0:
        dy = torch.from_numpy(det_uniform(tuple(ref.shape), 53, -1, 1)).double()
        ref.backward(dy)
        wd = w.detach().float().to(dev)
        wpk = torch.empty_like(wd)
        hipabi.check(L.straps_pack_conv_weight_dgrad(hipabi.ptr(wd), hipabi.ptr(wpk), Cout, Cin, k, k, None), 'pack dgrad')
        dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev)
        dx = torch.full((B, H, W, Cin), float('nan'), device=dev)
        hipabi.check(L.straps_conv_dgrad(hipabi.ptr(dyd), hipabi.ptr(wpk), None, hipabi.ptr(dx), B, H, W, Cin, Cout, k, k, stride, pad, 0, None), 'dgrad')
        _close(dx.permute(0, 3, 1, 2).cpu(), x.grad, 2e-5, 2e-5, 'dgrad')


@pytest.mark.parametrize('B,Cin,Cout,H,k,stride,tile', [
    (3, 64, 64, 13, 3, 1, 3), (2, 128, 128, 16, 3, 1, 1), (2, 64, 128, 18, 3, 2, 2), (2, 96, 64, 9, 1, 1, 3), (1, 256, 256, 8, 3, 1, 0)])
def test_conv_lds_dma_equals_register_staging(dev, B, Cin, Cout, H, k, stride, tile):
    """The default operand path (global -> LDS by the LDS-DMA, XOR-swizzled rows, zero constant for the padding pixels) against
    the register-staged path of the same kernel (tile_cfg bit 4): same MFMA order, so the outputs must agree bit for bit --
    ragged M (partial last tile), halo pixels, fused epilogue, statistics partials, and the four-class stride-2 data gradient."""
    L = hipabi.lib()
    pad = 1 if k == 3 else 0
    x = torch.from_numpy(det_uniform((B, Cin, H, H), 11, -1, 1))
    w = torch.from_numpy(det_uniform((Cout, Cin, k, k), 12, -1, 1)) * (2.0 / (Cin * k * k)) ** 0.5
    ya, pa = _run_conv(dev, x, w, stride, pad, cfg=tile, stats=True)
    yb, pb = _run_conv(dev, x, w, stride, pad, cfg=tile | 16, stats=True)
    assert torch.equal(ya, yb) and torch.equal(pa, pb)
    sc = torch.from_numpy(det_uniform((Cout,), 13, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((Cout,), 14, -0.5, 0.5))
    res = torch.from_numpy(det_uniform(tuple(ya.shape), 15, -1, 1))
    ya, _ = _run_conv(dev, x, w, stride, pad, sc, sh, res, True, tile)
    yb, _ = _run_conv(dev, x, w, stride, pad, sc, sh, res, True, tile | 16)
    assert torch.equal(ya, yb)
    if Cin % 64 == 0 and not (tile == 1 and Cin % 128):
        Ho = (H + 2 * pad - k) // stride + 1
        dy = torch.from_numpy(det_uniform((B, Ho, Ho, Cout), 16, -1, 1)).to(dev)
        wd = w.contiguous().to(dev)
        wpk = torch.empty_like(wd)
        hipabi.check(L.straps_pack_conv_weight_dgrad(hipabi.ptr(wd), hipabi.ptr(wpk), Cout, Cin, k, k, None), 'pack dgrad')
        outs = []
        for cfg in (tile, tile | 16):
            dx = torch.full((B, H, H, Cin), float('nan'), device=dev)
            hipabi.check(L.straps_conv_dgrad(hipabi.ptr(dy), hipabi.ptr(wpk), None, hipabi.ptr(dx), B, H, H, Cin, Cout, k, k, stride, pad,
                                             cfg, None), 'dgrad')
            outs.append(dx)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())


@pytest.mark.parametrize('B,C,H,W', [(2, 18, 256, 256), (1, 1, 64, 96), (3, 18, 40, 72)])
def test_stem_vs_cpu(dev, B, C, H, W):
    L = hipabi.lib()
    x = torch.from_numpy(det_uniform((B, C, H, W), 6, 0, 1))
    w = torch.from_numpy(det_uniform((64, C, 7, 7), 7, -1, 1)) * (2.0 / (C * 49)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    Ho, Wo = ref.shape[2:]
    xd, wd = x.to(dev), w.to(dev)
    wf = torch.empty(L.straps_stem_weight_floats(C), device=dev)
    hipabi.check(L.straps_pack_stem_weight(hipabi.ptr(wd), hipabi.ptr(wf), C, None), 'pack stem')
    y = torch.empty(B, Ho, Wo, 64, device=dev)
    part = torch.empty(L.straps_stem_stat_blocks(B, H, W), 64, 2, device=dev)
    hipabi.check(L.straps_stem_fwd(hipabi.ptr(xd), hipabi.ptr(wf), None, None, 0, hipabi.ptr(y), hipabi.ptr(part), None, B, C, H, W, None), 'stem')
    _close(y.permute(0, 3, 1, 2), ref, 2e-5, 2e-5, 'stem raw')
    s = part.double().sum(0).cpu()
    np.testing.assert_allclose(s[:, 0].numpy(), ref.sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    sc = torch.from_numpy(det_uniform((64,), 3, 0.5, 1.5))
    sh = torch.from_numpy(det_uniform((64,), 4, -0.5, 0.5))
    scd, shd = sc.to(dev), sh.to(dev)          # keep the device copies alive across the launch
    hipabi.check(L.straps_stem_fwd(hipabi.ptr(xd), hipabi.ptr(wf), hipabi.ptr(scd), hipabi.ptr(shd), 1, hipabi.ptr(y), None, None,
                                   B, C, H, W, None), 'stem fused')
    want = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    _close(y.permute(0, 3, 1, 2), want, 3e-5, 3e-5, 'stem fused')


def test_stem_zero_skipping_is_exact(dev):
    """proxy-like input (mostly exact zeros): probing, mask-driven and skip-defeated runs agree (the first two bit for bit)."""
    L = hipabi.lib()
    B, C, H, W = 3, 18, 96, 128
    x = torch.zeros(B, C, H, W)
    blob = torch.from_numpy(det_uniform((B, C, 16, 16), 30, 0.1, 1.0))
    for b in range(B):
        for c in range(C):
            if (b + c) % 5 == 4:
                continue
            y0, x0 = (7 * c + 13 * b) % (H - 16), (11 * c + 5 * b) % (W - 16)
            x[b, c, y0:y0 + 16, x0:x0 + 16] = blob[b, c]
    x[:, 0, 20:80, 40:90] = 1.0
    w = torch.from_numpy(det_uniform((64, C, 7, 7), 31, -1, 1)) * (2.0 / (C * 49)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    Ho, Wo = ref.shape[2:]
    xd, wd = x.to(dev), w.to(dev)
    wf = torch.empty(L.straps_stem_weight_floats(C), device=dev)
    hipabi.check(L.straps_pack_stem_weight(hipabi.ptr(wd), hipabi.ptr(wf), C, None), 'pack stem')
    mask = torch.empty(L.straps_stem_nzmask_words(B, C, H, W), device=dev, dtype=torch.int32)
    hipabi.check(L.straps_stem_nzmask(hipabi.ptr(xd), hipabi.ptr(mask), B, C, H, W, None), 'nzmask')
    cells = (x.reshape(B, C, H // 4, 4, W // 8, 8) != 0).any(5).any(3)                      # [B,C,H/4,W/8]
    bits = (mask.cpu().view(B, C, H // 4, 1).to(torch.int64) & 0xffffffff)
    got = torch.stack([(bits[..., 0] >> k) & 1 for k in range(W // 8)], dim=-1).bool()
    assert torch.equal(got, cells)
    outs, parts = [], []
    for inp, m in ((xd, None), (xd, mask), (torch.where(xd == 0, torch.full_like(xd, 1e-30), xd), None)):
        y = torch.empty(B, Ho, Wo, 64, device=dev)
        part = torch.empty(L.straps_stem_stat_blocks(B, H, W), 64, 2, device=dev)
        hipabi.check(L.straps_stem_fwd(hipabi.ptr(inp), hipabi.ptr(wf), None, None, 0, hipabi.ptr(y), hipabi.ptr(part), hipabi.ptr(m),
                                       B, C, H, W, None), 'stem')
        outs.append(y)
        parts.append(part)
    assert torch.equal(outs[0], outs[1]) and torch.equal(parts[0], parts[1])
    assert float((outs[0] - outs[2]).abs().max()) < 1e-20
    _close(outs[0].permute(0, 3, 1, 2), ref, 2e-5, 2e-5, 'stem sparse')
    assert float((outs[0] == 0).float().mean()) > 0.05            # there really are untouched output regions


def test_batched_weight_pack_equals_single_packs(dev):
    """straps_pack_conv_weights_batched (LDS-tiled transposes, all layers in one launch) against straps_pack_conv_weight and
    straps_pack_conv_weight_dgrad layer by layer: 1x1 / 3x3 / 5x5 taps, channel counts that are not multiples of the 32x32 tile,
    a layer without the data-gradient layout."""
    import numpy as np
    L = hipabi.lib()
    shapes = [(64, 64, 3, 3), (128, 64, 1, 1), (96, 40, 3, 3), (33, 70, 5, 5), (256, 128, 3, 3), (8, 8, 1, 1)]
    ws = [torch.from_numpy(det_uniform(sh, 40 + i, -1, 1)).to(dev) for i, sh in enumerate(shapes)]
    total = sum(w.numel() for w in ws)
    krsc = torch.full((total,), float('nan'), device=dev)
    crsk = torch.full((total,), float('nan'), device=dev)
    descs = (hipabi.PackDesc * len(ws))()
    off = 0
    for i, (d, w) in enumerate(zip(descs, ws)):
        d.src, d.dst_krsc = w.data_ptr(), krsc.data_ptr() + 4 * off
        d.dst_crsk = None if i == 2 else crsk.data_ptr() + 4 * off
        d.o, d.c, d.r, d.s, d.first = w.shape[0], w.shape[1], w.shape[2], w.shape[3], off
        off += w.numel()
    table = torch.from_numpy(np.frombuffer(bytes(descs), dtype=np.uint8).copy()).to(dev)
    hipabi.check(L.straps_pack_conv_weights_batched(hipabi.ptr(table), len(ws), total, None), 'batched pack')
    off = 0
    for i, w in enumerate(ws):
        O, C, R, S = w.shape
        a, b = torch.empty_like(w), torch.empty_like(w)
        hipabi.check(L.straps_pack_conv_weight(hipabi.ptr(w), hipabi.ptr(a), O, C, R, S, None), 'pack')
        hipabi.check(L.straps_pack_conv_weight_dgrad(hipabi.ptr(w), hipabi.ptr(b), O, C, R, S, None), 'pack dgrad')
        n = w.numel()
        assert torch.equal(krsc[off:off + n], a.reshape(-1)), shapes[i]
        assert torch.equal(a.reshape(O, R, S, C), w.permute(0, 2, 3, 1))
        if i == 2:
            assert bool(torch.isnan(crsk[off:off + n]).all())
        else:
            assert torch.equal(crsk[off:off + n], b.reshape(-1)), shapes[i]
        off += n


def test_pool_gap_bn_helpers(dev):
    L = hipabi.lib()
    x = torch.from_numpy(det_uniform((3, 64, 17, 22), 8, -1, 1))
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ref = F.max_pool2d(x, 3, 2, 1)
    y = torch.empty(3, ref.shape[2], ref.shape[3], 64, device=dev)
    hipabi.check(L.straps_maxpool_fwd(hipabi.ptr(xn), hipabi.ptr(y), 3, 17, 22, 64, None), 'maxpool')
    assert torch.equal(y.permute(0, 3, 1, 2).cpu(), ref)                       # max is exact
    g = torch.empty(3, 64, device=dev)
    hipabi.check(L.straps_gap_fwd(hipabi.ptr(xn), hipabi.ptr(g), 3, 17 * 22, 64, None), 'gap')
    _close(g, x.double().mean(dim=(2, 3)), 1e-6, 1e-6, 'gap')
    # BN training statistics: finalize + apply vs F.batch_norm
    C, rows = 64, 3 * 17 * 22
    gamma = torch.from_numpy(det_uniform((C,), 9, 0.5, 1.5))
    beta = torch.from_numpy(det_uniform((C,), 10, -0.5, 0.5))
    rm, rv = torch.zeros(C), torch.ones(C)
    want = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    flat = xn.view(rows, C)
    part = torch.stack([flat.sum(0), (flat * flat).sum(0)], dim=1)[None].contiguous()      # one "block"
    ss = torch.empty(4, C, device=dev)
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    hipabi.check(L.straps_bn_stats_finalize(hipabi.ptr(part), 1, C, rows, hipabi.ptr(gd), hipabi.ptr(bd), 1e-5, 0.1,
                                            hipabi.ptr(rmd), hipabi.ptr(rvd), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), hipabi.ptr(ss[2]),
                                            hipabi.ptr(ss[3]), None), 'finalize')
    out = torch.empty_like(xn)
    hipabi.check(L.straps_bn_apply(hipabi.ptr(xn), hipabi.ptr(ss[0]), hipabi.ptr(ss[1]), None, 0, hipabi.ptr(out), rows, C, None), 'apply')
    _close(out.permute(0, 3, 1, 2), want, 2e-5, 2e-5, 'bn train apply')
    _close(rmd, rm, 1e-6, 1e-5, 'running_mean')
    _close(rvd, rv, 1e-6, 1e-5, 'running_var')


# ------------------------------------------------------------------------------------------ IEF / full nets
def _load_det(reg, layers, dev):
    man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r%d.json' % layers)))['keys']
    sd = {k: torch.from_numpy(v) for k, v in det_state_dict(man).items()}
    reg.load_state_dict(sd, strict=True)
    return reg.to(dev), sd


@pytest.mark.parametrize('layers,F_', [(18, 512), (50, 2048)])
def test_ief_vs_reference_golden(dev, layers, F_):
    gold = np.load(os.path.join(GOLD, 'small_golden.npz'))
    reg, sd = _load_det(straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP), layers, dev)
    feat = torch.from_numpy(det_uniform((4, F_), 31, 0.0, 2.0))
    with torch.no_grad():
        cam, pose, shape = reg.ief_module(feat.to(dev))
    assert cam.shape == (4, 3) and pose.shape == (4, 144) and shape.shape == (4, 10)
    assert not pose.is_contiguous() and cam.data_ptr() == pose.data_ptr() - 12      # views of one buffer (:60-62)
    _close(torch.cat([cam, pose, shape], 1), torch.from_numpy(gold['ief_r%d_out' % layers]), 2e-5, 2e-5, 'IEF vs golden')
    # ragged batch (not a multiple of 32) vs oracle
    feat2 = torch.from_numpy(det_uniform((37, F_), 77, 0.0, 2.0))
    with torch.no_grad():
        out = torch.cat(reg.ief_module(feat2.to(dev)), 1)
        _, _, _, est = O.ief_forward(feat2, sd, O.ief_init_estimate(MP['pose'], MP['shape']), 3)
    _close(out, est, 2e-5, 2e-5, 'IEF ragged vs oracle')


@pytest.mark.parametrize('layers', [18, 50])
def test_regressor_eval_vs_reference_golden(dev, layers):
    """whole regressor on the committed golden vectors captured from the reference (eval mode)."""
    gold = np.load(os.path.join(GOLD, 'encoder_golden.npz'))
    reg, sd = _load_det(straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP), layers, dev)
    reg.eval()
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 4242, 0.0, 1.0)).to(dev)
    with torch.no_grad():
        feat = reg.image_encoder(x)
        cam, pose, shape = reg(x)
    tag = 'r%d_eval_' % layers
    e1 = _close(feat, torch.from_numpy(gold[tag + 'feat_full']), 2e-4, 2e-4, 'features vs golden')
    e2 = _close(torch.cat([cam, pose, shape], 1), torch.from_numpy(gold[tag + 'out']), 2e-4, 2e-4, 'outputs vs golden')
    print('r%d eval: feature err %.2e, output err %.2e' % (layers, e1, e2))


@pytest.mark.parametrize('layers', [18, 50])
def test_regressor_train_mode_forward_vs_reference_golden(dev, layers):
    """training-mode BatchNorm (batch statistics + running-stat update) vs the reference golden."""
    gold = np.load(os.path.join(GOLD, 'encoder_golden.npz'))
    reg, sd = _load_det(straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP), layers, dev)
    reg.train()
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 4242, 0.0, 1.0)).to(dev)
    with torch.no_grad():
        cam, pose, shape = reg(x)
    tag = 'r%d_train_' % layers
    _close(torch.cat([cam, pose, shape], 1), torch.from_numpy(gold[tag + 'out']), 5e-4, 5e-4, 'train-mode outputs vs golden')
    sd2 = reg.state_dict()
    for bn in ('image_encoder.bn1', 'image_encoder.layer2.0.downsample.1', 'image_encoder.layer4.1.bn2'):
        _close(sd2[bn + '.running_mean'], torch.from_numpy(gold[tag + bn + '.running_mean']), 1e-5, 1e-4, bn + ' running_mean')
        _close(sd2[bn + '.running_var'], torch.from_numpy(gold[tag + bn + '.running_var']), 1e-5, 1e-4, bn + ' running_var')
        assert int(sd2[bn + '.num_batches_tracked']) == 1


def test_batch64_forward_consistency(dev):
    """BASELINE config 2 shape (B=64): batch rows are independent -- the first 3 rows of a 64-batch
    equal a 3-batch run bit for bit (same tiles, same reduction order) and match the oracle."""
    reg, sd = _load_det(straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP), 18, dev)
    reg.eval()
    x = torch.from_numpy(det_uniform((64, 18, 256, 256), 5151, 0.0, 1.0))
    with torch.no_grad():
        big = torch.cat(reg(x.to(dev)), 1)
        small = torch.cat(reg(x[:3].to(dev)), 1)
        _, _, _, est = O.regressor_forward(x[:3], sd, O.ief_init_estimate(MP['pose'], MP['shape']), 18, 3, False)
    _close(big[:3], small, 1e-5, 1e-5, 'batch independence')
    _close(small, est, 2e-4, 2e-4, 'vs oracle')
    assert torch.isfinite(big).all()


def test_eval_after_train_mode_forward_sees_fresh_running_statistics():
    """the folded-BN cache is keyed on tensor versions, and running statistics are updated through raw pointers: a train-mode
    forward (BN recalibration under no_grad, no optimiser step in between) must still invalidate the eval-mode folds."""
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=straps_amd.synthetic_mean_params(0)).to(dev)
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 71, 0.0, 1.0)).to(dev)
    x2 = torch.from_numpy(det_uniform((2, 18, 256, 256), 72, 0.0, 1.0)).to(dev)
    with torch.no_grad():
        reg.train()
        reg(x)
        reg.eval()
        y1 = torch.cat(reg(x), 1).clone()
        reg.train()
        reg(x2)                                 # running statistics move again, no parameter version changes
        reg.eval()
        y2 = torch.cat(reg(x), 1).clone()
        fresh = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=straps_amd.synthetic_mean_params(0)).to(dev).eval()
        fresh.load_state_dict(reg.state_dict())
        y3 = torch.cat(fresh(x), 1)
    assert not torch.equal(y1, y2)
    assert torch.equal(y2, y3)


def test_calibration_and_utility_entry_points(dev):
    """straps_device_count, straps_wall_clock_khz, the two matrix-pipe calibration kernels (bench.py's roofline.sustained_mfma /
    tools/mfma_peak.py) and the clock accumulator: they run, write every output and report plausible rates (the bf16 stream between
    a fifth of the 2.5 PFLOP/s spec and the spec; the accumulated shader clock between 0.5 and 3 GHz)."""
    L = hipabi.lib()
    assert L.straps_device_count() >= 1
    khz = L.straps_wall_clock_khz()
    assert 1e4 < khz < 1e6                                   # (100 MHz on gfx950)
    blocks, iters = 512, 400
    out = torch.full((blocks * 256,), float('nan'), device=dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)

    def timed(fn):
        fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e-3
    t = timed(lambda: hipabi.check(L.straps_selftest_mfma_bf16(hipabi.ptr(out), hipabi.ptr(clk), blocks, iters, hipabi.stream_ptr()), 'selftest bf16'))
    assert torch.isfinite(out).all()
    rate = blocks * 4 * iters * 48 * 32768.0 / t
    assert 0.5e15 < rate < 2.6e15, rate
    c, w = (int(v) for v in clk.tolist())
    assert w > 0 and 500 < c / w * khz / 1e3 < 3000
    seed = torch.rand(512, device=dev)
    out.fill_(float('nan'))
    t = timed(lambda: hipabi.check(L.straps_selftest_mfma_peak(hipabi.ptr(seed), hipabi.ptr(out), blocks, iters, hipabi.stream_ptr()), 'selftest fp32'))
    assert torch.isfinite(out).all()
    rate32 = blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 2 / t
    assert 0.2e14 < rate32 < 1.6e14, rate32
    # clock accumulator: a convolution launched while it is set adds its ticks; unset afterwards
    acc = torch.zeros(2, dtype=torch.int64, device=dev)
    assert L.straps_set_clock_accumulator(hipabi.ptr(acc)) == 0
    try:
        reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(dev).eval()
        with torch.no_grad():
            reg(torch.rand(2, 18, 256, 256, device=dev))
        torch.cuda.synchronize()
    finally:
        L.straps_set_clock_accumulator(None)
    c, w = (int(v) for v in acc.tolist())
    assert c > 0 and w > 0 and 500 < c / w * khz / 1e3 < 3000


def test_measurement_environment_switches_do_not_reach_the_product_library(smpl_model):
    """STRAPS_SMPL_ABLATE=15 (no stores / loads / MFMAs in the tools build), STRAPS_SMPL_PF, STRAPS_WGRAD3_ABL ... set in the environment
    of a fresh process: the product library ignores them -- SMPL vertices stay float64-exact to the usual bar."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))
import straps_amd, straps_oracle as O
from detgen import det_uniform
dev = torch.device('cuda:0')
model = straps_amd.synthetic_smpl_model(0)
smpl = straps_amd.SMPL(model, batch_size=70).to(dev)
betas = torch.from_numpy(det_uniform((70, 10), 170, -2.5, 2.5))
R = O.batch_rodrigues(torch.from_numpy(det_uniform((70, 72), 270, -0.9, 0.9)).reshape(-1, 3)).view(70, 24, 3, 3)
v, j = smpl.forward_arrays(betas.to(dev), R.to(dev), precision='fp16x3_lbs')
v64, j64 = O.smpl_forward(model, betas.double(), rotmats=R.double(), dtype=torch.float64)
print('ERR %%.3e %%.3e' %% (float((v.cpu().double() - v64).abs().max()), float((j.cpu().double() - j64).abs().max())))
''' % (ROOT, ROOT)
    env = dict(os.environ, STRAPS_SMPL_ABLATE='15', STRAPS_SMPL_PF='1', STRAPS_SMPL_RPC='1', STRAPS_WGRAD3_ABL='2', STRAPS_WGRAD_TAP_FP32='1',
               STRAPS_STEM_WGRAD_ROT='0')
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-800:]
    ev, ej = (float(t) for t in [l for l in p.stdout.splitlines() if l.startswith('ERR')][-1].split()[1:])
    assert ev < 2e-5 and ej < 2e-5, (ev, ej)


@pytest.mark.parametrize('mode', ['fp16x3_lbs', 'fp16x3_lbs_pd16', 'fp16x3_lbs_p16'])
@pytest.mark.parametrize('B', [1, 70, 200, 2085])
def test_smpl_wide_kernel_vs_oracle_and_narrow(dev, smpl_model, B, mode):
    """the 64-body kernel of the matrix-pipe modes (smpl_verts_w_kernel: one 512-register wave per SIMD, every direction fragment feeding two
    body groups, skinning products K-packed, non-temporal buffer stores spread over the next tile's blend phase) forced at every batch size
    -- one ragged group, full + ragged groups, > 2048 bodies where it is the automatic choice: float64 bar of every SMPL test (2e-5 m; the
    north_star bar is 1e-4), BIT-IDENTICAL to the 32-body kernel (both run the same accumulation chains per output value, so the automatic
    switch at 2048 bodies changes no result), vertices-only form, chunked launches, batch rows independent of the batch they ride in."""
    smpl = straps_amd.SMPL(smpl_model, batch_size=B).to(dev)
    betas = torch.from_numpy(det_uniform((B, 10), 100 + B, -2.5, 2.5))
    betas[0] = torch.tensor([10.0, -8.0, 6.0, 4.0, -4.0, 3.0, 3.0, -3.0, 2.0, 2.0])          # an extreme body
    aa = torch.from_numpy(det_uniform((B, 72), 200 + B, -0.9, 0.9))
    aa[-1] = torch.from_numpy(det_uniform((72,), 7, -3.0, 3.0))                             # rotations up to pi per axis component
    R = O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)
    bd, Rd = betas.to(dev), R.to(dev)
    v, j = smpl.forward_arrays(bd, Rd, precision=mode, kernel='wide')
    vn, jn = smpl.forward_arrays(bd, Rd, precision=mode, kernel='narrow')
    idx = sorted(set(list(range(min(B, 48))) + list(range(max(0, B - 48), B))))
    v64, j64 = O.smpl_forward(smpl_model, betas[idx].double(), rotmats=R[idx].double(), dtype=torch.float64)
    ev, ej = float((v[idx].cpu().double() - v64).abs().max()), float((j[idx].cpu().double() - j64).abs().max())
    dn = float((v - vn).abs().max())
    print('SMPL wide kernel B=%d %s: |verts - f64| %.2e  |joints - f64| %.2e  |wide - narrow| %.2e' % (B, mode, ev, ej, dn))
    assert ev < 2e-5 and ej < 2e-5
    assert torch.equal(v, vn) and torch.equal(j, jn)
    assert torch.isfinite(v).all() and torch.isfinite(j).all()
    v2, none = smpl.forward_arrays(bd, Rd, want_joints=False, precision=mode, kernel='wide')
    assert none is None and torch.equal(v2, v)
    v3, j3 = smpl.forward_arrays(bd, Rd, precision=mode, kernel='wide', chunks=3)           # three chunks of tile rounds per body group
    assert torch.equal(v3, v) and torch.equal(j3, j)
    va, ja = smpl.forward_arrays(bd, Rd, precision=mode)                                     # the automatic choice
    assert torch.equal(va, v) and torch.equal(ja, j)
    if mode == 'fp16x3_lbs':
        # the product form issues its skinning chains as inline-assembly MFMAs with VGPR results, whose distance to their first VALU reader is counted by
        # hand (csrc/smpl.hip); the SAME kernel with the compiler's builtin (every hazard resolved by the compiler; ADVICE rounds 3-5) must give the same bits
        vb, jb = smpl.forward_arrays(bd, Rd, precision=mode, kernel='wide_builtin')
        assert torch.equal(vb, v) and torch.equal(jb, j)
    if B > 80:
        vs, js = smpl.forward_arrays(bd[60:71].contiguous(), Rd[60:71].contiguous(), precision=mode, kernel='wide')
        assert torch.equal(vs, v[60:71]) and torch.equal(js, j[60:71])
    with pytest.raises(ValueError):
        smpl.forward_arrays(bd, Rd, precision=mode, kernel='fast')
    with pytest.raises(RuntimeError):
        smpl.forward_arrays(bd, Rd, precision='fp32', kernel='wide')                        # the wide kernel exists for the matrix-pipe modes only
