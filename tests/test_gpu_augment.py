"""GPU: proxy-representation augmentation kernel (straps_augment_seg) and the target projection kernel against
plain restatements of the reference semantics (augmentation/proxy_rep_augmentation.py:52-101, utils/cam_utils.py:40-71)
with the random draws supplied explicitly."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_uniform
from straps_amd import hipabi

pytestmark = pytest.mark.gpu


def test_augment_seg_bit_exact_vs_oracle():
    """byte work -> bit-exact: the kernel and the oracle (pinned by the reference's own generator streams in
    tests/test_oracle_golden.py) consume the SAME float32 draws."""
    dev = torch.device('cuda:0')
    L = hipabi.load()
    B, wh, box = 64, 256, 48
    seg = np.floor(det_uniform((B, wh, wh), 50, 0.0, 6.999)).astype(np.float32)
    u = det_uniform((B, 9), 51, 0.0, 1.0)
    u[0, :7] = 0.0           # sample 0: every part removed and occluded
    u[1, :7] = 0.99          # sample 1: untouched
    u[2, 6], u[2, 7], u[2, 8] = 0.0, 0.0, 1.0 - 2.0 ** -24          # extreme box centres
    u[3, 6], u[3, 7], u[3, 8] = 0.0, 1.0 - 2.0 ** -24, 0.0
    u[4, 6], u[4, 7] = 0.0, np.float32((166.4 - 152.0) / 76.8)      # centre - box/2 lands (almost) on an integer
    u[5, 0] = np.float32(0.1)                                       # a draw equal to float32(prob): not removed
    probs = np.array([0.1, 0.1, 0.1, 0.1, 0.05, 0.05], np.float32)
    want = O.augment_seg(seg, u, remove_probs=probs, occlude_probability=0.5, occlude_box_dim=box)
    segd, ud, pd = torch.from_numpy(seg).to(dev), torch.from_numpy(u).to(dev), torch.from_numpy(probs).to(dev)
    out = torch.empty_like(segd)
    hipabi.check(L.straps_augment_seg(hipabi.ptr(segd), hipabi.ptr(ud), hipabi.ptr(pd), 0.5, box, hipabi.ptr(out), B, wh, None), 'augment')
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert np.array_equal(got[1], seg[1]) and got[0].sum() == 0 and (got != seg).any(axis=(1, 2)).sum() > B // 2
    # module-level entry point (reference signature): inputs untouched, same kernel, draws supplied or drawn on the device
    params = {'remove_appendages': True, 'deviate_joints2D': True, 'deviate_verts2D': True, 'occlude_seg': True,
              'remove_appendages_classes': [1, 2, 3, 4, 5, 6], 'remove_appendages_probabilities': [0.1, 0.1, 0.1, 0.1, 0.05, 0.05],
              'delta_j2d_dev_range': [-8, 8], 'delta_j2d_hip_dev_range': [-15, 15], 'delta_verts2d_dev_range': [-0.01, 0.01],
              'occlude_probability': 0.5, 'occlude_box_dim': 48}
    j = torch.from_numpy(det_uniform((B, 17, 2), 52, 20.0, 236.0)).to(dev)
    j0 = j.clone()
    uj = det_uniform((B, 17, 2), 53, 0.0, 1.0)
    s2, j2 = straps_amd.augmentation.augment_proxy_representation(segd, j, params, seg_uniforms=ud, joint_uniforms=torch.from_numpy(uj).to(dev))
    assert torch.equal(j, j0) and torch.equal(segd.cpu(), torch.from_numpy(seg))
    np.testing.assert_array_equal(s2.cpu().numpy(), want)
    np.testing.assert_array_equal(j2.cpu().numpy(), O.random_joints2D_deviation(j0.cpu(), uj, [-8, 8], [-15, 15]).numpy())
    straps_amd.device_rng.manual_seed(5, dev)
    s3, j3 = straps_amd.augmentation.augment_proxy_representation(segd, j, params)
    straps_amd.device_rng.manual_seed(5, dev)
    s4, j4 = straps_amd.augmentation.augment_proxy_representation(segd, j, params)
    assert torch.equal(s3, s4) and torch.equal(j3, j4) and not torch.equal(j3, j0)
    d = (j3 - j0).abs()
    assert float(d[:, [11, 12]].max()) <= 15.0 and float(d[:, [0, 5, 16]].max()) <= 8.0 and float(d[:, [11, 12]].max()) > 8.0


def test_project_targets_vs_oracle():
    dev = torch.device('cuda:0')
    L = hipabi.load()
    B = 5
    joints = torch.from_numpy(det_uniform((B, 90, 3), 53, -1, 1))
    cam_t = torch.tensor([[0., 0.2, 42.0]]).expand(B, -1) + torch.from_numpy(det_uniform((B, 3), 54, -0.1, 0.1))
    K = torch.from_numpy(O.intrinsics_matrix().astype(np.float32))[None].expand(B, -1, -1)
    want2d = O.perspective_project(joints[:, O.ALL_JOINTS_TO_COCO_MAP], torch.eye(3)[None].expand(B, -1, -1), cam_t, K)
    want3d = joints[:, O.ALL_JOINTS_TO_H36M_MAP][:, O.H36M_TO_J14]
    j2, j3 = torch.empty(B, 17, 2, device=dev), torch.empty(B, 14, 3, device=dev)
    jd, cd = joints.to(dev), cam_t.contiguous().to(dev)
    hipabi.check(L.straps_project_targets(hipabi.ptr(jd), hipabi.ptr(cd), 5000.0, 5000.0, 128.0, 128.0, hipabi.ptr(j2), hipabi.ptr(j3), B, None), 'project')
    assert float((j2.cpu() - want2d).abs().max()) < 2e-3          # pixels; fp32 division / fma order
    assert torch.equal(j3.cpu(), want3d)
    cam = torch.from_numpy(det_uniform((B, 3), 55, 0.5, 1.2)).to(dev)
    got = straps_amd.cam_utils.orthographic_project_torch(jd[:, :17], cam)
    assert float((got.cpu() - O.orthographic_project(joints[:, :17], cam.cpu())).abs().max()) < 1e-6


def test_point_metrics_vs_reference_golden_and_oracle():
    """on-device PVE / PVE-SC / PVE-PA (6890 vertices) and MPJPE / -SC / -PA (14 joints) sums."""
    import os
    from detgen import det_metrics_case
    dev = torch.device('cuda:0')
    small = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'small_golden.npz'))
    for tag, npts, seed in (('verts', 6890, 70), ('j14', 14, 72)):
        pv, tv = det_metrics_case(npts, seed)
        got = straps_amd.metrics.point_error_sums(torch.from_numpy(pv).to(dev), torch.from_numpy(tv).to(dev)).cpu().numpy()
        np.testing.assert_allclose(got, small['metrics_%s_sums' % tag], rtol=5e-5)
        np.testing.assert_allclose(got, O.point_metrics(pv, tv), rtol=5e-5)
    # a reflection-prone case (near-planar points): det(R) must stay +1 like the reference's Z fix
    pv, tv = det_metrics_case(40, 90)
    tv[:, :, 2] *= 1e-3
    pv = tv[:, :, [0, 1, 2]] * np.array([1.0, 1.0, -1.0], np.float32) + 0.01 * pv
    got = straps_amd.metrics.point_error_sums(torch.from_numpy(pv.astype(np.float32)).to(dev), torch.from_numpy(tv).to(dev)).cpu().numpy()
    np.testing.assert_allclose(got, O.point_metrics(pv.astype(np.float32), tv), rtol=1e-3)
    bm = straps_amd.metrics.BatchMetrics(dev)
    B = 3
    pd = {'verts': torch.from_numpy(det_metrics_case(6890, 70)[0]).to(dev), 'joints3D': torch.from_numpy(det_metrics_case(14, 72)[0]).to(dev),
          'joints2D': torch.zeros(B, 17, 2, device=dev), 'shape_params': torch.zeros(B, 10, device=dev),
          'pose_params_rot_matrices': torch.zeros(B, 24, 3, 3, device=dev)}
    td = {'verts': torch.from_numpy(det_metrics_case(6890, 70)[1]).to(dev), 'joints3D': torch.from_numpy(det_metrics_case(14, 72)[1]).to(dev),
          'joints2D': torch.full((B, 17, 2), 128.0, device=dev), 'shape_params': torch.ones(B, 10, device=dev),
          'pose_params_rot_matrices': torch.zeros(B, 24, 3, 3, device=dev)}
    bm.update(pd, td)
    s = bm.summary()
    assert s['pves_pa'] < s['pves_sc'] < s['pves'] and s['shape_mses'] == pytest.approx(1.0) and s['joints2D_l2es'] == pytest.approx(0.0, abs=1e-6)
    assert s['pves'] == pytest.approx(float(small['metrics_verts_sums'][:, 0].sum()) / (3 * 6890), rel=1e-4)


def test_crop_resize_vs_oracle_and_reference_boxes():
    """on-device bbox crop + nearest resize vs the oracle (whose crop half is pinned by the reference golden)."""
    import os
    from detgen import det_crop_case
    dev = torch.device('cuda:0')
    small = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'small_golden.npz'))
    seg, j = det_crop_case()
    np.random.seed(11)
    u = np.zeros((seg.shape[0], 3), np.float32)
    for i in range(seg.shape[0]):
        u[i, 0] = np.random.rand()
        u[i, 1:3] = np.random.rand(2)
    for uni in (u, None):
        kw = dict(delta_scale_range=[-0.2, 0.2], delta_centre_range=[-5, 5], uniforms=torch.from_numpy(u).to(dev)) if uni is not None else {}
        out, jout, boxes = straps_amd.image_utils.batch_crop_and_resize(torch.from_numpy(seg).to(dev), torch.from_numpy(j).to(dev), 256, 1.2, **kw)
        want, wj, wb = O.crop_resize(seg, j, uni.astype(np.float64) if uni is not None else None)
        bx = boxes.cpu().numpy()
        # float32 uniforms vs the float64 draw: a corner can land on the other side of an integer only within 1e-6 of it
        assert np.array_equal(bx[:, :4], wb), (bx, wb)
        shapes = small['crop_shapes'] if uni is not None else small['crop0_shapes']
        assert np.array_equal(np.stack([bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]], 1), shapes)
        assert np.array_equal(out.cpu().numpy(), want)
        np.testing.assert_allclose(jout.cpu().numpy(), wj, rtol=1e-5, atol=1e-3)


def test_cam_utils_projections_vs_reference_golden_and_autograd():
    """the module-level helpers of utils/cam_utils.py on their HIP kernels (csrc/pose.hip): forward against the reference's outputs
    (tests/golden/small_golden.npz: ortho_out / persp_out, made by importing the reference), a general rotation + per-body and shared
    intrinsics against the oracle, a strided camera view (est[:, :3] of the [B,160] estimate buffer), 6890 points (predict_3D.py:144), and
    the orthographic projection's gradient against autograd of the same expression in float64."""
    import os
    dev = torch.device('cuda:0')
    small = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'small_golden.npz'))
    pts = torch.from_numpy(det_uniform((3, 17, 3), 33, -1.0, 1.0))
    cam = torch.from_numpy(det_uniform((3, 3), 34, 0.5, 1.2))
    got = straps_amd.cam_utils.orthographic_project_torch(pts.to(dev), cam.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), small['ortho_out'], rtol=1e-6, atol=1e-7)
    K = torch.from_numpy(O.intrinsics_matrix().astype(np.float32))[None].expand(3, -1, -1)
    tr = torch.tensor([[0., 0.2, 42.0]]).expand(3, -1) + torch.from_numpy(det_uniform((3, 3), 35, -0.1, 0.1))
    R = torch.eye(3)[None].expand(3, -1, -1)
    got = straps_amd.cam_utils.perspective_project_torch(pts.to(dev), R.to(dev), tr.to(dev), cam_K=K.to(dev))
    np.testing.assert_allclose(got.cpu().numpy(), small['persp_out'], rtol=2e-5, atol=2e-3)       # pixels (focal length 5000: fp32 division / fma order)
    # general rotation, many points, intrinsics from (focal_length, img_wh)
    B, N = 4, 6890
    P = torch.from_numpy(det_uniform((B, N, 3), 91, -1.0, 1.0))
    Rg = O.batch_rodrigues(torch.from_numpy(det_uniform((B, 3), 92, -1.0, 1.0)))
    tg = torch.tensor([[0., 0.2, 12.0]]).expand(B, -1) + torch.from_numpy(det_uniform((B, 3), 93, -0.5, 0.5))
    Kg = torch.from_numpy(O.intrinsics_matrix(256, 256, 5000.0).astype(np.float32))[None].expand(B, -1, -1)
    want = O.perspective_project(P.double(), Rg.double(), tg.double(), Kg.double())
    got = straps_amd.cam_utils.perspective_project_torch(P.to(dev), Rg.to(dev), tg.to(dev), focal_length=5000.0, img_wh=256)
    assert float((got.cpu().double() - want).abs().max()) < 5e-3                                  # pixels, values up to ~700
    got = straps_amd.cam_utils.perspective_project_torch(P.to(dev), Rg.to(dev), tg.to(dev), cam_K=Kg.contiguous().to(dev))
    assert float((got.cpu().double() - want).abs().max()) < 5e-3
    # strided camera rows + gradient
    est = torch.from_numpy(det_uniform((B, 160), 94, 0.5, 1.2)).to(dev).requires_grad_()
    Pg = P.to(dev).requires_grad_()
    out = straps_amd.cam_utils.orthographic_project_torch(Pg, est[:, :3])
    w = torch.from_numpy(det_uniform((B, N, 2), 95, -1.0, 1.0)).to(dev)
    (out * w).sum().backward()
    P64, c64 = P.double().requires_grad_(), est.detach().cpu().double()[:, :3].clone().requires_grad_()
    (O.orthographic_project(P64, c64) * w.cpu().double()).sum().backward()
    assert float((out.detach().cpu().double() - O.orthographic_project(P64, c64).detach()).abs().max()) < 1e-6
    assert float((Pg.grad.cpu().double() - P64.grad).abs().max()) < 1e-6
    g = est.grad.cpu().double()
    assert float((g[:, :3] - c64.grad).abs().max() / c64.grad.abs().max()) < 2e-6 and float(g[:, 3:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        straps_amd.cam_utils.orthographic_project_torch(P, cam)                                    # CPU tensors: no fallback


@pytest.mark.parametrize('std', [2, 4, 6])
def test_heatmaps_with_other_standard_deviations_vs_oracle(std):
    """convert_2Djoints_to_gaussian_heatmaps_torch(..., std): the reference takes the standard deviation (utils/label_conversions.py:90);
    patch = 4 std x 4 std samples, truncated 2 std from the joint, borders and skipped joints as in the std = 4 golden test."""
    dev = torch.device('cuda:0')
    j = torch.from_numpy(det_uniform((3, 17, 2), 96, -20.0, 276.0))
    j[0, 0] = torch.tensor([0.0, 255.0])
    j[0, 1] = torch.tensor([255.9, 0.2])
    j[0, 2] = torch.tensor([-2.0 * std + 0.5, 100.0])
    j[0, 3] = torch.tensor([255.0 + 2 * std - 1, 128.0])
    want = O.joints2d_to_heatmaps(j, 256, std)
    got = straps_amd.label_conversions.convert_2Djoints_to_gaussian_heatmaps_torch(j.to(dev), 256, std).cpu()
    assert torch.equal(got != 0, want != 0)                        # the same support, pixel for pixel
    assert float((got - want).abs().max()) < 1e-6
    with pytest.raises(ValueError):
        straps_amd.label_conversions.convert_2Djoints_to_gaussian_heatmaps_torch(j.to(dev), 256, 2.5)


def test_memset_zero_is_exact_for_any_alignment_and_size():
    """straps_memset_zero (a fill kernel since round 3: memset nodes misbehave inside captured hipGraphs) clears exactly the requested bytes:
    unaligned starts, sizes below / at / above the 16-byte vector width, zero bytes, and a size with a partial last block."""
    import ctypes as C
    import torch
    from straps_amd import hipabi
    L = hipabi.lib()
    dev = torch.device('cuda:0')
    for off, n in ((0, 0), (0, 1), (3, 15), (5, 16), (1, 17), (3, 777), (16, 4096), (7, 1 << 20), (0, (1 << 22) + 12)):
        buf = torch.full((off + n + 40,), 7, dtype=torch.uint8, device=dev)
        hipabi.check(L.straps_memset_zero(C.c_void_p(buf.data_ptr() + off), n, hipabi.stream_ptr()), 'straps_memset_zero')
        torch.cuda.synchronize()
        assert bool((buf[:off] == 7).all()) and bool((buf[off:off + n] == 0).all()) and bool((buf[off + n:] == 7).all()), (off, n)
