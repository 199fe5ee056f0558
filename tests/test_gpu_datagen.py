"""GPU: the random half of the training step (SURVEY 8a rows G1, G2, G3 and the data-generation part of T) against the
oracle: the Philox draw buffers, every augmentation kernel with the draws supplied, the rasteriser's fused vertex noise,
and `TrainStep.make_batch` stage by stage on identical inputs (train loop :112-182)."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_uniform
from straps_amd import hipabi
from straps_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
MP = straps_amd.synthetic_mean_params(0)
W = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
LOSSES = ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']


def test_philox_fill_vs_oracle():
    """integer work -> the uniform buffer is bit-exact; the normal buffer goes through logf / sinf / cosf (1e-5 abs)."""
    g = straps_amd.device_rng.DeviceDraws((7 << 40) + 1234, DEV)
    for step in (0, 3, (1 << 32) + 5):
        g.set_step(step)
        for n in (1, 7, 4096, 100003):
            u = g.uniform(n, substream=0).cpu().numpy()
            np.testing.assert_array_equal(u, O.philox_uniform(g.seed, step, 0, n))
            z = g.normal(n, substream=1).cpu().numpy()
            np.testing.assert_allclose(z, O.philox_normal(g.seed, step, 1, n), rtol=0, atol=2e-5)
    g.set_step(10)
    g.advance()
    g.advance(2)
    assert g.step() == 13
    a = g.uniform(64, substream=2)
    assert torch.equal(a, g.uniform(64, substream=2)) and not torch.equal(a, g.uniform(64, substream=3))
    # unaligned destination (a view at an odd offset) takes the scalar tail path
    buf = torch.zeros(70, device=DEV)
    g.fill(buf[1:66], 0, 0)
    np.testing.assert_array_equal(buf[1:66].cpu().numpy(), O.philox_uniform(g.seed, 13, 0, 65))
    assert float(buf[0]) == 0 and float(buf[66:].abs().sum()) == 0


def test_augment_smpl_and_cam_kernels_vs_oracle():
    """G1 / G2 with the draws supplied: shape and camera translation bit-exact, rotation matrices to sin/cos accuracy."""
    B = 37
    mean_shape = torch.from_numpy(MP['shape'])
    pose = torch.from_numpy(det_uniform((B, 72), 301, -0.6, 0.6))
    pose[3, 3:6] = 0.0                                              # a zero axis-angle row: the 1e-8 path of batch_rodrigues
    zn = torch.from_numpy(O.philox_normal(11, 0, 1, B * 10)).view(B, 10)
    zu = torch.from_numpy(O.philox_uniform(11, 0, 0, B * 10)).view(B, 10)
    orig = torch.from_numpy(det_uniform((B, 10), 302, -1, 1))
    std = [1.5, 1.5, 1.5, 1.5, 1.5, 1.0, 0.5, 2.0, 1.5, 1.5]
    for params, draws in (({'augment_shape': True, 'delta_betas_distribution': 'normal', 'delta_betas_std_vector': std, 'delta_betas_range': [-3., 3.]}, zn),
                          ({'augment_shape': True, 'delta_betas_distribution': 'uniform', 'delta_betas_std_vector': None, 'delta_betas_range': [-3., 3.]}, zu),
                          ({'augment_shape': False, 'delta_betas_distribution': 'normal', 'delta_betas_std_vector': None, 'delta_betas_range': [-3., 3.]}, None)):
        pd = pose.to(DEV)
        got_shape, got_pose_rot, got_glob = straps_amd.augmentation.augment_smpl(orig.to(DEV), pd[:, 3:], pd[:, :3], mean_shape.to(DEV), params,
                                                                                 shape_draws=None if draws is None else draws.to(DEV))
        want_shape, want_pose_rot, want_glob = O.augment_smpl(orig, pose[:, 3:], pose[:, :3], mean_shape, params, shape_draws=draws)
        np.testing.assert_array_equal(got_shape.cpu().numpy(), want_shape.numpy())
        assert tuple(got_pose_rot.shape) == (B, 23, 3, 3) and tuple(got_glob.shape) == (B, 1, 3, 3)
        assert float((got_pose_rot.cpu() - want_pose_rot).abs().max()) < 2e-6 and float((got_glob.cpu() - want_glob).abs().max()) < 2e-6
    # separate (non-view) pose / global-orientation tensors take the concatenating path: same result
    s2, r2, g2 = straps_amd.augmentation.augment_smpl(orig.to(DEV), pose[:, 3:].contiguous().to(DEV), pose[:, :3].contiguous().to(DEV),
                                                      mean_shape.to(DEV), params)
    assert torch.equal(r2, got_pose_rot) and torch.equal(g2, got_glob) and torch.equal(s2.cpu(), orig)
    # sampling helpers (reference names) + the drawn-on-device path is reproducible under manual_seed
    np.testing.assert_array_equal(straps_amd.augmentation.normal_sample_shape(B, mean_shape.to(DEV), torch.tensor(std), normals=zn.to(DEV)).cpu().numpy(),
                                  O.sample_shape(mean_shape, zn, 'normal', std_vector=std).numpy())
    np.testing.assert_array_equal(straps_amd.augmentation.uniform_sample_shape(B, mean_shape.to(DEV), [-3., 3.], uniforms=zu.to(DEV)).cpu().numpy(),
                                  O.sample_shape(mean_shape, zu, 'uniform', delta_betas_range=[-3., 3.]).numpy())
    # the raw kernel with a pose pool + index draw (the dataset stand-in of the training step)
    L = hipabi.load()
    pool = torch.from_numpy(det_uniform((50, 72), 303, -0.5, 0.5)).to(DEV)
    ui = torch.from_numpy(np.concatenate([[0.0, 1.0 - 2.0 ** -24, 0.5], O.philox_uniform(3, 1, 0, B - 3)]).astype(np.float32)).to(DEV)
    shp, rot, gathered = torch.empty(B, 10, device=DEV), torch.empty(B, 24, 3, 3, device=DEV), torch.empty(B, 72, device=DEV)
    hipabi.check(L.straps_augment_smpl(hipabi.ptr(pool), 50, hipabi.ptr(ui), None, hipabi.ptr(mean_shape.to(DEV)), hipabi.ptr(zn.to(DEV)), 1,
                                       hipabi.ptr(torch.tensor(std, device=DEV)), 0.0, 0.0, hipabi.ptr(shp), hipabi.ptr(rot), hipabi.ptr(gathered),
                                       B, None), 'augment_smpl')
    idx = np.minimum((ui.cpu().numpy() * np.float32(50)).astype(np.int64), 49)
    assert idx[0] == 0 and idx[1] == 49 and idx[2] == 25
    np.testing.assert_array_equal(gathered.cpu().numpy(), pool.cpu().numpy()[idx])
    want_rot = O.batch_rodrigues(pool.cpu()[idx].reshape(-1, 3)).view(B, 24, 3, 3)
    assert float((rot.cpu() - want_rot).abs().max()) < 2e-6
    # G2
    mean_cam_t = torch.tensor([[0., 0.2, 42.0]]).expand(B, -1).contiguous()
    nx, uz = torch.from_numpy(O.philox_normal(12, 0, 1, 2 * B)).view(B, 2), torch.from_numpy(O.philox_uniform(12, 0, 0, B))
    got = straps_amd.augmentation.augment_cam_t(mean_cam_t.to(DEV), xy_std=0.05, delta_z_range=[-5, 5], normals_xy=nx.to(DEV), uniform_z=uz.to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), O.augment_cam_t(mean_cam_t, nx, uz, 0.05, [-5, 5]).numpy())
    straps_amd.device_rng.manual_seed(77, DEV)
    a = straps_amd.augmentation.augment_cam_t(mean_cam_t.to(DEV))
    b = straps_amd.augmentation.augment_cam_t(mean_cam_t.to(DEV))
    straps_amd.device_rng.manual_seed(77, DEV)
    assert torch.equal(a, straps_amd.augmentation.augment_cam_t(mean_cam_t.to(DEV))) and not torch.equal(a, b)
    assert float((a[:, 2] - 42.0).abs().max()) <= 5.0 and float((a[:, :2] - mean_cam_t[:, :2].to(DEV)).abs().max()) < 0.4


def test_vertex_noise_materialised_and_fused_into_the_rasteriser():
    """random_verts2D_deviation: the materialised copy is bit-exact vs the oracle, and the rasteriser fed the clean mesh +
    the uniforms renders EXACTLY what it renders from the noisy copy (and what the oracle's rasteriser does)."""
    model = straps_amd.synthetic_smpl_model(0)
    B = 3
    smpl = straps_amd.SMPL(model, batch_size=B).to(DEV)
    betas = torch.from_numpy(det_uniform((B, 10), 910, -1.5, 1.5)).to(DEV)
    aa = det_uniform((B, 24, 3), 911, -0.3, 0.3)
    R = straps_amd.batch_rodrigues(torch.from_numpy(aa).reshape(-1, 3).to(DEV)).view(B, 24, 3, 3)
    verts, _ = smpl.forward_arrays(betas, R.contiguous())
    u = torch.from_numpy(O.philox_uniform(21, 4, 0, B * 6890 * 2)).view(B, 6890, 2)
    rng = [-0.01, 0.01]
    noisy = straps_amd.augmentation.random_verts2D_deviation(verts, rng, uniforms=u.to(DEV))
    want_noisy = O.random_verts2D_deviation(verts.cpu(), u, rng)
    np.testing.assert_array_equal(noisy.cpu().numpy(), want_noisy.numpy())
    assert torch.equal(noisy[:, :, 2], verts[:, :, 2]) and 0.005 < float((noisy - verts).abs().max()) <= 0.01
    cam_t = torch.tensor([[0.0, 0.2, 42.0], [0.05, 0.15, 38.0], [-0.05, 0.25, 46.0]], device=DEV)
    K = O.intrinsics_matrix().astype(np.float32)
    r = straps_amd.NMRRenderer(B, K, np.eye(3, dtype=np.float32), 256, rend_parts_seg=True, faces=smpl.faces, face_parts=smpl.face_parts).to(DEV)
    fused = r.render_arrays(verts, cam_t, vert_noise_u=u.to(DEV).contiguous(), noise_range=rng)
    from_copy = r.render_arrays(noisy, cam_t)
    clean = r.render_arrays(verts, cam_t)
    assert torch.equal(fused, from_copy)
    assert not torch.equal(fused, clean) and float((fused != clean).float().mean()) < 0.15          # only silhouette / part edges move (+-1.2 px)
    want = O.rasterize_parts(want_noisy.numpy(), model['faces'], model['face_parts'], K, np.eye(3), cam_t.cpu().numpy())
    np.testing.assert_array_equal(fused.cpu().numpy(), want)


def _step(B, seed=0, **kw):
    torch.manual_seed(seed)
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(DEV).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(DEV)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(LOSSES, init_loss_weights=W, reduction='mean').to(DEV)
    return TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=4321, **kw), smpl


@pytest.mark.parametrize('uniform_betas', [False, True])
def test_make_batch_stage_by_stage_vs_oracle(uniform_betas):
    """T (data-generation half): each stage of make_batch against the oracle on the stage's own inputs -- integer / byte
    stages bit-exact, floating-point stages to fp32 rounding."""
    B = 5
    kw = {}
    if uniform_betas:
        kw['smpl_augment_params'] = {'augment_shape': True, 'delta_betas_distribution': 'uniform', 'delta_betas_std_vector': None,
                                     'delta_betas_range': [-3., 3.]}
    ts, smpl = _step(B, **kw)
    model = straps_amd.synthetic_smpl_model(0)
    ts.draws.set_step(6)
    keep = {}
    with torch.no_grad():
        batch = ts.make_batch(keep=keep)
    torch.cuda.synchronize()
    assert ts.draws.step() == 7                                   # one step of the generator per batch
    lay = ts.draw_layout()
    U, N = keep['uniforms'].cpu().numpy(), keep['normals'].cpu().numpy()
    np.testing.assert_array_equal(U, O.philox_uniform(4321, 6, 0, lay['n_uniform']))
    np.testing.assert_allclose(N, O.philox_normal(4321, 6, 1, lay['n_normal']), rtol=0, atol=2e-5)

    def useg(name):
        o, n = lay['uniform'][name]
        return U[o:o + n]

    def nseg(name):
        o, n = lay['normal'][name]
        return N[o:o + n]
    # G1: pose-pool row, shape, rotation matrices
    pool = ts.pose_pool.cpu()
    idx = np.minimum((useg('pose_index') * np.float32(pool.shape[0])).astype(np.int64), pool.shape[0] - 1)
    if uniform_betas:
        want_shape = O.sample_shape(MP['shape'], useg('shape_uniform').reshape(B, 10), 'uniform', delta_betas_range=[-3., 3.])
    else:
        want_shape = O.sample_shape(MP['shape'], nseg('shape').reshape(B, 10), 'normal', std_vector=[1.5] * 10)
    np.testing.assert_array_equal(batch['shape'].cpu().numpy(), want_shape.numpy())
    want_rot = O.batch_rodrigues(pool[idx].reshape(-1, 3)).view(B, 24, 3, 3)
    assert float((batch['rot'].cpu() - want_rot).abs().max()) < 2e-6
    # G2
    want_cam = O.augment_cam_t(torch.tensor([[0., 0.2, 42.0]]).expand(B, -1), nseg('cam_xy').reshape(B, 2), useg('cam_z'), 0.05, [-5, 5])
    np.testing.assert_array_equal(batch['cam_t'].cpu().numpy(), want_cam.numpy())
    # SMPL #1 / #2 on identical (beta, R)
    ov, oj = O.smpl_forward(model, batch['shape'].cpu(), rotmats=batch['rot'].cpu())
    assert float((batch['verts'].cpu() - ov).abs().max()) < 1e-5 and float((keep['joints'].cpu() - oj).abs().max()) < 1e-5
    orv, _ = O.smpl_forward(model, batch['shape'].cpu(), rotmats=torch.eye(3).expand(B, 24, 3, 3))
    assert float((batch['reposed'].cpu() - orv).abs().max()) < 1e-5
    # P2 + H36M-LSP joints on the GPU's own joints
    jg = keep['joints'].cpu()
    K = torch.from_numpy(O.intrinsics_matrix().astype(np.float32))[None].expand(B, -1, -1)
    want2d = O.perspective_project(jg[:, O.ALL_JOINTS_TO_COCO_MAP], torch.eye(3)[None].expand(B, -1, -1), batch['cam_t'].cpu(), K)
    assert float((keep['joints2d_uncropped'].cpu() - want2d).abs().max()) < 2e-3
    assert torch.equal(batch['joints3d'].cpu(), jg[:, O.ALL_JOINTS_TO_H36M_MAP][:, O.H36M_TO_J14])
    # vertex noise + rasteriser: bit-exact on the GPU's vertices and the step's uniforms
    noisy = O.random_verts2D_deviation(batch['verts'].cpu(), useg('verts2d').reshape(B, 6890, 2), [-0.01, 0.01])
    want_seg = O.rasterize_parts(noisy.numpy(), model['faces'], model['face_parts'], O.intrinsics_matrix().astype(np.float32), np.eye(3),
                                 batch['cam_t'].cpu().numpy())
    np.testing.assert_array_equal(keep['seg'].cpu().numpy(), want_seg)
    clean_seg = O.rasterize_parts(batch['verts'].cpu().numpy(), model['faces'], model['face_parts'], O.intrinsics_matrix().astype(np.float32),
                                  np.eye(3), batch['cam_t'].cpu().numpy())
    assert (clean_seg != want_seg).any()                           # the noise really reached the silhouette edges
    # crop + resize (reference float64 box arithmetic on the float32 draws), joints follow the crop
    wc, wj, wb = O.crop_resize(want_seg, keep['joints2d_uncropped'].cpu().numpy(), useg('crop').reshape(B, 3).astype(np.float64))
    np.testing.assert_array_equal(keep['boxes'].cpu().numpy()[:, :4], wb)
    np.testing.assert_array_equal(keep['seg_cropped'].cpu().numpy(), wc)
    np.testing.assert_allclose(batch['joints2d'].cpu().numpy(), wj, rtol=1e-5, atol=1e-3)
    # G3
    want_aug = O.augment_seg(wc, useg('seg').reshape(B, 9))
    np.testing.assert_array_equal(keep['seg_aug'].cpu().numpy(), want_aug)
    want_jin = O.random_joints2D_deviation(batch['joints2d'].cpu(), useg('joints2d').reshape(B, 17, 2), [-8, 8], [-8, 8])
    np.testing.assert_array_equal(keep['joints2d_input'].cpu().numpy(), want_jin.numpy())
    # G4 + G5
    want_x = O.build_proxy_input(torch.from_numpy(want_aug), want_jin).numpy()
    got_x = batch['input'].cpu().numpy()
    assert np.array_equal(got_x != 0, want_x != 0)
    np.testing.assert_allclose(got_x, want_x, rtol=0, atol=2e-6)
    assert np.array_equal(got_x[:, 0], (want_aug != 0).astype(np.float32))
    # a second batch continues the generator, and a fresh step object with the same seed repeats the first
    with torch.no_grad():
        b2 = ts.make_batch()
    assert not torch.equal(b2['shape'], batch['shape'])
    ts2, _ = _step(B, **kw)
    ts2.draws.set_step(6)
    with torch.no_grad():
        b3 = ts2.make_batch()
    for k in ('input', 'verts', 'joints2d', 'joints3d', 'shape', 'rot', 'reposed', 'cam_t'):
        assert torch.equal(b3[k], batch[k]), k


def test_make_batch_switches_follow_run_train_dictionaries():
    """the flags of run_train.py:133-190: every augmentation can be turned off, and then the stage is the identity."""
    B = 4
    off = {'remove_appendages': False, 'deviate_joints2D': False, 'deviate_verts2D': False, 'occlude_seg': False,
           'remove_appendages_classes': [1, 2, 3, 4, 5, 6], 'remove_appendages_probabilities': [0.1] * 6, 'delta_j2d_dev_range': [-8, 8],
           'delta_j2d_hip_dev_range': [-8, 8], 'delta_verts2d_dev_range': [-0.01, 0.01], 'occlude_probability': 0.5, 'occlude_box_dim': 48}
    ts, smpl = _step(B, proxy_rep_augment_params=off, bbox_augment_params={'crop_input': False, 'mean_scale_factor': 1.2,
                                                                             'delta_scale_range': [-0.2, 0.2], 'delta_centre_range': [-5, 5]},
                     smpl_augment_params={'augment_shape': False, 'delta_betas_distribution': 'normal', 'delta_betas_std_vector': None,
                                          'delta_betas_range': [-3., 3.]})
    keep = {}
    with torch.no_grad():
        batch = ts.make_batch(keep=keep)
    assert torch.equal(batch['shape'], ts.mean_shape[None].expand(B, 10))
    assert torch.equal(keep['seg_aug'], keep['seg']) and keep['boxes'] is None
    assert torch.equal(batch['joints2d'], keep['joints2d_uncropped']) and torch.equal(keep['joints2d_input'], batch['joints2d'])
    model = straps_amd.synthetic_smpl_model(0)
    want_seg = O.rasterize_parts(batch['verts'].cpu().numpy(), model['faces'], model['face_parts'], O.intrinsics_matrix().astype(np.float32), np.eye(3),
                                 batch['cam_t'].cpu().numpy())
    np.testing.assert_array_equal(keep['seg'].cpu().numpy(), want_seg)


def test_rasteriser_is_reproducible_beside_convolution_kernels():
    """Round 4 (profiles/r04_raster_determinism.txt): the part rasteriser must give the same z-buffer, bit for bit, whatever runs on
    another stream.  With its pixel-centre table in LDS it did not -- a few keys per launch differed whenever bf16x3 convolution kernels
    (a replayed hipGraph of them, so that the chip stays full) ran beside it, which is exactly where the training step's data stream puts
    it; two replays of a resnet50 step then differed in one 60-step run of three.  300 launches beside such a graph: every part map and
    every projected vertex equals the first launch's."""
    from straps_amd.encoder_exec import split3, weight_planes
    from straps_amd.nmr_renderer import NMRRenderer
    from straps_amd import config
    L = hipabi.lib()
    B = 4
    g = torch.Generator().manual_seed(0)
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=1).to(DEV)
    betas = torch.randn(B, 10, generator=g).to(DEV)
    R = straps_amd.batch_rodrigues((torch.randn(B, 72, generator=g) * 0.4).to(DEV).view(-1, 3)).view(B, 24, 3, 3).contiguous()
    verts, _ = smpl.forward_arrays(betas, R)
    K = torch.tensor([[config.FOCAL_LENGTH, 0., 128.], [0., config.FOCAL_LENGTH, 128.], [0., 0., 1.]])
    rend = NMRRenderer(B, K, torch.eye(3), 256, rend_parts_seg=True, faces=smpl.faces, face_parts=smpl.face_parts).to(DEV)
    cam_t = torch.tensor([0., 0.2, 42.], device=DEV).expand(B, 3).contiguous()
    noise = torch.rand(B, 6890, 2, generator=g).to(DEV)
    # the load: four launches of a layer3-sized bf16x3 convolution forward per replay
    x = torch.randn(32, 32, 32, 256, generator=g).to(DEV)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).to(DEV)
    x3, xps = split3(L, x)
    w3, wps = weight_planes(L, w)
    y = torch.empty(32, 32, 32, 256, device=DEV)
    part = torch.empty(max(L.straps_conv_x3_stat_blocks(32, 32, 32, 256, 256, 3, 3, 1, 1, 0), 1) * 512, device=DEV)

    def conv():
        hipabi.check(L.straps_conv_fwd_x3(hipabi.ptr(x3), xps, hipabi.ptr(w3), wps, None, None, None, 0, hipabi.ptr(y), hipabi.ptr(part), 32, 32, 32, 256, 256, 3, 3, 1,
                                          1, 0, hipabi.stream_ptr()), 'straps_conv_fwd_x3')
    conv()
    torch.cuda.synchronize()
    load = torch.cuda.CUDAGraph()
    with torch.cuda.graph(load):
        for _ in range(4):
            conv()
    work = torch.cuda.Stream()
    ref, ref_depth = rend.render_arrays(verts, cam_t, want_depth=True, vert_noise_u=noise, noise_range=(-0.01, 0.01))
    ref, ref_depth = ref.clone(), ref_depth.clone()
    assert 0.1 < float((ref > 0).float().mean()) < 0.6
    bad = torch.zeros(2, device=DEV, dtype=torch.int64)
    torch.cuda.synchronize()
    for _ in range(300):
        load.replay()
        with torch.cuda.stream(work):
            seg, depth = rend.render_arrays(verts, cam_t, want_depth=True, vert_noise_u=noise, noise_range=(-0.01, 0.01))
            bad[0] += (seg != ref).sum()
            bad[1] += (depth != ref_depth).sum()
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0], 'part map / depth elements that differed from the first launch: %s' % bad.tolist()
