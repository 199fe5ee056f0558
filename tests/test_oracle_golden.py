"""CPU: the oracle (oracle/straps_oracle.py) against golden vectors captured from the imported
reference (oracle/make_golden.py).  This is what PINS the oracle for every stage that imports."""
import json
import os

import numpy as np
import pytest
import torch

import straps_oracle as O
from detgen import det_uniform, det_state_dict

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _sd(layers):
    man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r%d.json' % layers)))['keys']
    return {k: torch.from_numpy(v) for k, v in det_state_dict(man).items()}


def _init():
    import straps_amd
    mp = straps_amd.synthetic_mean_params(0)
    return O.ief_init_estimate(mp['pose'], mp['shape'])


@pytest.fixture(scope='module')
def enc_gold():
    return np.load(os.path.join(GOLD, 'encoder_golden.npz'))


@pytest.fixture(scope='module')
def small():
    return np.load(os.path.join(GOLD, 'small_golden.npz'))


@pytest.mark.parametrize('layers', [18, 50])
def test_regressor_eval_matches_reference(enc_gold, layers):
    torch.set_num_threads(8)
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 4242, 0.0, 1.0))
    sd = _sd(layers)
    taps = {}
    with torch.no_grad():
        feat = O.resnet_forward(x, sd, layers, False, taps=taps)
        cam, pose, shape, est = O.ief_forward(feat, sd, _init(), 3)
    tag = 'r%d_eval_' % layers
    np.testing.assert_allclose(feat.numpy(), enc_gold[tag + 'feat_full'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(est.numpy(), enc_gold[tag + 'out'], rtol=1e-5, atol=2e-6)
    for k in ('stem', 'pool', 'layer1', 'layer2', 'layer3', 'layer4'):
        flat = taps[k].reshape(-1).numpy()
        np.testing.assert_allclose(flat[enc_gold[tag + k + '_idx']], enc_gold[tag + k + '_val'],
                                   rtol=1e-5, atol=1e-6, err_msg=k)
        assert abs(float(flat.astype(np.float64).mean()) - enc_gold[tag + k + '_stats'][0]) < 1e-6


@pytest.mark.parametrize('layers', [18, 50])
def test_regressor_train_mode_matches_reference(enc_gold, layers):
    torch.set_num_threads(8)
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 4242, 0.0, 1.0))
    sd = _sd(layers)
    with torch.no_grad():
        _, _, _, est = O.regressor_forward(x, sd, _init(), layers, 3, training=True)
    tag = 'r%d_train_' % layers
    np.testing.assert_allclose(est.numpy(), enc_gold[tag + 'out'], rtol=2e-5, atol=5e-6)
    for bn in ('image_encoder.bn1', 'image_encoder.layer2.0.downsample.1', 'image_encoder.layer4.1.bn2'):
        np.testing.assert_allclose(sd[bn + '.running_mean'].numpy(), enc_gold[tag + bn + '.running_mean'],
                                   rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sd[bn + '.running_var'].numpy(), enc_gold[tag + bn + '.running_var'],
                                   rtol=1e-5, atol=1e-6)
        assert int(sd[bn + '.num_batches_tracked']) == int(enc_gold[tag + bn + '.nbt']) == 1


def test_train_mode_grads_match_reference():
    """autograd through the oracle reproduces the reference's parameter gradients (r18)."""
    torch.set_num_threads(8)
    gold = json.load(open(os.path.join(GOLD, 'grad_checks_r18.json')))
    x = torch.from_numpy(det_uniform((2, 18, 256, 256), 4242, 0.0, 1.0))
    sd = _sd(18)
    names = [k for k in gold]
    for k in names:
        sd[k].requires_grad_(True)
    _, _, _, est = O.regressor_forward(x, sd, _init(), 18, 3, training=True)
    coef = torch.from_numpy(det_uniform((2, 157), 555))
    (est * coef).sum().backward()
    for k in names:
        g = sd[k].grad
        # ief_layers.* aliases share storage with fc* in the reference: grads only on fc*
        ref_norm, ref_head = gold[k]
        assert abs(float(g.double().norm()) - ref_norm) <= 2e-4 * max(1.0, ref_norm), k
        np.testing.assert_allclose(g.reshape(-1)[:3].numpy(), np.array(ref_head, dtype=np.float32),
                                   rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.parametrize('layers,F_', [(18, 512), (50, 2048)])
def test_ief(small, layers, F_):
    sd = _sd(layers)
    feat = torch.from_numpy(det_uniform((4, F_), 31, 0.0, 2.0))
    with torch.no_grad():
        _, _, _, est = O.ief_forward(feat, sd, _init(), 3)
    np.testing.assert_allclose(est.numpy(), small['ief_r%d_out' % layers], rtol=1e-5, atol=2e-6)


def test_rot6d(small):
    x6 = torch.from_numpy(det_uniform((4, 144), 32, -1.5, 1.5))
    R = O.rot6d_to_rotmat(x6)
    np.testing.assert_allclose(R.numpy(), small['rot6d_out'], rtol=1e-6, atol=1e-7)
    eye = torch.eye(3).expand(96, 3, 3)
    assert float((R.transpose(1, 2) @ R - eye).abs().max()) < 1e-5


def test_projections_and_visibility(small):
    pts = torch.from_numpy(det_uniform((3, 17, 3), 33, -1.0, 1.0))
    cam = torch.from_numpy(det_uniform((3, 3), 34, 0.5, 1.2))
    np.testing.assert_allclose(O.orthographic_project(pts, cam).numpy(), small['ortho_out'], rtol=1e-6, atol=1e-7)
    K = torch.from_numpy(O.intrinsics_matrix().astype(np.float32))[None].expand(3, -1, -1)
    R = torch.eye(3)[None].expand(3, -1, -1)
    tr = torch.tensor([[0., 0.2, 42.0]]).expand(3, -1) + torch.from_numpy(det_uniform((3, 3), 35, -0.1, 0.1))
    np.testing.assert_allclose(O.perspective_project(pts, R, tr, K).numpy(), small['persp_out'], rtol=1e-5, atol=1e-4)
    vis = O.check_joints2d_visibility(torch.from_numpy(small['vis_in']))
    assert np.array_equal(vis.numpy(), small['vis_out'])
    assert bool(vis[0, 0]) and not bool(vis[0, 1]) and not bool(vis[0, 2])


def test_heatmaps_and_binary(small):
    hm = O.joints2d_to_heatmaps(torch.from_numpy(small['heat_in']))
    flat = hm.reshape(-1)
    nz = flat.nonzero().squeeze(1).numpy()
    assert np.array_equal(nz, small['heat_nz_idx'])
    np.testing.assert_allclose(flat[nz].numpy(), small['heat_nz_val'], rtol=1e-6, atol=1e-7)
    assert abs(float(hm.max()) - 0.9824) < 1e-3            # no exact-centre sample
    assert float(hm[:, :, 255, :].abs().max()) == 0.0 and float(hm[:, :, :, 255].abs().max()) == 0.0
    seg = (torch.from_numpy(det_uniform((2, 256, 256), 38, 0.0, 1.0)) > 0.7).float() * \
        torch.from_numpy(np.floor(det_uniform((2, 256, 256), 39, 1.0, 6.999)))
    assert float(O.multiclass_to_binary(seg).sum()) == float(small['binary_sum'])


def test_multi_task_loss_and_grads(small):
    B = 4
    lab = {'verts': torch.from_numpy(det_uniform((B, 6890, 3), 40)),
           'joints2D': torch.from_numpy(det_uniform((B, 17, 2), 41, -40.0, 300.0)),
           'joints3D': torch.from_numpy(det_uniform((B, 14, 3), 42)),
           'shape_params': torch.from_numpy(det_uniform((B, 10), 43, -2, 2)),
           'pose_params_rot_matrices': torch.from_numpy(det_uniform((B, 24, 3, 3), 44))}
    lab['vis'] = O.check_joints2d_visibility(lab['joints2D'])
    assert int(lab['vis'].sum()) == int(small['loss_nvis'])
    outp = {'verts': torch.from_numpy(det_uniform((B, 6890, 3), 45)).requires_grad_(),
            'joints2D': torch.from_numpy(det_uniform((B, 17, 2), 46)).requires_grad_(),
            'joints3D': torch.from_numpy(det_uniform((B, 14, 3), 47)).requires_grad_(),
            'shape_params': torch.from_numpy(det_uniform((B, 10), 48, -2, 2)).requires_grad_(),
            'pose_params_rot_matrices': torch.from_numpy(det_uniform((B, 24, 3, 3), 49)).requires_grad_()}
    w = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
    lv = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in O.init_log_vars(w).items()}
    total, parts = O.multi_task_loss(lab, outp, lv)
    total.backward()
    assert abs(float(total) - float(small['loss_total'])) < 1e-5 * abs(float(small['loss_total']))
    order = ('verts', 'joints2D', 'joints3D', 'shape_params', 'pose_params')
    np.testing.assert_allclose([float(parts[k]) for k in order], small['loss_parts'], rtol=1e-5)
    np.testing.assert_allclose([float(lv[k].grad) for k in order], small['loss_grad_logvars'], rtol=1e-5)
    np.testing.assert_allclose(outp['joints2D'].grad.numpy(), small['loss_grad_j2d'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(outp['shape_params'].grad.numpy(), small['loss_grad_shape'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(outp['verts'].grad.reshape(-1)[:64].numpy(), small['loss_grad_verts_head'], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(outp['joints3D'].grad.numpy(), small['loss_grad_j3d'], rtol=1e-5, atol=1e-8)


def test_adam_two_steps(small):
    man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r18.json')))
    sd = _sd(18)
    ps = [sd[n].clone() for n in man['param_order']] + [torch.zeros(()) for _ in range(5)]
    grads = [torch.from_numpy(det_uniform(tuple(p.shape), 9000 + i, -1e-2, 1e-2)).reshape(p.shape) for i, p in enumerate(ps)]
    before = [p.clone() for p in ps]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    O.adam_step(ps, grads, m, v, 1)
    O.adam_step(ps, grads, m, v, 2)
    ds = np.array([float((p - b).double().sum()) for p, b in zip(ps, before)])
    da = np.array([float((p - b).double().abs().sum()) for p, b in zip(ps, before)])
    np.testing.assert_allclose(da, small['adam_delta_abs'], rtol=1e-4)
    np.testing.assert_allclose(ds, small['adam_delta_sum'], rtol=1e-3, atol=1e-6)


def test_evaluation_metrics(small):
    """PVE / PVE-SC / PVE-PA and MPJPE sums: oracle restatement vs the reference's utils/eval_utils.py."""
    from detgen import det_metrics_case
    for tag, npts, seed in (('verts', 6890, 70), ('j14', 14, 72)):
        pv, tv = det_metrics_case(npts, seed)
        np.testing.assert_allclose(O.point_metrics(pv, tv), small['metrics_%s_sums' % tag], rtol=2e-5)


def test_crop_boxes_match_reference(small):
    """pure-numpy half of utils/image_utils.py (batch_crop_seg_to_bounding_box) with and without jitter; the jitter
    draws are reproduced from numpy's seeded stream in the reference's call order (rand(), rand(2) per sample)."""
    from detgen import det_crop_case
    seg, j = det_crop_case()
    np.random.seed(11)
    u = np.zeros((seg.shape[0], 3))
    for i in range(seg.shape[0]):
        u[i, 0] = np.random.rand()
        u[i, 1:3] = np.random.rand(2)
    boxes, cj = O.crop_boxes(seg, j, u)
    assert np.array_equal(np.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1), small['crop_shapes'])
    np.testing.assert_allclose(cj, small['crop_joints'], rtol=0, atol=1e-9)
    np.testing.assert_allclose([seg[i, b[0]:b[2], b[1]:b[3]].sum() for i, b in enumerate(boxes)], small['crop_sums'])
    boxes0, cj0 = O.crop_boxes(seg, j, None)
    assert np.array_equal(np.stack([boxes0[:, 2] - boxes0[:, 0], boxes0[:, 3] - boxes0[:, 1]], 1), small['crop0_shapes'])
    np.testing.assert_allclose(cj0, small['crop0_joints'], rtol=0, atol=1e-9)
    out, oj, _ = O.crop_resize(seg, j, u)
    assert out.shape == (seg.shape[0], 256, 256) and set(np.unique(out)) <= set(range(7))
