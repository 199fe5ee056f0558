"""GPU: BASELINE.json configs[0] -- single-image predict plumbing at batch 1 (predict/predict_3D.py:116-149) on the committed
STAND-IN proxy (tests/golden/predict_golden.npz: a rendered silhouette of the synthetic SMPL-shaped model + 17 joints; the
reference ships no precomputed proxy, detectron2 builds it at run time).  What is pinned by the reference: the proxy layout
and heat-maps (its numpy routine, utils/label_conversions.py:58-87) and the regressor outputs on that proxy."""
import json
import os

import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_state_dict

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
MP = straps_amd.synthetic_mean_params(0)


def _standin():
    g = np.load(os.path.join(GOLD, 'predict_golden.npz'))
    sil = np.unpackbits(g['sil_bits'])[:256 * 256].reshape(256, 256)
    heat = np.zeros(256 * 256 * 17, np.float32)
    heat[g['heat_idx']] = g['heat_val']
    ref_proxy = np.concatenate([sil.astype(np.float32)[None], np.transpose(heat.reshape(256, 256, 17), (2, 0, 1))], axis=0)   # predict_3D.py:71-74
    return g, sil, ref_proxy


def test_create_proxy_representation_matches_reference_numpy_routine():
    g, sil, ref_proxy = _standin()
    x = straps_amd.checkpoint_utils.create_proxy_representation(sil, g['joints2D'], 256, DEV)       # [17,3] joints incl. confidence column
    assert tuple(x.shape) == (1, 18, 256, 256) and x.dtype == torch.float32 and x.is_cuda
    got = x[0].cpu().numpy()
    assert np.array_equal(got != 0, ref_proxy != 0)                 # same pixels written: truncation, clipping at 255, skipped joint
    np.testing.assert_allclose(got, ref_proxy, rtol=0, atol=2e-7)
    assert np.array_equal(got[0], sil.astype(np.float32))
    # float joints are truncated like .astype(np.int16) (:71), not rounded
    j = g['joints2D'].copy()
    j[:, :2] = np.trunc(j[:, :2]) + np.where(j[:, :2] >= 0, 0.9, -0.9)
    assert torch.equal(straps_amd.checkpoint_utils.create_proxy_representation(sil, j, 256, DEV), x)


@pytest.mark.parametrize('layers', [18, 50])
def test_single_image_predict_b1(layers):
    g, sil, ref_proxy = _standin()
    man = json.load(open(os.path.join(GOLD, 'state_dict_keys_r%d.json' % layers)))['keys']
    reg = straps_amd.SingleInputRegressor(resnet_in_channels=18, resnet_layers=layers, ief_iters=3, mean_params=MP)       # run_predict.py:9-11
    reg.load_state_dict({k: torch.from_numpy(v) for k, v in det_state_dict(man).items()}, strict=True)                  # :15-16
    reg.to(DEV).eval()
    model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(model, batch_size=1).to(DEV)                                                                   # predict_3D.py:91
    proxy = straps_amd.checkpoint_utils.create_proxy_representation(sil, g['joints2D'], 256, DEV)
    with torch.no_grad():
        cam, pose, shape = reg(proxy)
        assert tuple(cam.shape) == (1, 3) and tuple(pose.shape) == (1, 144) and tuple(shape.shape) == (1, 10)
        R = straps_amd.rot6d_to_rotmat(pose.contiguous()).view(-1, 24, 3, 3)
        out = smpl(body_pose=R[:, 1:], global_orient=R[:, 0].unsqueeze(1), betas=shape, pose2rot=False)
        v2d = straps_amd.cam_utils.undo_keypoint_normalisation(straps_amd.cam_utils.orthographic_project_torch(out.vertices, cam), 512)
        reposed = smpl(betas=shape)
    est = torch.cat([cam, pose, shape], 1).cpu().numpy()
    np.testing.assert_allclose(est, g['out_r%d' % layers], rtol=2e-4, atol=2e-4)                 # the imported reference's output
    assert tuple(out.vertices.shape) == (1, 6890, 3) and tuple(out.joints.shape) == (1, 90, 3)
    # SMPL at B = 1 on identical (theta, beta): north_star bar 1e-4, measured ~1e-6
    ov, oj = O.smpl_forward(model, shape.cpu(), rotmats=R.cpu())
    assert float((out.vertices.cpu() - ov).abs().max()) < 1e-5 and float((out.joints.cpu() - oj).abs().max()) < 1e-5
    orv, _ = O.smpl_forward(model, shape.cpu(), rotmats=torch.eye(3).expand(1, 24, 3, 3))
    assert float((reposed.vertices.cpu() - orv).abs().max()) < 1e-5
    want2d = (O.orthographic_project(out.vertices.cpu(), cam.cpu()) + 1) * 256.0
    assert float((v2d.cpu() - want2d).abs().max()) < 1e-3 and tuple(v2d.shape) == (1, 6890, 2)


def test_real_layout_smpl_files_load_and_run(tmp_path):
    """S0: a model written in the layout of the files the reference loads (additional/smpl/SMPL_NEUTRAL.pkl as chumpy-free
    pickle with scipy-sparse J_regressor, [6890,3,207] posedirs, kintree_table, uint32 faces; + the three additional/*.npy
    regressors, config.py:3-8) gives the same meshes as the in-memory dict."""
    from test_host_logic import write_real_layout_model
    model = straps_amd.synthetic_smpl_model(0)
    d, extra = write_real_layout_model(tmp_path, model)
    a = straps_amd.SMPL(d, batch_size=3, extra_regressor_paths=extra).to(DEV)
    b = straps_amd.SMPL(model, batch_size=3).to(DEV)
    betas = torch.randn(3, 10, generator=torch.Generator().manual_seed(1)).to(DEV)
    aa = (torch.randn(3, 72, generator=torch.Generator().manual_seed(2)) * 0.3).to(DEV)
    with torch.no_grad():
        oa = a(body_pose=aa[:, 3:], global_orient=aa[:, :3], betas=betas)                      # pose2rot=True path (val loop :258-260)
        ob = b(body_pose=aa[:, 3:], global_orient=aa[:, :3], betas=betas)
    assert torch.equal(oa.vertices, ob.vertices) and torch.equal(oa.joints, ob.joints)
    assert a.faces.shape == (13776, 3) and np.array_equal(a.faces, model['faces'])
