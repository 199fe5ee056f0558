"""CPU: SMPL restatement in the oracle is PARITY-UNPINNED vs smplx (absent); these analytic
invariants + fp64/fp32 agreement are what pins it."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_uniform


@pytest.fixture(scope='module')
def model():
    return straps_amd.synthetic_smpl_model(0)


def _rand_pose(B, seed, scale=0.4):
    aa = torch.from_numpy(det_uniform((B, 72), seed, -scale, scale))
    return aa, O.batch_rodrigues(aa.reshape(-1, 3)).view(B, 24, 3, 3)


def test_model_shapes_and_sparsity(model):
    assert model['v_template'].shape == (6890, 3) and model['posedirs'].shape == (207, 20670)
    assert model['shapedirs'].shape == (6890, 3, 10) and model['weights'].shape == (6890, 24)
    assert (np.count_nonzero(model['weights'], axis=1) <= 4).all()
    np.testing.assert_allclose(model['weights'].sum(1), 1.0, atol=1e-6)
    for k, r in (('J_regressor', 24), ('J_regressor_extra', 9), ('J_regressor_cocoplus', 19), ('J_regressor_h36m', 17)):
        assert model[k].shape == (r, 6890)
        np.testing.assert_allclose(model[k].sum(1), 1.0, atol=1e-5)
    m2 = straps_amd.synthetic_smpl_model(0)
    assert all(np.array_equal(model[k], m2[k]) for k in model)


def test_fp32_vs_fp64(model):
    B = 3
    betas = torch.from_numpy(det_uniform((B, 10), 1, -2, 2))
    _, R = _rand_pose(B, 2)
    v32, j32 = O.smpl_forward(model, betas, rotmats=R, dtype=torch.float32)
    v64, j64 = O.smpl_forward(model, betas, rotmats=R.double(), dtype=torch.float64)
    assert float((v32.double() - v64).abs().max()) < 2e-6
    assert float((j32.double() - j64).abs().max()) < 2e-6
    assert v32.shape == (B, 6890, 3) and j32.shape == (B, 90, 3)


def test_zero_pose_is_shaped_template(model):
    B = 2
    betas = torch.from_numpy(det_uniform((B, 10), 3, -2, 2)).double()
    R = torch.eye(3, dtype=torch.float64).expand(B, 24, 3, 3)
    v, j = O.smpl_forward(model, betas, rotmats=R, dtype=torch.float64)
    v_shaped = torch.from_numpy(model['v_template']).double()[None] + torch.einsum(
        'bl,mkl->bmk', betas, torch.from_numpy(model['shapedirs']).double())
    # skinning weights are stored fp32: rows sum to 1 only to ~1e-7, so identity skinning is exact to |v|*1e-7
    assert float((v - v_shaped).abs().max()) < 2e-7
    J = torch.einsum('bik,ji->bjk', v_shaped, torch.from_numpy(model['J_regressor']).double())
    assert float((j[:, :24] - J).abs().max()) < 1e-12
    assert float((j[:, 24:45] - v[:, model['extra_vertex_ids'].tolist()]).abs().max()) == 0.0


def test_global_rotation_is_rigid_about_root(model):
    B = 2
    betas = torch.from_numpy(det_uniform((B, 10), 4, -1, 1)).double()
    _, R = _rand_pose(B, 5)
    R = R.double()
    v0, j0 = O.smpl_forward(model, betas, rotmats=R, dtype=torch.float64)
    Rg = O.batch_rodrigues(torch.tensor([[0.3, -1.1, 0.5]], dtype=torch.float64))[0]
    R2 = R.clone()
    R2[:, 0] = Rg @ R[:, 0]
    v1, j1 = O.smpl_forward(model, betas, rotmats=R2, dtype=torch.float64)
    root = j0[:, 0:1]
    assert float((j1[:, 0:1] - root).abs().max()) < 1e-12       # root joint does not move
    want = (v0 - root) @ Rg.T + root
    # pose-corrective offsets depend only on R[1:], so the rotation is exactly rigid
    assert float((v1 - want).abs().max()) < 5e-7
    assert float((j1 - ((j0 - root) @ Rg.T + root)).abs().max()) < 5e-9       # J_regressor rows sum to 1 only to fp32 rounding


def test_pose2rot_equals_rotmat_path(model):
    B = 2
    betas = torch.from_numpy(det_uniform((B, 10), 6, -1, 1))
    aa, R = _rand_pose(B, 7)
    v0, j0 = O.smpl_forward(model, betas, rotmats=R)
    v1, j1 = O.smpl_forward(model, betas, full_pose_aa=aa)
    assert torch.equal(v0, v1) and torch.equal(j0, j1)


def test_rodrigues_is_rotation():
    aa = torch.from_numpy(det_uniform((50, 3), 8, -3, 3)).double()
    aa[0] = 0
    R = O.batch_rodrigues(aa)
    # the +1e-8 inside the norm (smplx convention) makes the axis unit only to ~1e-8
    assert float((R.transpose(1, 2) @ R - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-6
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-6
    assert float((R[0] - torch.eye(3, dtype=torch.float64)).abs().max()) < 1e-7
