"""GPU tests of the whole training step (train_step.TrainStep): consistency with the autograd route
through the drop-in modules, Adam update vs torch.optim.Adam, and that training reduces the loss."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from straps_amd import hipabi
from straps_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
MP = straps_amd.synthetic_mean_params(0)
W = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
LOSSES = ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']


def _setup(B, seed=0, layers=18):
    dev = torch.device('cuda:0')
    torch.manual_seed(seed)
    reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B).to(dev)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(LOSSES, init_loss_weights=W, reduction='mean').to(dev)
    return dev, reg, smpl, crit


def test_step_gradients_match_autograd_route_and_adam_matches_torch():
    B = 6
    dev, reg, smpl, crit = _setup(B)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'])
    sd0 = {k: v.clone() for k, v in reg.state_dict().items()}
    with torch.no_grad():
        batch = ts.make_batch()
    assert batch['input'].shape == (B, 18, 256, 256)
    assert 0.02 < float(batch['input'][:, 0].mean()) < 0.6            # a silhouette-sized foreground
    with torch.no_grad():
        loss = ts.forward_backward(batch)
    g_fused = {p: ts.gviews[p].clone() for p in ts.params}
    # ---- the same step through the drop-in modules + autograd (fresh BN running stats restored first) ----
    reg.load_state_dict(sd0)
    cam, pose, shape = reg(batch['input'])
    R = straps_amd.rot6d_to_rotmat(pose).view(-1, 24, 3, 3)
    out = smpl(body_pose=R[:, 1:], global_orient=R[:, 0:1], betas=shape, pose2rot=False)
    j_coco = out.joints[:, straps_amd.config.ALL_JOINTS_TO_COCO_MAP]
    j_h36m = out.joints[:, straps_amd.config.ALL_JOINTS_TO_H36M_MAP][:, straps_amd.config.H36M_TO_J14]
    pred = {'joints2D': straps_amd.cam_utils.orthographic_project_torch(j_coco, cam), 'verts': out.vertices, 'shape_params': shape,
            'pose_params_rot_matrices': R, 'joints3D': j_h36m}
    lab = {'joints2D': batch['joints2d'], 'verts': batch['verts'], 'shape_params': batch['shape'], 'pose_params_rot_matrices': batch['rot'],
           'joints3D': batch['joints3d'], 'vis': straps_amd.cam_utils.check_joints2d_visibility_torch(batch['joints2d'], 256)}
    total, parts = crit(lab, pred)
    total.backward()
    assert float(total) == pytest.approx(float(loss[0]), rel=1e-4)
    for n, p in list(reg.named_parameters()) + list(crit.named_parameters()):
        a, b = p.grad.double(), g_fused[p].double()
        denom = float(a.norm().clamp_min(1e-12))
        assert float((a - b).norm()) / denom < 2e-3, n
    # ---- Adam: two fused steps on fixed gradients == torch.optim.Adam ----
    ref = [p.detach().clone().requires_grad_() for p in ts.params]
    opt = torch.optim.Adam(ref, lr=1e-4)
    for _ in range(2):
        for r, p in zip(ref, ts.params):
            r.grad = g_fused[p].clone()
        opt.step()
        ts.flat_g.copy_(torch.cat([g_fused[p].reshape(-1) for p in ts.params]))
        ts.optimise()
    for r, p in zip(ref, ts.params):
        assert float((r.detach() - p.detach()).abs().max()) <= 1e-6 + 1e-5 * float(r.abs().max())
    sd = ts.state_dict()
    assert set(sd.keys()) == {'state', 'param_groups'} and len(sd['state']) == len(ts.params) == 71
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ts.params], lr=1e-4).load_state_dict(sd)    # schema-compatible


def test_training_reduces_loss_and_updates_running_stats():
    B = 16
    dev, reg, smpl, crit = _setup(B, seed=1)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-3, mean_shape=MP['shape'])
    rm0 = reg.image_encoder.bn1.running_mean.clone()
    losses = [float(ts.step()[0]) for _ in range(30)]
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < np.mean(losses[:5]), losses
    assert int(reg.image_encoder.bn1.num_batches_tracked) == 30
    assert not torch.equal(rm0, reg.image_encoder.bn1.running_mean)
    # eval-mode inference with the trained weights still works (packed-weight caches were refreshed)
    reg.eval()
    with torch.no_grad():
        cam, pose, shape = reg(ts.make_batch()['input'])
    assert torch.isfinite(cam).all() and torch.isfinite(pose).all()


def test_tracked_metrics_match_oracle_on_the_step_outputs():
    """f3 wired into the step: the on-device metric sums equal the oracle's per-sample numpy/SVD evaluation of the same
    predictions (utils/eval_utils.py semantics), also when the step is replayed from a hipGraph."""
    B = 8
    dev, reg, smpl, crit = _setup(B, seed=3)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], track_metrics=True, use_graph=False)
    with torch.no_grad():
        batch = ts.make_batch()
        ts.forward_backward(batch)
    ts.steps = 1
    got = ts.metrics_summary()
    verts, joints, reposed = ts.last['verts'].cpu(), ts.last['joints'].cpu(), ts.last['reposed'].cpu()
    j14 = [straps_amd.config.ALL_JOINTS_TO_H36M_MAP[k] for k in straps_amd.config.H36M_TO_J14]
    pv = O.point_metrics(verts.numpy(), batch['verts'].cpu().numpy()).sum(0) / (B * 6890)
    pj = O.point_metrics(joints[:, j14].numpy(), batch['joints3d'].cpu().numpy()).sum(0) / (B * 14)
    pt = O.point_metrics(reposed.numpy(), batch['reposed'].cpu().numpy()).sum(0) / (B * 6890)
    want = {'pves': pv[0], 'pves_sc': pv[1], 'pves_pa': pv[2], 'pve-ts': pt[0], 'pve-ts_sc': pt[1], 'mpjpes': pj[0], 'mpjpes_sc': pj[1],
            'mpjpes_pa': pj[2]}
    for k, v in want.items():
        assert got[k] == pytest.approx(float(v), rel=2e-4), k
    assert got['shape_mses'] > 0 and got['pose_mses'] > 0 and got['joints2D_l2es'] > 0
    # graph replay accumulates too
    dev, reg, smpl, crit = _setup(B, seed=3)
    tg = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], track_metrics=True, use_graph=True)
    for _ in range(5):
        tg.step()
    s5 = tg.metrics_summary()
    assert tg.graph is not None and all(np.isfinite(list(s5.values()))) and s5['pves'] > 0
