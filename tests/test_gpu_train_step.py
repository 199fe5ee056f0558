"""GPU tests of the whole training step (train_step.TrainStep): consistency with the autograd route
through the drop-in modules, Adam update vs torch.optim.Adam, and that training reduces the loss."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
import decisions
from straps_amd import hipabi
from straps_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
MP = straps_amd.synthetic_mean_params(0)
W = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
LOSSES = ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']


def _setup(B, seed=0, layers=18, conv_precision='fp32'):
    dev = torch.device('cuda:0')
    torch.manual_seed(seed)
    reg = straps_amd.SingleInputRegressor(18, layers, 3, mean_params=MP).to(dev).train()
    reg.image_encoder.conv_precision = conv_precision          # 'fp32': exact-fp32 MFMA chain; 'bf16x3': three-plane bf16 operands (csrc/conv_x3.hip)
    # (the bf16x3 parametrisations also run the step's SMPL forwards on the fully split matrix-pipe kernel: bench.py's default pairing)
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=B, precision='fp16x3_lbs' if conv_precision == 'bf16x3' else 'fp32').to(dev)
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


def _oracle_step(ts, reg, batch, layers, dtype, **kw):
    sd = {k: v.detach().cpu().clone() for k, v in reg.state_dict().items()}
    cpu_batch = {k: batch[k].cpu() for k in ('input', 'verts', 'joints2d', 'joints3d', 'shape', 'rot')}
    lv = {n: float(getattr(ts.crit, n + '_log_var')) for n in O.LOSS_TASKS}
    return O.train_step_loss_and_grads(cpu_batch, sd, O.ief_init_estimate(MP['pose'], MP['shape']), straps_amd.synthetic_smpl_model(0), layers, 3,
                                       lv, dtype=dtype, **kw)


def _ief_relu_flips(ts, taps64):
    """ReLU decisions of the IEF head on the GPU (from the stored activations) against the float64 oracle's pre-activations.
    Returns (gpu masks per iteration, list of flipped units (iteration, layer, row, unit, z64), observed fp32 evaluation error of the
    pre-activations = max |h_gpu - z64| over the units active on both sides)."""
    masks, flips, err = [], [], 0.0
    for it, (rec, (z1, z2)) in enumerate(zip(ts.last['ief_tape'], taps64)):
        pair = []
        for li, (h, z) in enumerate(((rec['h1'], z1), (rec['h2'], z2))):
            h = h.detach().cpu().double()
            m = h > 0
            both = m & (z > 0)
            if both.any():
                err = max(err, float((h - z)[both].abs().max()))
            for r, u in (m != (z > 0)).nonzero().tolist():
                flips.append((it, li, r, u, float(z[r, u])))
            pair.append(m)
        masks.append(tuple(pair))
    return masks, flips, err


def _whole_step_vs_float64(ts, reg, crit, layers, tag, loss_rel=1e-5, enc_bar=5e-5, with_fp32=True):
    """forward + loss + backward of `ts` on a fresh batch against autograd of the float64 oracle on the SAME batch: loss, the five loss
    weights and EVERY parameter tensor, relative L2 error per tensor < 1e-5 (IEF head) / enc_bar (encoder).

    These bars are statements about a differentiable function.  The network is piecewise linear in 24 million (resnet18, B=8) to 400
    million (resnet50, B=32) ReLU / max-pool decisions; where a pre-activation sits within the fp32 evaluation error of zero the GPU and
    float64 may take different sides, and the two gradients then differ by that unit's whole term -- which training-mode BatchNorm on
    a small batch spreads over every tensor upstream AND downstream of it (rounds 1-2 read the resulting 5e-3 .. 2e-2 as an
    ill-conditioned problem and compared against the float32 CPU oracle's own distance; measured here it is 2-4 decisions of 24 million
    for resnet18 and ~100 of 100 million for resnet50 and nothing else).  Such a unit is not waved through by a looser bar: every
    decision is compared unit by unit (tests/decisions.py), every differing one is shown to be a TIE (|z64|, or the float64 gap between
    the window's candidates, within 4x the activation error observed on the units of the same layer both sides agree on; their number
    bounded), and the float64 oracle is re-evaluated with the GPU's decisions forced -- the gradient of the function the GPU did
    evaluate.  The float32 CPU oracle's distance from the plain float64 run (the reference's own arithmetic, with its own ties) is
    printed beside it."""
    torch.set_num_threads(min(32, __import__('os').cpu_count() or 8))
    with torch.no_grad():
        batch = ts.make_batch()
    taps, rec64 = [], {'record': True}
    total, parts, grads, glv = _oracle_step(ts, reg, batch, layers, torch.float64, ief_taps=taps, enc_decisions=rec64)   # before the step touches the running statistics
    e32 = float('nan')
    if with_fp32:          # (informative only: the reference's own arithmetic against the same plain float64 run)
        _, _, g32, _ = _oracle_step(ts, reg, batch, layers, torch.float32)
        e32 = max(float((g32[n].double().reshape(-1) - grads[n].reshape(-1)).norm() / grads[n].norm().clamp_min(1e-30)) for n in grads)
        del g32
    ts.keep_enc_tape = True
    with torch.no_grad():
        loss = ts.forward_backward(batch)
    torch.cuda.synchronize()
    dec = decisions.decisions_from_tape(reg.image_encoder, ts.last['enc_tape'])
    ts.keep_enc_tape, ts.last['enc_tape'] = False, None
    err_cap = decisions.ERR_CAP if layers == 18 else decisions.ERR_CAP_R50
    n_relu, n_pool, tie_relu, tie_pool, act_err = decisions.compare_encoder_decisions(dec, rec64, err_cap)
    n_units = sum(m.numel() for m in dec['relu'])
    del rec64
    masks, flips, zerr = _ief_relu_flips(ts, taps)
    if flips:
        tol = 4 * zerr + 1e-7
        for it, li, r, u, z in flips:
            assert abs(z) <= tol, 'IEF unit (iteration %d, fc%d, row %d, unit %d) flipped its ReLU with |z64| = %.3e > %.3e: not a tie' % (it, li + 1, r, u, abs(z), tol)
    assert tie_relu <= 4.0 and tie_pool <= 4.0, 'a differing encoder decision is not a tie: |z64| %.2f x / gap %.2f x the activation error' % (tie_relu, tie_pool)
    assert n_relu <= 3 + 5e-6 * n_units and n_pool <= 3 + 2e-6 * ts.B * 64 * 64 * 64
    plain_worst = None
    if flips or n_relu or n_pool:
        plain_worst = max(float((ts.gviews[p].detach().cpu().double().reshape(-1) - grads[n].reshape(-1)).norm() / grads[n].norm().clamp_min(1e-30))
                          for n, p in reg.named_parameters())
        total, parts, grads, glv = _oracle_step(ts, reg, batch, layers, torch.float64, ief_masks=masks,
                                                enc_decisions={'relu': dec['relu'], 'pool': dec['pool']})
    del dec
    assert float(loss[0]) == pytest.approx(float(total), rel=loss_rel)
    for k, name in enumerate(O.LOSS_TASKS):                                           # kernel task order == oracle's LOSS_TASKS
        assert float(loss[1 + k]) == pytest.approx(float(parts[name]), rel=2 * loss_rel), name
    assert int(loss[11]) == int(O.check_joints2d_visibility(batch['joints2d'].cpu()).sum())
    table, bad = [], []
    for n, p in reg.named_parameters():
        r = grads[n].reshape(-1)
        e_gpu = float((ts.gviews[p].detach().cpu().double().reshape(-1) - r).norm() / r.norm().clamp_min(1e-30))
        bar = 1e-5 if n.startswith('ief_module.') else enc_bar
        table.append((n, e_gpu, bar))
        if not e_gpu < bar:
            bad.append('%-52s gpu %.2e  bar %.2e' % table[-1])
    enc = sorted(t[1] for t in table if t[0].startswith('image_encoder.'))
    print(tag, '%d tensors | %d of %d encoder ReLU decisions, %d of %d pooling windows, %d IEF ReLU decisions differ from float64 (ties: |z64| / gap <= %.2f x / %.2f x '
          'the layer\'s activation error, itself <= %.1e of the layer\'s largest float64 pre-activation: cap %.0e) | relative gradient error vs the float64 oracle on the GPU\'s decisions: IEF worst %.2e, encoder median %.2e worst %.2e'
          ' | vs the plain float64 run: worst %s (float32 CPU oracle: %.2e)'
          % (len(table), n_relu, n_units, n_pool, ts.B * 64 * 64 * 64, len(flips), tie_relu, tie_pool, act_err, err_cap, max(t[1] for t in table if t[0].startswith('ief_module.')),
             enc[len(enc) // 2], enc[-1], 'same run' if plain_worst is None else '%.2e' % plain_worst, e32))
    assert not bad, '\n'.join(bad)
    for name in O.LOSS_TASKS:
        g = float(ts.gviews[getattr(crit, name + '_log_var')])
        assert g == pytest.approx(float(glv[name]), rel=1e-4, abs=1e-7), name
    return table


@pytest.mark.parametrize('conv_precision', ['fp32', 'bf16x3'])
def test_whole_step_loss_and_all_71_gradients_vs_oracle_autograd(conv_precision):
    """(both convolution routes meet the same bars)
    T: forward + loss + backward of TrainStep on a B=8 batch against autograd of the float64 oracle on the SAME batch (running
    statistics restored first).  Loss to 1e-5.  Gradients, relative L2 error per tensor against the float64 oracle evaluated on the
    GPU's ReLU / max-pool decisions (see _whole_step_vs_float64: every differing decision identified and shown to be a tie):
    IEF head and loss weights < 1e-5, every encoder tensor < 5e-5 (measured: 1.3e-5 exact-fp32 route, 1.5e-5 bf16x3; against the
    PLAIN float64 run the same gradients sit at 1e-3 .. 8e-3 because of 2-4 tied decisions out of 24 million, and the float32 CPU
    oracle -- the reference's own arithmetic -- at 5e-3 .. 1e-2 for the same reason)."""
    B = 8
    dev, reg, smpl, crit = _setup(B, seed=5, conv_precision=conv_precision)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'])
    table = _whole_step_vs_float64(ts, reg, crit, 18, conv_precision + ' r18 B=8')
    assert len(table) == 66


def test_whole_step_at_the_bench_size_b64_vs_float64_oracle():
    """configs[2] itself -- resnet18, 64 bodies, the default bf16x3 route -- not a scaled-down stand-in: loss + all 71 gradients of one
    step against the float64 oracle on the GPU's decisions (two float64 passes over a 64-body batch: about a minute of CPU time on the
    GPU box's host).  Measured: 30 of 192 937 984 ReLU decisions and 3 of 16 777 216 pooling windows tied; IEF worst 1.6e-7, encoder worst
    3.4e-5 (bar 1e-4); 3.5e-3 against the plain float64 run."""
    B = 64
    dev, reg, smpl, crit = _setup(B, seed=11, conv_precision='bf16x3')
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=5)
    table = _whole_step_vs_float64(ts, reg, crit, 18, 'bf16x3 r18 B=64 (configs[2])', loss_rel=2e-5, enc_bar=1e-4, with_fp32=False)
    assert len(table) == 66


@pytest.mark.parametrize('conv_precision', ['fp32', 'bf16x3'])
def test_resnet50_whole_step_all_165_gradients_vs_float64_oracle(conv_precision):
    """the resnet18 test above for the Bottleneck encoder (models/resnet.py:80-121): loss + all 165 regressor gradients + the five loss
    weights against autograd of the FLOAT64 oracle on the GPU's decisions, both convolution routes; encoder bar 2e-4 (53 convolutions; measured
    7.1e-5 / 7.0e-5 worst, ~100 tied decisions of 100 million; against the plain float64 run 2e-2, the float32 CPU oracle 2.5e-2)."""
    B = 8
    dev, reg, smpl, crit = _setup(B, seed=9, layers=50, conv_precision=conv_precision)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=77)
    table = _whole_step_vs_float64(ts, reg, crit, 50, conv_precision + ' r50 B=8', enc_bar=2e-4)
    assert len(table) == 165


@pytest.mark.parametrize('conv_precision', ['fp32', 'bf16x3'])
def test_resnet50_step_configs3_shape(conv_precision):
    """configs[3] per-GPU shape: resnet50, 32 bodies.  hipGraph replay (single and split capture) == eager launches bit for
    bit, deterministic, Bottleneck flat-buffer layout consistent with the autograd route; then, on the parameters those four
    steps left, one more batch: loss + all 165 gradients against autograd of the float64 oracle on the GPU's decisions (encoder bar 2e-4)."""
    B = 32

    def run(use_graph, comm_overlap=False, steps=4):
        dev, reg, smpl, crit = _setup(B, seed=9, layers=50, conv_precision=conv_precision)
        ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=77, use_graph=use_graph, comm_overlap=comm_overlap)
        losses = torch.stack([ts.step().clone() for _ in range(steps)]).cpu()
        torch.cuda.synchronize()
        return losses, ts.flat_p.clone().cpu(), reg.image_encoder.layer3[5].bn3.running_var.clone().cpu(), ts, reg, crit

    l_e, p_e, rv_e, ts_e, reg_e, crit_e = run(False)
    assert torch.isfinite(l_e).all() and len(ts_e.params) == 165 + 5
    l_g, p_g, rv_g, ts_g, _, _ = run(True)
    assert ts_g.graph is not None, 'hipGraph capture fell back to eager launches'
    assert torch.equal(l_e, l_g) and torch.equal(p_e, p_g) and torch.equal(rv_e, rv_g)
    l_s, p_s, rv_s, ts_s, _, _ = run(True, comm_overlap=True)
    assert ts_s.graph is not None and ts_s.graph_tail is not None and ts_s.exchange.split_off > 0
    assert torch.equal(l_e, l_s) and torch.equal(p_e, p_s) and torch.equal(rv_e, rv_s)
    assert int(reg_e.image_encoder.layer4[2].bn3.num_batches_tracked) == 4
    # the two-bucket split sits at layer3's first parameter, as for resnet18
    names = [n for n, _ in reg_e.named_parameters()]
    off = sum(p.numel() for n, p in reg_e.named_parameters() if names.index(n) < names.index('image_encoder.layer3.0.conv1.weight'))
    assert ts_s.exchange.split_off == off
    del ts_g, ts_s
    # one more batch through the fused route: float64 oracle, every tensor (the eager run's data pipeline holds the next batch in its
    # other buffer set; make_batch() draws a fresh one, which is all this check needs)
    table = _whole_step_vs_float64(ts_e, reg_e, crit_e, 50, conv_precision + ' r50 B=32 after 4 steps', loss_rel=2e-5, enc_bar=2e-4)
    assert len(table) == 165


def test_resume_from_checkpoint_equals_uninterrupted_training(tmp_path):
    """save (reference .tar schema) -> fresh objects -> load_state_dict -> the next steps equal the uninterrupted run bit
    for bit: Adam moments, step count (host and device), BatchNorm buffers, loss weights and the generator's position."""
    B = 8

    def fresh(seed):
        dev, reg, smpl, crit = _setup(B, seed=seed)
        return reg, crit, TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=31)

    reg_a, crit_a, ts_a = fresh(2)
    for _ in range(3):
        ts_a.step()
    path = str(tmp_path / 'ck.tar')
    straps_amd.checkpoint_utils.save_checkpoint(path, 0, reg_a, ts_a, crit_a)
    draw_step = ts_a.data_state()               # generator step of the batch the NEXT step consumes (it is already in flight)
    assert draw_step == 3 and ts_a.draws.step() == 4
    tail_a = torch.stack([ts_a.step().clone() for _ in range(3)]).cpu()
    reg_b, crit_b, ts_b = fresh(3)                                   # different initial weights: everything must come from the file
    ck = straps_amd.checkpoint_utils.load_checkpoint(path)
    reg_b.load_state_dict(ck['model_state_dict'])
    crit_b.load_state_dict(ck['criterion_state_dict'])
    ts_b.load_state_dict(ck['optimiser_state_dict'])
    ts_b.set_data_state(draw_step)
    assert ts_b.steps == 3 and int(ts_b.step_t) == 3
    assert torch.equal(ts_b.exp_avg, ts_a_moments(ck, ts_b, 'exp_avg'))
    tail_b = torch.stack([ts_b.step().clone() for _ in range(3)]).cpu()
    assert torch.equal(tail_a, tail_b)
    assert torch.equal(ts_a.flat_p, ts_b.flat_p) and torch.equal(ts_a.exp_avg_sq, ts_b.exp_avg_sq)
    # torch.optim.Adam reads the same dict (schema compatibility both ways), and bad states are refused
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ts_b.params], lr=1e-4).load_state_dict(ts_b.state_dict())
    bad = ts_b.state_dict()
    bad['param_groups'][0]['betas'] = (0.8, 0.999)
    with pytest.raises(ValueError, match='defaults'):
        ts_b.load_state_dict(bad)
    bad = ts_b.state_dict()
    bad['param_groups'][0]['params'] = bad['param_groups'][0]['params'][:-1]
    with pytest.raises(ValueError, match='parameters'):
        ts_b.load_state_dict(bad)


def ts_a_moments(ck, ts, key):
    st = ck['optimiser_state_dict']['state']
    return torch.cat([st[i][key].reshape(-1) for i in range(len(ts.params))]).to(ts.dev)


def test_last_outputs_follow_the_replayed_graph_parity():
    """with the data pipeline two graphs are captured (one per buffer parity): ts.last must be the outputs of the graph
    that just ran, i.e. consistent with the loss it returned."""
    B = 4
    dev, reg, smpl, crit = _setup(B, seed=6)
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=True)
    for i in range(7):
        loss = ts.step()
        torch.cuda.synchronize()
        assert ts.last['loss'].data_ptr() == loss.data_ptr()
        cur = ts._bufs[1 - ts._cur]                                   # the batch this step trained on (the parity flipped afterwards)
        mse = float(((ts.last['verts'] - cur['verts']) ** 2).mean())
        assert mse == pytest.approx(float(loss[6]), rel=1e-4), i      # loss[6] = raw vertex MSE of THIS step
    assert ts.graph is not None and len(ts._last_by_parity) == 2
    assert ts._last_by_parity[0]['verts'].data_ptr() != ts._last_by_parity[1]['verts'].data_ptr()


def test_set_data_state_with_captured_graphs_recaptures():
    """re-seating the data stream after the hipGraphs were captured drops them (they would consume the batch that was in flight)
    and the run continues exactly like an uninterrupted one."""
    B = 4

    def run(reseat):
        dev, reg, smpl, crit = _setup(B, seed=8)
        ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], seed=5, use_graph=True)
        out = []
        for i in range(8):
            if reseat and i == 4:
                assert ts.graph is not None
                ts.set_data_state(ts.data_state())
                assert ts.graph is None
            out.append(ts.step().clone())
        torch.cuda.synchronize()
        assert ts.graph is not None
        return torch.stack(out).cpu(), ts.flat_p.clone().cpu()

    la, pa = run(False)
    lb, pb = run(True)
    assert torch.equal(la, lb) and torch.equal(pa, pb)


def _nine_losses(use_graph, pipeline, alias_capture_stream=False, B=4):
    dev, reg, smpl, crit = _setup(B, seed=6)
    keep = []
    if alias_capture_stream:
        # torch.cuda.Stream() hands out 32 pooled streams round-robin and torch.cuda.graph captures on ONE class-level stream from the same
        # pool: arrange that the TrainStep's data stream IS that capture stream (in a long process -- the full GPU suite -- this happens by itself)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            torch.zeros(4, device=dev) + 1
        cap = torch.cuda.graph.default_capture_stream.cuda_stream
        for _ in range(64):
            keep.append(torch.cuda.Stream(device=dev))
            if keep[-1].cuda_stream == cap:
                break
        assert keep[-1].cuda_stream == cap
        keep += [torch.cuda.Stream(device=dev) for _ in range(31)]
    ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=MP['shape'], use_graph=use_graph, pipeline_data=pipeline)
    if alias_capture_stream:
        assert ts.data_stream.cuda_stream == cap
    out = torch.stack([ts.step().clone() for _ in range(9)]).cpu()
    torch.cuda.synchronize()
    assert ts.graph is not None or not use_graph
    return out, ts.flat_p.clone().cpu()


def test_graph_replay_on_one_stream_and_with_an_aliased_data_stream_equals_eager_over_nine_steps():
    """Round 3 regression.  With the batch generation and the step captured on ONE stream (no data pipeline -- or a data stream that happens to
    be torch's graph-capture stream, which the 32-entry stream pool makes inevitable in a long process) the two share pool memory, and the
    runtime's memset nodes (hipMemsetAsync of the z-buffer and of the flat gradient buffer) clobbered it: wrong losses from the fourth replay on,
    or a memory fault.  The clears are fill kernels now (csrc/augment.hip straps_fill_bytes).  Nine steps: eager == every graph form, bit for bit."""
    ref, p_ref = _nine_losses(False, False)
    for kw in (dict(use_graph=True, pipeline=False), dict(use_graph=True, pipeline=True), dict(use_graph=True, pipeline=True, alias_capture_stream=True),
               dict(use_graph=False, pipeline=True)):
        got, p = _nine_losses(**kw)
        assert torch.equal(got, ref), kw
        assert torch.equal(p, p_ref), kw
