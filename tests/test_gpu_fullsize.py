"""GPU: size-independent properties at BASELINE.json's full sizes (the oracle cannot run these in seconds):
determinism, hipGraph replay == eager launches, batch-slice independence, full-batch rasteriser == per-body renders."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_uniform
from straps_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MP = straps_amd.synthetic_mean_params(0)
W = {'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}
LOSSES = ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D']


def _train(use_graph, steps, overlap=True, comm_overlap=False, pipeline=True):
    torch.manual_seed(7)
    reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=MP).to(DEV).train()
    smpl = straps_amd.SMPL(straps_amd.synthetic_smpl_model(0), batch_size=64).to(DEV)
    crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(LOSSES, init_loss_weights=W, reduction='mean').to(DEV)
    ts = TrainStep(reg, smpl, crit, 64, lr=1e-4, seed=99, mean_shape=MP['shape'], use_graph=use_graph, overlap_wgrad=overlap,
                   comm_overlap=comm_overlap, pipeline_data=pipeline)
    losses = [ts.step().clone() for _ in range(steps)]
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), ts.flat_p.clone().cpu(), reg.image_encoder.bn1.running_var.clone().cpu(), ts


def test_train_step_b64_deterministic_and_graph_equals_eager():
    """configs[2] at full size: same seeds -> bit-identical losses, parameters and running statistics, whether the step
    is replayed from a hipGraph with weight gradients on a side stream or launched kernel by kernel on one stream."""
    l_graph, p_graph, rv_graph, ts = _train(True, 5)
    assert ts.graph is not None, 'hipGraph capture fell back to eager launches'
    l_eager, p_eager, rv_eager, _ = _train(False, 5, overlap=False)
    l_again, p_again, _, _ = _train(True, 5)
    assert torch.isfinite(l_graph).all()
    assert torch.equal(l_graph, l_again) and torch.equal(p_graph, p_again)
    assert torch.equal(l_graph, l_eager) and torch.equal(p_graph, p_eager) and torch.equal(rv_graph, rv_eager)
    assert int(ts.reg.image_encoder.layer4[1].bn2.num_batches_tracked) == 5
    # the multi-GPU capture (two hipGraphs split where the layer3.. gradients are final, the all-reduce of that bucket starts
    # in between) runs the same kernels: identical results on one GPU, where the exchange itself is a no-op
    l_split, p_split, rv_split, ts2 = _train(True, 5, comm_overlap=True)
    assert ts2.graph is not None and ts2.graph_tail is not None, 'split capture fell back'
    assert ts2.exchange.split_off > 0
    assert torch.equal(l_graph, l_split) and torch.equal(p_graph, p_split) and torch.equal(rv_graph, rv_split)
    l_se, p_se, _, _ = _train(False, 5, comm_overlap=True)
    assert torch.equal(l_graph, l_se) and torch.equal(p_graph, p_se)
    # the data pipeline (next batch generated on a second stream during the step) does not change a bit either
    l_np, p_np, _, _ = _train(False, 5, overlap=False, pipeline=False)
    assert torch.equal(l_graph, l_np) and torch.equal(p_graph, p_np)
    l_npg, p_npg, _, _ = _train(True, 5, pipeline=False)
    assert torch.equal(l_graph, l_npg) and torch.equal(p_graph, p_npg)


def test_smpl_bench_size_slices_are_batch_independent():
    """configs[4] at the bench size (65 536 bodies per launch): a slice computed alone equals the same rows of the big launch."""
    model = straps_amd.synthetic_smpl_model(0)
    B = 65536
    smpl = straps_amd.SMPL(model, batch_size=1).to(DEV)
    g = torch.Generator().manual_seed(11)
    betas = (torch.randn(B, 10, generator=g) * 1.5).to(DEV)
    R = O.batch_rodrigues((torch.randn(B * 24, 3, generator=g) * 0.3)).view(B, 24, 3, 3).to(DEV).contiguous()
    v, j = smpl.forward_arrays(betas, R)
    assert torch.isfinite(v).all() and torch.isfinite(j).all()
    for lo, n in ((0, 33), (31999, 70), (65500, 36)):
        vs, js = smpl.forward_arrays(betas[lo:lo + n].contiguous(), R[lo:lo + n].contiguous())
        assert torch.equal(vs, v[lo:lo + n]) and torch.equal(js, j[lo:lo + n])
    # and three of them against the oracle
    vo, jo = O.smpl_forward(model, betas[:3].cpu(), rotmats=R[:3].cpu())
    assert float((v[:3].cpu() - vo).abs().max()) < 1e-5 and float((j[:3].cpu() - jo).abs().max()) < 1e-5
    # the same through the kernel bench.py --config 4 times (fp16x3_lbs at this size = the 64-body kernel): slices alone == rows of the big launch
    vw, jw = smpl.forward_arrays(betas, R, precision='fp16x3_lbs')
    assert torch.isfinite(vw).all() and float((vw - v).abs().max()) < 1e-5 and float((jw - j).abs().max()) < 1e-5
    for lo, n in ((0, 33), (31999, 70), (65500, 36)):
        vs, js = smpl.forward_arrays(betas[lo:lo + n].contiguous(), R[lo:lo + n].contiguous(), precision='fp16x3_lbs', kernel='wide')
        assert torch.equal(vs, vw[lo:lo + n]) and torch.equal(js, jw[lo:lo + n])


def test_rasteriser_b64_equals_per_body_renders():
    model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(model, batch_size=64).to(DEV)
    betas = torch.from_numpy(det_uniform((64, 10), 21, -2.0, 2.0)).to(DEV)
    aa = det_uniform((64, 24, 3), 22, -0.25, 0.25)
    aa[:, 0, 1] = det_uniform((64,), 23, -1.5, 1.5)
    R = straps_amd.batch_rodrigues(torch.from_numpy(aa).reshape(-1, 3).to(DEV)).view(64, 24, 3, 3).contiguous()
    verts, _ = smpl.forward_arrays(betas, R)
    cam_t = torch.from_numpy(np.stack([det_uniform((64,), 24, -0.1, 0.1), det_uniform((64,), 25, 0.1, 0.3), det_uniform((64,), 26, 37.0, 47.0)], 1)).to(DEV)
    K = O.intrinsics_matrix().astype(np.float32)
    r = straps_amd.NMRRenderer(64, K, np.eye(3, dtype=np.float32), 256, rend_parts_seg=True, faces=smpl.faces, face_parts=smpl.face_parts).to(DEV)
    big = r.render_arrays(verts, cam_t)
    for b in (0, 17, 63):
        one = r.render_arrays(verts[b:b + 1].contiguous(), cam_t[b:b + 1].contiguous())
        assert torch.equal(one[0], big[b])
    assert 0.02 < float((big > 0).float().mean()) < 0.6
