"""GPU: the part-segmentation rasteriser (straps_rasterize_parts / NMRRenderer) against the oracle's
restatement of neural_renderer's algorithm.  Integer output -> the bar is bit-exact (the kernel and the oracle do the
same unfused fp32 arithmetic in the same order)."""
import numpy as np
import pytest
import torch

import straps_amd
import straps_oracle as O
from detgen import det_uniform
from straps_amd import hipabi

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _random_scene(B, N, F, seed):
    v = det_uniform((B, N, 3), seed, -1.0, 1.0) * np.array([0.5, 0.9, 0.3], np.float32)
    faces = np.floor(det_uniform((F, 3), seed + 1, 0.0, N - 1e-3)).astype(np.int32)
    faces[1] = faces[0]                                   # duplicate face: the tie goes to the lower id
    faces[2, 1] = faces[2, 0]                             # degenerate face
    parts = (1 + np.floor(det_uniform((F,), seed + 2, 0.0, 5.999))).astype(np.uint8)
    parts[1] = 6 if parts[0] != 6 else 5
    return v, faces, parts


def test_random_scenes_bit_exact_with_per_body_cameras():
    B, N, F, wh = 3, 200, 320, 64
    v, faces, parts = _random_scene(B, N, F, 900)
    K = np.stack([O.intrinsics_matrix(wh, wh, f) for f in (60.0, 90.0, 140.0)]).astype(np.float32)
    ang = np.array([0.0, 0.4, -0.7], np.float32)
    R = np.stack([np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32) for a in ang])
    t = np.array([[0.0, 0.0, 2.5], [0.1, -0.2, 3.0], [-0.3, 0.1, 1.2]], np.float32)      # body 2 crosses the near plane
    want, wdepth = O.rasterize_parts(v, faces, parts, K, R, t, wh=wh, return_depth=True)
    r = straps_amd.NMRRenderer(B, K, R, img_wh=wh, rend_parts_seg=True, faces=faces, face_parts=parts).to(DEV)
    got, gdepth = r.render_arrays(torch.from_numpy(v).to(DEV), torch.from_numpy(t).to(DEV), want_depth=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(gdepth.cpu().numpy(), wdepth)
    assert (want > 0).mean() > 0.2
    out = r(torch.from_numpy(v).to(DEV), torch.from_numpy(t).to(DEV)[:, None])      # module call: [B,1,3] cam_ts, long ids
    assert out.dtype == torch.long and tuple(out.shape) == (B, wh, wh)
    np.testing.assert_array_equal(out.cpu().numpy(), want.astype(np.int64))
    again = r.render_arrays(torch.from_numpy(v).to(DEV), torch.from_numpy(t).to(DEV))
    assert torch.equal(again, got)                                                  # deterministic


def test_posed_synthetic_bodies_full_size_bit_exact():
    model = straps_amd.synthetic_smpl_model(0)
    smpl = straps_amd.SMPL(model, batch_size=2).to(DEV)
    betas = torch.from_numpy(det_uniform((2, 10), 910, -1.5, 1.5)).to(DEV)
    aa = det_uniform((2, 24, 3), 911, -0.3, 0.3)
    aa[1, 0] = [0.0, 1.2, 0.0]
    R = straps_amd.batch_rodrigues(torch.from_numpy(aa).reshape(-1, 3).to(DEV)).view(2, 24, 3, 3)
    verts, _ = smpl.forward_arrays(betas, R.contiguous())
    cam_t = torch.tensor([[0.0, 0.2, 42.0], [0.05, 0.15, 38.0]], device=DEV)
    K = O.intrinsics_matrix().astype(np.float32)
    r = straps_amd.NMRRenderer(2, K, np.eye(3, dtype=np.float32), 256, rend_parts_seg=True, faces=smpl.faces, face_parts=smpl.face_parts).to(DEV)
    got = r.render_arrays(verts, cam_t)
    want = O.rasterize_parts(verts.cpu().numpy(), model['faces'], model['face_parts'], K, np.eye(3), cam_t.cpu().numpy())
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert set(np.unique(want)) == {0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0}
    # the projected COCO joints of the same bodies land on the silhouette's bounding box (same camera convention as P2)
    ys, xs = np.nonzero(want[0])
    assert xs.min() >= 0 and xs.max() <= 255 and 40 < ys.max() - ys.min() < 256


def test_error_behaviour():
    K, I = O.intrinsics_matrix().astype(np.float32), np.eye(3, dtype=np.float32)
    with pytest.raises(RuntimeError, match='out of scope'):
        straps_amd.NMRRenderer(1, K, I, 256, rend_parts_seg=False, faces=np.zeros((1, 3)), face_parts=np.zeros(1))
    r = straps_amd.NMRRenderer(1, K, I, 256, rend_parts_seg=True, faces=np.zeros((4, 3)), face_parts=np.ones(4))
    with pytest.raises(RuntimeError, match='GPU tensor'):
        r(torch.zeros(1, 5, 3), torch.zeros(1, 3))
    L = hipabi.load()
    assert L.straps_rasterize_parts(None, None, None, None, None, None, None, None, None, 1, 1, 1, 8, 0, 0.1, 100.0, None, 0.0, 0.0, None) != 0
    assert b'null pointer' in L.straps_last_error()
    r = r.to(DEV)
    out = r(torch.zeros(1, 5, 3, device=DEV), torch.tensor([[0.0, 0.0, 5.0]], device=DEV))      # all faces degenerate: empty image
    assert int(out.abs().sum()) == 0
