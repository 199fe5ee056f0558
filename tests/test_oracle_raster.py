"""CPU: properties of the part-segmentation rasteriser oracle.  The reference delegates to the
third-party `neural_renderer` extension, which is absent -> no golden vectors exist for this row (parity unpinned);
these tests pin the restated algorithm's conventions instead."""
import numpy as np

from oracle import straps_oracle as O

K32 = O.intrinsics_matrix(32, 32, 100.0)
T5 = np.array([[0., 0., 5.]], np.float32)


def _render(v, faces, parts, **kw):
    return O.rasterize_parts(np.asarray(v, np.float32)[None], faces, np.asarray(parts, np.uint8), K32, np.eye(3), T5, wh=32, **kw)


def test_pixel_convention_matches_perspective_projection():
    # pixel (row i, col j) samples the image plane at (u, v) = (j + 0.5, i + 0.5) in the pixel coordinates of
    # utils/cam_utils.py perspective_project (same K): a small triangle around a 3D point covers its projected pixel
    p = np.array([0.31, -0.22, 0.4], np.float32)
    tri = p + np.array([[-0.06, -0.05, 0], [0.06, -0.05, 0], [0, 0.07, 0]], np.float32)
    img = _render(tri, [[0, 1, 2]], [4])[0]
    import torch
    uv = O.perspective_project(torch.tensor(p)[None, None], torch.eye(3)[None], torch.tensor(T5), torch.tensor(K32, dtype=torch.float32)[None])[0, 0].numpy()
    assert img[int(np.floor(uv[1])), int(np.floor(uv[0]))] == 4
    assert img.sum() > 0 and set(np.unique(img)) == {0.0, 4.0}


def test_area_and_two_sidedness():
    sq = [[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0]]
    a = _render(sq, [[0, 1, 2], [0, 2, 3]], [1, 1])[0]
    b = _render(sq, [[2, 1, 0], [3, 2, 0]], [1, 1])[0]                  # reversed winding renders identically (fill_back)
    np.testing.assert_array_equal(a, b)
    # the 1 x 1 square at depth 5 with f = 100 is 20 x 20 pixels, centred
    assert a.sum() == 400 and a[6:26, 6:26].sum() == 400


def test_depth_order_ties_and_clipping():
    near_tri = [[-0.5, -0.5, -1], [0.5, -0.5, -1], [0, 0.5, -1]]
    far_tri = [[-0.5, -0.5, 0], [0.5, -0.5, 0], [0, 0.5, 0]]
    v = near_tri + far_tri
    img, depth = _render(v, [[3, 4, 5], [0, 1, 2]], [2, 5], return_depth=True)
    assert img[0][16, 16] == 5 and depth[0][16, 16] == np.float32(4.0)      # the nearer face wins regardless of order
    assert depth[0][0, 0] == np.float32(100.0) and img[0][0, 0] == 0         # background: part 0, depth = far
    dup = _render(far_tri, [[0, 1, 2], [0, 1, 2]], [3, 6])[0]
    assert set(np.unique(dup)) == {0.0, 3.0}                                  # equal depth: the first face keeps the pixel
    assert _render(far_tri, [[0, 1, 2]], [3], near=5.5)[0].sum() == 0        # in front of the near plane
    assert _render(far_tri, [[0, 1, 2]], [3], far=4.5)[0].sum() == 0         # beyond the far plane
    behind = np.asarray(far_tri, np.float32) - np.array([0, 0, 10], np.float32)
    assert _render(behind, [[0, 1, 2]], [3])[0].sum() == 0                   # behind the camera
    assert _render(far_tri, [[0, 0, 1]], [3])[0].sum() == 0                  # degenerate face


def test_synthetic_body_renders_all_parts():
    import straps_amd
    m = straps_amd.synthetic_smpl_model(0)
    K = O.intrinsics_matrix(64, 64, 1250.0)
    img = O.rasterize_parts(m['v_template'][None], m['faces'][::7], m['face_parts'][::7], K, np.eye(3), np.array([[0., 0.2, 42.]]), wh=64)[0]
    assert set(np.unique(img)) == {0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0}
    assert 0.05 < (img > 0).mean() < 0.6
