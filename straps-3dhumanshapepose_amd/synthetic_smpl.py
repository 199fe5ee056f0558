"""SMPL model data: loader for a real model file + a seeded synthetic SMPL-shaped stand-in.

The real `SMPL_NEUTRAL.pkl` and the three extra joint regressors (reference config.py:4-8) cannot
be redistributed or downloaded here, so benchmarks and tests run on a synthetic model with the
same tensor shapes, sparsity pattern and magnitudes:

  v_template[6890,3]  shapedirs[6890,3,10]  posedirs[207,20670]  J_regressor[24,6890] (sparse rows)
  weights[6890,24] (<=4 non-zeros per vertex, rows sum to 1)  parents[24]  extra_vertex_ids[21]
  J_regressor_extra[9,6890]  J_regressor_cocoplus[19,6890]  J_regressor_h36m[17,6890]
  faces[13776,3] (stand-in for additional/smpl_faces.npy)  face_parts[13776] (1..6, stand-in for the part texture)

`load_smpl_model` reads the real files when a user has them (same dict out), so the SMPL module
is a drop-in for reference `models/smpl_official.py:15-25`.
"""
import os
import pickle

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_FEATS = 207
NUM_FACES = 13776

# 6-part convention of renderers/nmr_renderer.py:12-20 (0 background, 1 left arm, 2 right arm, 3 head, 4 left leg,
# 5 right leg, 6 torso) assigned through each vertex's dominant SMPL joint
JOINT_TO_PART = np.array([6, 4, 5, 6, 4, 5, 6, 4, 5, 6, 4, 5, 3, 6, 6, 3, 1, 2, 1, 2, 1, 2, 1, 2], dtype=np.uint8)

# kinematic tree of the SMPL body model (published with the model)
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                         20, 21], dtype=np.int32)

# vertices appended by smplx's VertexJointSelector for SMPL (joints 24..44 of the 90-joint output):
# face (nose, reye, leye, rear, lear), feet (L big toe, L small toe, L heel, R ...), finger tips
# (l thumb..pinky, r thumb..pinky).  Published vertex ids of the SMPL topology.
EXTRA_VERTEX_IDS = np.array([332, 6260, 2800, 4071, 583,
                             3216, 3226, 3387, 6617, 6624, 6787,
                             2746, 2319, 2445, 2556, 2673,
                             6191, 5782, 5905, 6016, 6133], dtype=np.int32)

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _u(shape, seed, lo=-1.0, hi=1.0):
    """platform-independent uniform floats from a splitmix64 counter hash."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _G + np.uint64(seed) * _M2
        for _ in range(2):
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / 16777216.0
    return (lo + (hi - lo) * u).reshape(shape)


def _sparse_rows(centres, verts, nnz, seed):
    """row-stochastic [len(centres), V] matrix supported on the `nnz` vertices nearest each centre."""
    d = np.linalg.norm(verts[None, :, :] - centres[:, None, :], axis=2)          # [R,V]
    idx = np.argsort(d, axis=1, kind='stable')[:, :nnz]
    w = _u(idx.shape, seed, 0.2, 1.0)
    w /= w.sum(axis=1, keepdims=True)
    out = np.zeros((centres.shape[0], verts.shape[0]), dtype=np.float64)
    np.put_along_axis(out, idx, w, axis=1)
    return out


_MODEL_CACHE = {}


def synthetic_smpl_model(seed=0):
    """cached front end of _build_synthetic_smpl_model (fresh array copies per call)."""
    if seed not in _MODEL_CACHE:
        _MODEL_CACHE[seed] = _build_synthetic_smpl_model(seed)
    return {k: v.copy() for k, v in _MODEL_CACHE[seed].items()}


def _build_synthetic_smpl_model(seed=0):
    """Seeded SMPL-shaped model (float32 arrays).  Magnitudes follow the real model: template spans
    a ~1.7 m body-sized box, shapedirs ~1e-2 m/unit beta, posedirs ~1e-3 m/unit pose feature."""
    s = 1000 * seed
    box = np.array([0.45, 0.85, 0.15])
    verts = _u((NUM_VERTS, 3), s + 1) * box
    # a crude skeleton inside the box: joints placed along limbs so that skinning is non-trivial
    jc = _u((NUM_JOINTS, 3), s + 2) * box * 0.9
    jc[0] = 0.0
    # skinning weights: 4 nearest joints, positive, normalised (<=4-sparse rows)
    d = np.linalg.norm(verts[:, None, :] - jc[None, :, :], axis=2)                # [V,24]
    near = np.argsort(d, axis=1, kind='stable')[:, :4]
    dn = np.take_along_axis(d, near, axis=1)
    w4 = np.exp(-25.0 * (dn - dn[:, :1])) + 1e-3
    # roughly a third of the vertices -- those much closer to one joint than to any other -- are rigidly bound to
    # it, like the real model; spatially coherent, so neighbouring vertices move together under a pose
    rigid = (dn[:, 1] - dn[:, 0]) > np.quantile(dn[:, 1] - dn[:, 0], 0.67)
    w4[rigid, 1:] = 0.0
    w4 /= w4.sum(axis=1, keepdims=True)
    weights = np.zeros((NUM_VERTS, NUM_JOINTS))
    np.put_along_axis(weights, near, w4, axis=1)
    # triangles over the point cloud: each vertex with its nearest neighbours (SMPL-sized faces of a few cm), and one
    # part label per face from its first vertex's dominant joint
    faces = np.zeros((NUM_FACES, 3), np.int32)
    for lo in range(0, NUM_VERTS, 1024):
        blk = verts[lo:lo + 1024]
        dd = sum((blk[:, c:c + 1] - verts[None, :, c]) ** 2 for c in range(3))         # squared distances
        cand = np.argpartition(dd, 4, axis=1)[:, :4]                                   # self + 3 nearest, unordered
        cd = np.take_along_axis(dd, cand, axis=1)
        nn = np.take_along_axis(cand, np.lexsort((cand, cd), axis=1), axis=1)[:, 1:4]  # by distance, then index
        idx = np.arange(lo, min(lo + 1024, NUM_VERTS))
        for k in range(2):
            rows = 2 * idx + k
            ok = rows < NUM_FACES
            faces[rows[ok]] = np.stack([idx[ok], nn[ok, k], nn[ok, k + 1]], axis=1)
    face_parts = JOINT_TO_PART[np.argmax(weights, axis=1)][faces[:, 0]]
    # shape directions: smooth displacement fields over the body (a few low spatial frequencies, ~1e-2 m per unit
    # beta like the real model) plus 5 % per-vertex detail, so a shaped mesh keeps SMPL-sized faces
    freq = _u((NUM_BETAS, 3, 4, 3), s + 13) * np.array([6.0, 3.5, 12.0])              # [l][c][m][xyz] rad / m
    phase = _u((NUM_BETAS, 3, 4), s + 14) * np.pi
    amp = _u((NUM_BETAS, 3, 4), s + 15) * 0.5e-2
    arg = np.einsum('lcmk,vk->vclm', freq, verts) + phase.transpose(1, 0, 2)[None]    # [v][c][l][m]
    shapedirs = (np.cos(arg) * amp.transpose(1, 0, 2)[None]).sum(-1) + _u((NUM_VERTS, 3, NUM_BETAS), s + 4) * 5e-4
    model = {
        'faces': faces, 'face_parts': face_parts,
        'v_template': verts,
        'shapedirs': shapedirs,
        'posedirs': _u((NUM_POSE_FEATS, NUM_VERTS * 3), s + 5) * 1e-3,
        'J_regressor': _sparse_rows(jc, verts, 32, s + 6),
        'weights': weights,
        'parents': SMPL_PARENTS.copy(),
        'extra_vertex_ids': EXTRA_VERTEX_IDS.copy(),
        'J_regressor_extra': _sparse_rows(_u((9, 3), s + 7) * box, verts, 24, s + 8),
        'J_regressor_cocoplus': _sparse_rows(_u((19, 3), s + 9) * box, verts, 48, s + 10),
        'J_regressor_h36m': _sparse_rows(_u((17, 3), s + 11) * box, verts, 96, s + 12),
    }
    return {k: (v.astype(np.float32) if v.dtype.kind == 'f' else v) for k, v in model.items()}


def synthetic_mean_params(seed=0):
    """stand-in for additional/neutral_smpl_mean_params_6dpose.npz (keys 'pose'[144], 'shape'[10]):
    near-identity 6D rotations (interleaved a1x,a2x,a1y,a2y,a1z,a2z) plus small noise."""
    ident = np.tile(np.array([1., 0., 0., 1., 0., 0.]), 24)
    pose = ident + 0.05 * _u((144,), 77 + seed)
    shape = 0.2 * _u((10,), 78 + seed)
    return {'pose': pose.astype(np.float32), 'shape': shape.astype(np.float32)}


def _dense(a):
    return np.asarray(a.todense() if hasattr(a, 'todense') else a)


def load_smpl_model(model_path, gender='neutral', extra_regressor_paths=None):
    """Load a real SMPL model: `model_path` may be the directory holding SMPL_<GENDER>.pkl (what
    reference run_train.py:109 passes as config.SMPL_MODEL_DIR), a .pkl or an .npz with the keys
    above.  `extra_regressor_paths` = (extra, cocoplus, h36m) .npy files (reference config.py:6-8);
    defaults to the reference's relative locations."""
    if os.path.isdir(model_path):
        model_path = os.path.join(model_path, 'SMPL_%s.pkl' % gender.upper())
    if not os.path.isfile(model_path):
        raise FileNotFoundError('SMPL model file not found: %s' % model_path)
    if model_path.endswith('.npz'):
        raw = dict(np.load(model_path, allow_pickle=True))
    else:
        with open(model_path, 'rb') as f:
            raw = pickle.load(f, encoding='latin1')
    posedirs = np.asarray(raw['posedirs'], dtype=np.float64)
    if posedirs.ndim == 3:                                    # [6890,3,207] as stored in the pkl
        posedirs = posedirs.reshape(-1, posedirs.shape[-1]).T
    parents = raw['parents'] if 'parents' in raw else np.asarray(raw['kintree_table'])[0].astype(np.int64)
    parents = np.asarray(parents, dtype=np.int64).copy()
    parents[0] = -1
    model = {
        'v_template': np.asarray(raw['v_template'], dtype=np.float32),
        'shapedirs': np.asarray(raw['shapedirs'], dtype=np.float32)[:, :, :NUM_BETAS],
        'posedirs': posedirs.astype(np.float32),
        'J_regressor': _dense(raw['J_regressor']).astype(np.float32),
        'weights': np.asarray(raw['weights'], dtype=np.float32),
        'parents': parents.astype(np.int32),
        'extra_vertex_ids': np.asarray(raw.get('extra_vertex_ids', EXTRA_VERTEX_IDS), dtype=np.int32),
    }
    if extra_regressor_paths is None:
        extra_regressor_paths = ('additional/J_regressor_extra.npy', 'additional/cocoplus_regressor.npy',
                                 'additional/J_regressor_h36m.npy')
    for key, path in zip(('J_regressor_extra', 'J_regressor_cocoplus', 'J_regressor_h36m'),
                         extra_regressor_paths):
        model[key] = np.asarray(raw[key] if key in raw else np.load(path), dtype=np.float32)
    if 'f' in raw or 'faces' in raw:                          # mesh topology (smplx exposes it as .faces)
        model['faces'] = np.asarray(raw['faces'] if 'faces' in raw else raw['f']).astype(np.int32)
    if 'face_parts' in raw:
        model['face_parts'] = np.asarray(raw['face_parts']).astype(np.uint8)
    return model
