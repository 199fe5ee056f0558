"""Counter-based deterministic value generator (TEST INFRASTRUCTURE -- part of oracle/).

The golden fixtures under tests/golden/ store only *outputs*; the matching inputs and network
weights are far too large to commit (a resnet18 state dict is 47.6 MB), so they are regenerated
bit-identically on every machine from this integer hash.  Nothing here depends on numpy's or
torch's RNG streams: it is pure uint64 arithmetic (splitmix64), so a value depends only on
(seed, index).

Used by: oracle/make_golden.py (which feeds the same tensors to the imported reference),
tests/ (oracle-vs-golden on CPU, HIP-vs-oracle on GPU).
"""
import zlib

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array of `shape`, values uniform on a 2^-24 grid in [lo, hi). Pure function of (seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _GAMMA + np.uint64(seed) * _M2
        bits = _splitmix64(_splitmix64(idx)) >> np.uint64(40)      # 24 random bits
    u = bits.astype(np.float64) * (1.0 / 16777216.0)               # [0,1) exactly representable in f32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def key_seed(name, base=0):
    """Stable 32-bit seed from a tensor name (crc32) so each state-dict entry gets its own stream."""
    return (zlib.crc32(name.encode('utf-8')) + 7919 * base) & 0xFFFFFFFF


def det_state_dict(shapes, base=0):
    """Deterministic, well-conditioned values for a regressor state dict.

    `shapes`: ordered {key: shape} (the manifest in tests/golden/state_dict_keys_*.json).
    conv / linear weights ~ U * sqrt(3 * 2 / fan_in) (He-variance so activations stay O(1) through
    53 layers), BN gamma ~ 1 +- 0.1, beta / running_mean ~ +-0.1, running_var in [0.8, 1.2],
    linear biases ~ +-0.01, num_batches_tracked = 0.
    Returns {key: np.ndarray} (float32; int64 for num_batches_tracked).
    """
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        # ief_layers.{0,2,4} are the SAME modules as fc{1,2,3} in the reference
        # (models/ief_module.py:24-28) -> identical values under both key families
        canon = k.replace('ief_layers.0.', 'fc1.').replace('ief_layers.2.', 'fc2.').replace('ief_layers.4.', 'fc3.')
        s = key_seed(canon, base)
        if k.endswith('num_batches_tracked'):
            out[k] = np.zeros(shp, dtype=np.int64)
        elif k.endswith('running_var'):
            out[k] = det_uniform(shp, s, 0.8, 1.2)
        elif k.endswith('running_mean'):
            out[k] = det_uniform(shp, s, -0.1, 0.1)
        elif len(shp) == 4:                                   # conv OIHW
            fan_in = shp[1] * shp[2] * shp[3]
            a = float(np.sqrt(6.0 / fan_in))
            out[k] = det_uniform(shp, s, -a, a)
        elif len(shp) == 2:                                   # linear [out, in]
            a = 0.4 * float(np.sqrt(3.0 / shp[1]))
            out[k] = det_uniform(shp, s, -a, a)
        elif '.bn' in k or 'downsample.1' in k or k.startswith('image_encoder.bn1'):
            if k.endswith('.weight'):
                out[k] = det_uniform(shp, s, 0.9, 1.1)
            else:
                out[k] = det_uniform(shp, s, -0.1, 0.1)
        else:                                                 # linear bias
            out[k] = det_uniform(shp, s, -0.01, 0.01)
    return out


def det_metrics_case(npts, seed, batch=3):
    """(pred, target) float32 [batch,npts,3] for the evaluation-metric goldens: the prediction is a rotated, scaled,
    shifted and noisy copy of the target, so the alignment steps matter."""
    tv = det_uniform((batch, npts, 3), seed, -1.0, 1.0).astype(np.float64)
    ang = 0.4
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.0]])
    pv = 1.1 * tv.dot(Rz.T) + np.array([0.05, -0.02, 0.1]) + 0.03 * det_uniform((batch, npts, 3), seed + 1, -1.0, 1.0)
    return pv.astype(np.float32), tv.astype(np.float32)


def det_crop_case(batch=5, wh=256):
    """(seg [B,wh,wh] part ids, joints2d [B,17,2]) for the crop/resize goldens: a blob per sample, two of them touching
    the image border so that the corner clamping paths are exercised."""
    seg = np.zeros((batch, wh, wh), np.float32)
    boxes = [(60, 200, 80, 170), (0, 140, 30, 120), (100, 256, 90, 256), (20, 60, 10, 250), (120, 135, 120, 131)]
    for i in range(batch):
        r0, r1, c0, c1 = boxes[i % len(boxes)]
        blob = np.floor(det_uniform((r1 - r0, c1 - c0), 300 + i, 0.0, 6.999)).astype(np.float32)
        seg[i, r0:r1, c0:c1] = blob
        seg[i, r0, c0] = 1.0
        seg[i, r1 - 1, c1 - 1] = 2.0
    joints = det_uniform((batch, 17, 2), 310, 10.0, 246.0).astype(np.float32)
    return seg, joints
