"""CPU checks of the numerics the bf16x3 convolution route (csrc/conv_x3.hip) rests on, through the numpy model oracle/bf16x3_emul.py:
no GPU needed.  The GPU side of the same claims -- the kernel's planes equal this model's bit for bit, the convolutions against
float64 -- is tests/test_gpu_conv_x3.py.

What the reference computes here is an fp32 convolution (models/resnet.py:28-36); the bars below say that the three-plane route is
an fp32-accuracy evaluation of it, not a reduced-precision one.
"""
import numpy as np

import bf16x3_emul as em


def _values(rng, n):
    return np.concatenate([rng.standard_normal(n).astype(np.float32),
                           (rng.standard_normal(n // 8) * 1e-20).astype(np.float32), (rng.standard_normal(n // 8) * 1e20).astype(np.float32),
                           np.array([0.0, -0.0, 1.0, -1.0, 2.0 ** -100, 2.0 ** 100, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0, 255.0, 257.0,
                                     3.3895313892515355e38], np.float32)])


def test_three_planes_reconstruct_fp32_exactly():
    """3 x 8 significand bits cover fp32's 24: x == b1 + b2 + b3 bit for bit over the normal range (the bar for a representation
    change is exactness, not a tolerance); the planes are ordered by magnitude, 2^-8 apart."""
    x = _values(np.random.default_rng(11), 200000)
    _, p = em.split3(x)
    assert np.array_equal(p.astype(np.float64).sum(0), x.astype(np.float64))
    nz = p[0] != 0
    assert np.all(np.abs(p[1][nz]) <= np.abs(p[0][nz]) * 2.0 ** -8)
    assert np.all(np.abs(p[2][nz]) <= np.abs(p[0][nz]) * 2.0 ** -16)


def test_split_limits_are_where_the_documentation_puts_them():
    """below 2^-110 the residue falls under bf16's smallest subnormal (error <= 2^-133); above bf16's largest finite value the leading
    plane rounds to infinity -- neither occurs in a network whose activations and gradients are O(1e-10 .. 1e4)."""
    tiny = (np.random.default_rng(3).standard_normal(4096) * 1e-36).astype(np.float32)
    _, p = em.split3(tiny)
    assert np.abs(p.astype(np.float64).sum(0) - tiny.astype(np.float64)).max() <= 2.0 ** -133
    with np.errstate(over='ignore', invalid='ignore'):
        _, p = em.split3(np.array([3.4e38], np.float32))
    assert np.isinf(p[0][0])


def test_dropped_products_are_below_fp32_rounding():
    """a*b - (six products) = a2*b3 + a3*b2 + a3*b3, each factor pair at most 2^-24 of |a*b| up to the rounding slack of the leading
    planes: the truncation is <= 2^-22 |a*b| per term (bar written as that bound), i.e. of the size of ONE fp32 rounding of the
    product -- an error the fp32 chain does not commit per product but commits per accumulation."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal(100000).astype(np.float32)
    b = (rng.standard_normal(100000) * 0.05).astype(np.float32)
    _, ap = em.split3(a)
    _, bp = em.split3(b)
    six = sum(ap[i].astype(np.float64) * bp[j].astype(np.float64) for i, j in zip(em.TA, em.TB))
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -22, rel.max()


def test_dot_products_match_the_fp32_chain_class():
    """resnet-sized reductions (K = 576 ... 4608, activations ~ N(0,1) after BatchNorm + ReLU, weights ~ kaiming): the error of the
    six-product evaluation against float64 stays within 1.5x of the exact-fp32 chain's on the same operands (measured here: it is
    SMALLER -- the chain rounds once per k, the matrix instruction once per 16 k and product) and under 2e-5 of the largest output,
    the bar tests/test_gpu_forward.py holds the fp32 kernels to."""
    rng = np.random.default_rng(9)
    for K in (576, 1152, 2304, 4608):
        a = np.maximum(rng.standard_normal((256, K)), 0).astype(np.float32)
        b = (rng.standard_normal((256, K)) * np.sqrt(2.0 / K)).astype(np.float32)
        ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        e3 = np.abs(em.dot_x3(a, b).astype(np.float64) - ref).max()
        e1 = np.abs(em.dot_fp32_chain(a, b).astype(np.float64) - ref).max()
        assert e3 <= 1.5 * e1 + 1e-9, (K, e3, e1)
        assert e3 <= 2e-5 * np.abs(ref).max(), (K, e3)
