"""CPU model (numpy) of the three-plane bf16 operand representation of csrc/conv_x3.hip -- TEST INFRASTRUCTURE, like the rest of oracle/:
only tests/ import it; the product path never does.

It restates no reference code: the reference computes its convolutions in fp32 (models/resnet.py:28-36 -> cuDNN / ATen).  What is
modelled is the build's own way of reaching fp32 accuracy on the bf16 matrix pipe, so that its numerics claims can be checked without a GPU:

  split3(x)      -- common.h `split3`: b1 = rn_bf16(x), b2 = rn_bf16(x - b1), b3 = rn_bf16(x - b1 - b2); integer round-to-nearest-even
                    on the fp32 bit pattern, the two subtractions exact in fp32.  x == b1 + b2 + b3 for 2^-110 <= |x| <= bf16's largest
                    finite value (3.3895e38; above it the leading plane rounds to infinity).
  dot_x3(a, b)   -- one output element of the implicit GEMM: per 16-wide k block the six plane products of weight >= 2^-16
                    (a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2, smallest first as the kernel issues them), every product exact (8 x 8
                    significand bits), accumulated in fp32.  The order of the additions inside one MFMA instruction is not architected;
                    the model adds a block's 16 products in float64 and rounds once per instruction -- it bounds the error, it does
                    not reproduce the accumulator bit for bit (the GPU tests compare against float64 with a tolerance for that reason).
"""
import numpy as np

TA = (1, 0, 2, 0, 1, 0)      # plane of A / of B in the six products, in issue order (conv_x3.hip)
TB = (1, 2, 0, 1, 0, 0)


def bf16_rn_bits(x):
    """fp32 array -> uint16 bf16 bit patterns, round to nearest even on the bit pattern (common.h `bf16_rn`)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFFFFFF
    return (u >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def split3(x):
    """fp32 array -> (bits [3, ...] uint16, values [3, ...] float32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b1 = bf16_rn_bits(x)
    with np.errstate(invalid='ignore'):
        r1 = np.where(np.isinf(x), np.float32(0), x - bf16_bits_to_f32(b1)).astype(np.float32)      # an infinite value keeps zero residues (csrc/common.h)
    b2 = bf16_rn_bits(r1)
    r2 = r1 - bf16_bits_to_f32(b2)
    b3 = bf16_rn_bits(r2)
    bits = np.stack([b1, b2, b3])
    return bits, bf16_bits_to_f32(bits)


def dot_x3(a, b):
    """a [..., K], b [..., K] fp32 (K % 16 == 0) -> fp32 dot products as the x3 kernel evaluates them (see the module docstring)."""
    _, ap = split3(a)
    _, bp = split3(b)
    K = a.shape[-1]
    assert K % 16 == 0
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), np.float32)
    for k0 in range(0, K, 16):
        for t in range(6):
            prod = ap[TA[t]][..., k0:k0 + 16].astype(np.float64) * bp[TB[t]][..., k0:k0 + 16].astype(np.float64)     # exact
            acc = (acc.astype(np.float64) + prod.sum(-1)).astype(np.float32)
    return acc


def dot_fp32_chain(a, b):
    """the exact-fp32 MFMA chain of csrc/conv.hip: one fused multiply-add per k, fp32 accumulator (fmaf: the product is not rounded)."""
    K = a.shape[-1]
    acc = np.zeros(np.broadcast_shapes(a.shape[:-1], b.shape[:-1]), np.float32)
    for k in range(K):
        acc = (acc.astype(np.float64) + a[..., k].astype(np.float64) * b[..., k].astype(np.float64)).astype(np.float32)
    return acc
