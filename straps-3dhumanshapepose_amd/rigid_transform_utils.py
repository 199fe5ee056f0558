"""Rotation representations on the GPU (reference utils/rigid_transform_utils.py:27-41 and the
`smplx.lbs.batch_rodrigues` the reference imports at train loop :5, predict_3D.py:5)."""
import torch

from . import hipabi


def rot6d_to_rotmat(x):
    """6-D rotation representation (Zhou et al.) -> rotation matrices, same contract as the
    reference: x[..., 6k] is read as [-1,3,2] (interleaved), output [-1,3,3] with columns b1,b2,b3.
    Accepts the non-contiguous `pose` view of the IEF estimate directly (row stride passed down)."""
    hipabi.require_gpu_tensor(x, 'rot6d input', torch.float32)
    if torch.is_grad_enabled() and x.requires_grad:
        from .autograd_ops import rot6d_autograd
        return rot6d_autograd(x)
    return _rot6d_fwd(x.detach())


def _rot6d_fwd(x):
    if x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 6 == 0 and x.stride(0) >= x.shape[1]:
        rows, per_row, ld = x.shape[0], x.shape[1] // 6, x.stride(0)
    else:
        x = x.contiguous().view(-1, 6)
        rows, per_row, ld = x.shape[0], 1, 6
    out = torch.empty(rows * per_row, 3, 3, device=x.device, dtype=torch.float32)
    hipabi.check(hipabi.lib().straps_rot6d_fwd(hipabi.ptr(x), ld, per_row, hipabi.ptr(out), rows, hipabi.stream_ptr()),
                 'straps_rot6d_fwd')
    return out


def batch_rodrigues(rot_vecs):
    """axis-angle [N,3] -> [N,3,3] with smplx's convention angle = ||r + 1e-8||."""
    hipabi.require_gpu_tensor(rot_vecs, 'axis-angle input', torch.float32)
    r = rot_vecs.detach().contiguous().view(-1, 3)
    out = torch.empty(r.shape[0], 3, 3, device=r.device, dtype=torch.float32)
    hipabi.check(hipabi.lib().straps_rodrigues_fwd(hipabi.ptr(r), hipabi.ptr(out), r.shape[0], hipabi.stream_ptr()),
                 'straps_rodrigues_fwd')
    return out
