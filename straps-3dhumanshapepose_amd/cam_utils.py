"""Camera helpers of the contract (reference utils/cam_utils.py), on HIP kernels behind the C ABI (csrc/pose.hip):
`orthographic_project_torch` is differentiable (autograd Function, forward + backward kernels) like the reference's torch expression --
`predict/predict_3D.py:144` projects the 6890 predicted vertices through it; inside the fused training step the same projection is part
of straps_loss_fwd_bwd.  `perspective_project_torch` runs in the reference's data generation under no_grad (train loop :141)."""
import numpy as np
import torch

from . import hipabi


def get_intrinsics_matrix(img_width, img_height, focal_length):
    """utils/cam_utils.py:29-37"""
    return np.array([[focal_length, 0., img_width / 2.0], [0., focal_length, img_height / 2.0], [0., 0., 1.]])


class _OrthoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points3D, cam_params):
        p = points3D.detach().float().contiguous()
        c = cam_params.detach().float()
        if c.dim() != 2 or c.stride(1) != 1:
            c = c.contiguous()
        B, N = p.shape[0], p.shape[1]
        out = torch.empty(B, N, 2, device=p.device, dtype=torch.float32)
        hipabi.check(hipabi.lib().straps_orthographic_project(hipabi.ptr(p), hipabi.ptr(c), c.stride(0), hipabi.ptr(out), B, N, hipabi.stream_ptr()),
                     'straps_orthographic_project')
        ctx.save_for_backward(p, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        p, c = ctx.saved_tensors
        B, N = p.shape[0], p.shape[1]
        need_p, need_c = ctx.needs_input_grad
        dp = torch.empty_like(p) if need_p else None
        dc = torch.empty(B, 3, device=p.device, dtype=torch.float32)
        hipabi.check(hipabi.lib().straps_orthographic_project_bwd(hipabi.ptr(p), hipabi.ptr(c), c.stride(0), hipabi.ptr(dout.float().contiguous()), hipabi.ptr(dp),
                                                                  hipabi.ptr(dc), B, N, hipabi.stream_ptr()), 'straps_orthographic_project_bwd')
        return dp, (dc if need_c else None)


@hipabi.on_tensor_device
def orthographic_project_torch(points3D, cam_params):
    """utils/cam_utils.py:5-26: u = s(x+tx), v = s(y+ty); points3D [B,N,3], cam_params [B,3] (any row stride) -> [B,N,2]."""
    hipabi.require_gpu_tensor(points3D, 'points3D')
    hipabi.require_gpu_tensor(cam_params, 'cam_params')
    if points3D.dim() != 3 or points3D.shape[-1] != 3 or cam_params.shape != (points3D.shape[0], 3):
        raise RuntimeError('orthographic_project_torch expects points [B,N,3] and cam_params [B,3], got %s and %s' % (tuple(points3D.shape), tuple(cam_params.shape)))
    return _OrthoFn.apply(points3D, cam_params)


@hipabi.on_tensor_device
def perspective_project_torch(points, rotation, translation, cam_K=None, focal_length=None, img_wh=None):
    """utils/cam_utils.py:40-71: p = R x + t; p /= p_z; (K p)[:2].  points [B,N,3], rotation [B,3,3], translation [B,3], cam_K [B,3,3]
    or focal_length + img_wh.  No gradient (the reference calls it under no_grad, train loop :141)."""
    hipabi.require_gpu_tensor(points, 'points')
    if any(t.requires_grad for t in (points, rotation, translation)) and torch.is_grad_enabled():
        raise NotImplementedError('perspective_project_torch: gradients are not implemented (the reference path calls it under no_grad)')
    B, N = points.shape[0], points.shape[1]
    dev = points.device
    if cam_K is None:
        cam_K = torch.from_numpy(get_intrinsics_matrix(img_wh, img_wh, focal_length).astype(np.float32)).to(dev)
    cam_K = cam_K.detach().float().to(dev).contiguous()
    per_body = int(cam_K.dim() == 3)
    if per_body and cam_K.shape[0] != B:
        raise RuntimeError('perspective_project_torch: cam_K batch %d != points batch %d' % (cam_K.shape[0], B))
    p = points.detach().float().contiguous()
    R = rotation.detach().float().to(dev).contiguous()
    t = translation.detach().float().to(dev).contiguous()
    out = torch.empty(B, N, 2, device=dev, dtype=torch.float32)
    hipabi.check(hipabi.lib().straps_perspective_project(hipabi.ptr(p), hipabi.ptr(R), hipabi.ptr(t), hipabi.ptr(cam_K), per_body, hipabi.ptr(out), B, N,
                                                         hipabi.stream_ptr()), 'straps_perspective_project')
    return out


def check_joints2d_visibility_torch(joints2d, img_wh):
    """utils/joints2d_utils.py:23-32 (strict comparisons: 0 and img_wh count as visible)."""
    x, y = joints2d[:, :, 0], joints2d[:, :, 1]
    return ~((x > img_wh) | (y > img_wh) | (x < 0) | (y < 0))


def undo_keypoint_normalisation(normalised_keypoints, img_wh):
    """utils/joints2d_utils.py:5-10"""
    return (normalised_keypoints + 1) * (img_wh / 2.0)
