"""Camera helpers of the contract (reference utils/cam_utils.py).  These run under no_grad in the
reference's data generation (train loop :141) or on 17x2 values; they are thin torch expressions on
the GPU tensors they are given -- the differentiable use inside the train step (pred joints2D) is
fused into straps_loss_fwd_bwd."""
import numpy as np
import torch


def get_intrinsics_matrix(img_width, img_height, focal_length):
    """utils/cam_utils.py:29-37"""
    return np.array([[focal_length, 0., img_width / 2.0], [0., focal_length, img_height / 2.0], [0., 0., 1.]])


def orthographic_project_torch(points3D, cam_params):
    """utils/cam_utils.py:5-26: u = s(x+tx), v = s(y+ty)."""
    s, tx, ty = cam_params[:, 0:1], cam_params[:, 1:2], cam_params[:, 2:3]
    return torch.stack([s * (points3D[:, :, 0] + tx), s * (points3D[:, :, 1] + ty)], dim=-1)


def perspective_project_torch(points, rotation, translation, cam_K=None, focal_length=None, img_wh=None):
    """utils/cam_utils.py:40-71."""
    if cam_K is None:
        cam_K = torch.from_numpy(get_intrinsics_matrix(img_wh, img_wh, focal_length).astype(np.float32)).to(points.device)
        cam_K = cam_K[None].expand(points.shape[0], -1, -1)
    p = torch.einsum('bij,bkj->bki', rotation, points) + translation.unsqueeze(1)
    p = p / p[:, :, -1].unsqueeze(-1)
    return torch.einsum('bij,bkj->bki', cam_K, p)[:, :, :-1]


def check_joints2d_visibility_torch(joints2d, img_wh):
    """utils/joints2d_utils.py:23-32 (strict comparisons: 0 and img_wh count as visible)."""
    x, y = joints2d[:, :, 0], joints2d[:, :, 1]
    return ~((x > img_wh) | (y > img_wh) | (x < 0) | (y < 0))


def undo_keypoint_normalisation(normalised_keypoints, img_wh):
    """utils/joints2d_utils.py:5-10"""
    return (normalised_keypoints + 1) * (img_wh / 2.0)
