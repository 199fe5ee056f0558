"""Augmentation entry points with the reference's names and argument meaning
(augmentation/smpl_augmentation.py, cam_augmentation.py, proxy_rep_augmentation.py), on GPU tensors.
Random numbers come from torch's device generator (the reference mixes device torch RNG and host
numpy RNG, train loop :121-175); the arithmetic on them runs in HIP kernels where it touches images
(`straps_augment_seg`) or rotations (`straps_rodrigues_fwd`).  `train_step.TrainStep.make_batch` is
the fused form of the same sequence."""
import torch

from . import hipabi
from .rigid_transform_utils import batch_rodrigues


def uniform_sample_shape(batch_size, mean_shape, delta_betas_range):
    """smpl_augmentation.py:6-15"""
    l, h = delta_betas_range
    return (h - l) * torch.rand(batch_size, 10, device=mean_shape.device) + l + mean_shape


def normal_sample_shape(batch_size, mean_shape, std_vector):
    """smpl_augmentation.py:18-25"""
    return torch.randn(batch_size, 10, device=mean_shape.device) * std_vector + mean_shape


def augment_smpl(orig_shape, pose, global_orients, mean_shape, smpl_augment_params):
    """smpl_augmentation.py:27-61: resample betas around the mean shape, axis-angle -> rotation matrices.
    Returns (shape [B,10], pose_rotmats [B,23,3,3], glob_rotmats [B,1,3,3])."""
    B = orig_shape.shape[0]
    if smpl_augment_params['augment_shape']:
        dist = smpl_augment_params['delta_betas_distribution']
        assert dist in ['uniform', 'normal']
        if dist == 'uniform':
            new_shape = uniform_sample_shape(B, mean_shape, smpl_augment_params['delta_betas_range'])
        else:
            assert smpl_augment_params['delta_betas_std_vector'] is not None
            new_shape = normal_sample_shape(B, mean_shape, smpl_augment_params['delta_betas_std_vector'])
    else:
        new_shape = orig_shape
    pose_rotmats = batch_rodrigues(pose.contiguous().view(-1, 3)).view(-1, 23, 3, 3)
    glob_rotmats = batch_rodrigues(global_orients.contiguous().view(-1, 3)).unsqueeze(1)
    return new_shape, pose_rotmats, glob_rotmats


def augment_cam_t(mean_cam_t, xy_std=0.05, delta_z_range=(-5, 5)):
    """cam_augmentation.py:4-14"""
    B, dev = mean_cam_t.shape[0], mean_cam_t.device
    new_cam_t = mean_cam_t.clone()
    new_cam_t[:, :2] = mean_cam_t[:, :2] + torch.randn(B, 2, device=dev) * xy_std
    l, h = delta_z_range
    new_cam_t[:, 2] = mean_cam_t[:, 2] + (h - l) * torch.rand(B, device=dev) + l
    return new_cam_t


def random_verts2D_deviation(vertices, delta_verts2d_dev_range=(-0.01, 0.01)):
    """proxy_rep_augmentation.py:5-22"""
    l, h = delta_verts2d_dev_range
    noisy = vertices.clone()
    noisy[:, :, :2] += (h - l) * torch.rand(vertices.shape[0], vertices.shape[1], 2, device=vertices.device) + l
    return noisy


def random_joints2D_deviation(joints2D, delta_j2d_dev_range=(-5, 5), delta_j2d_hip_dev_range=(-15, 15)):
    """proxy_rep_augmentation.py:25-49 (in place on its argument, like the reference)."""
    hip, other = [11, 12], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16]
    B, dev = joints2D.shape[0], joints2D.device
    l, h = delta_j2d_dev_range
    joints2D[:, other, :] = joints2D[:, other, :] + (h - l) * torch.rand(B, len(other), 2, device=dev) + l
    l, h = delta_j2d_hip_dev_range
    joints2D[:, hip, :] = joints2D[:, hip, :] + (h - l) * torch.rand(B, len(hip), 2, device=dev) + l
    return joints2D


def augment_proxy_representation(orig_segs, orig_joints2D, proxy_rep_augment_params):
    """proxy_rep_augmentation.py:104-123: body-part removal + box occlusion of the part segmentation (one HIP kernel,
    per-sample decisions from device uniforms) and joint jitter.  Inputs are not modified."""
    hipabi.require_gpu_tensor(orig_segs, 'segmentation', torch.float32)
    p = proxy_rep_augment_params
    B, wh = orig_segs.shape[0], orig_segs.shape[-1]
    new_joints2D = orig_joints2D.clone()
    probs = torch.zeros(6, device=orig_segs.device)
    if p['remove_appendages']:
        for c, pr in zip(p['remove_appendages_classes'], p['remove_appendages_probabilities']):
            probs[c - 1] = pr
    occl = p['occlude_probability'] if p['occlude_seg'] else 0.0
    u = torch.rand(B, 9, device=orig_segs.device)
    new_segs = torch.empty_like(orig_segs, memory_format=torch.contiguous_format)
    hipabi.check(hipabi.lib().straps_augment_seg(hipabi.ptr(orig_segs.contiguous()), hipabi.ptr(u), hipabi.ptr(probs), float(occl),
                                                 int(p['occlude_box_dim']), hipabi.ptr(new_segs), B, wh, hipabi.stream_ptr()),
                 'straps_augment_seg')
    if p['deviate_joints2D']:
        new_joints2D = random_joints2D_deviation(new_joints2D, p['delta_j2d_dev_range'], p['delta_j2d_hip_dev_range'])
    return new_segs, new_joints2D
