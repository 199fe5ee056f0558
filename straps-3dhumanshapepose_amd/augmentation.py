"""Augmentation entry points with the reference's names and argument meaning
(augmentation/smpl_augmentation.py, cam_augmentation.py, proxy_rep_augmentation.py), on GPU tensors.

Every function is a thin host wrapper over ONE C-ABI call (csrc/augment.hip, train.hip) -- the same calls
`train_step.TrainStep.make_batch` makes -- and draws its random numbers from the device generator of `device_rng`
(Philox, `device_rng.manual_seed`).  The reference mixes torch's device generator with numpy's host generator
(train loop :121-175); streams cannot match across generators, so each function also accepts its draws explicitly
(`normals=` / `uniforms=`): with the draws supplied the results equal the oracle's restatement of the reference
arithmetic bit for bit (tests/test_gpu_augment.py)."""
import torch

from . import device_rng, hipabi

HIP_JOINTS = (11, 12)          # proxy_rep_augmentation.py:37


def _draw(kind, shape, device, given, substream):
    if given is not None:
        hipabi.require_gpu_tensor(given, 'supplied draws', torch.float32)
        n = 1
        for s in shape:
            n *= s
        if given.numel() != n:
            raise RuntimeError('supplied draws hold %d values, expected %s' % (given.numel(), tuple(shape)))
        return given.contiguous()
    g = device_rng.default_draws(device)
    out = g.fill(torch.empty(*shape, device=device, dtype=torch.float32), kind, substream)
    g.advance()
    return out


def uniform_sample_shape(batch_size, mean_shape, delta_betas_range, uniforms=None):
    """smpl_augmentation.py:6-15: mean_shape + U[l, h)."""
    return _sample_shape(batch_size, mean_shape, 2, None, delta_betas_range, uniforms)


def normal_sample_shape(batch_size, mean_shape, std_vector, normals=None):
    """smpl_augmentation.py:18-25: mean_shape + N(0,1) * std_vector."""
    return _sample_shape(batch_size, mean_shape, 1, std_vector, (0.0, 0.0), normals)


def _sample_shape(B, mean_shape, mode, std_vector, rng, draws):
    hipabi.require_gpu_tensor(mean_shape, 'mean_shape', torch.float32)
    dev = mean_shape.device
    draws = _draw(device_rng.NORMAL if mode == 1 else device_rng.UNIFORM, (B, 10), dev, draws, 1 if mode == 1 else 0)
    std = None
    if mode == 1:
        std = torch.as_tensor(std_vector, dtype=torch.float32, device=dev).expand(10).contiguous()
    shape = torch.empty(B, 10, device=dev)
    # shape-only call of the augment_smpl kernel: a zero pose row per body feeds the (discarded) rotation half
    zeros = torch.zeros(B, 72, device=dev)
    rot = torch.empty(B, 24, 3, 3, device=dev)
    hipabi.check(hipabi.lib().straps_augment_smpl(hipabi.ptr(zeros), B, None, None, hipabi.ptr(mean_shape.contiguous()), hipabi.ptr(draws), mode,
                                                  hipabi.ptr(std), float(rng[0]), float(rng[1]), hipabi.ptr(shape), hipabi.ptr(rot), None, B,
                                                  hipabi.stream_ptr()), 'straps_augment_smpl')
    return shape


def _pose_rows(pose, global_orients):
    """[B,72] axis-angle rows (global orientation first).  The reference passes the two slices target_pose[:, 3:] and
    target_pose[:, :3] of one tensor (train loop :121-126): recognised and used in place, anything else is concatenated."""
    B = pose.shape[0]
    if (pose.dim() == 2 and global_orients.dim() == 2 and pose.shape[1] == 69 and global_orients.shape[1] == 3
            and pose.stride() == (72, 1) and global_orients.stride() == (72, 1)
            and pose.data_ptr() == global_orients.data_ptr() + 12):
        return global_orients.as_strided((B, 72), (72, 1))
    return torch.cat([global_orients.reshape(B, 3), pose.reshape(B, 69)], dim=1).contiguous()


@hipabi.on_tensor_device
def augment_smpl(orig_shape, pose, global_orients, mean_shape, smpl_augment_params, shape_draws=None):
    """smpl_augmentation.py:27-61: resample betas around the mean shape, axis-angle -> rotation matrices.
    Returns (shape [B,10], pose_rotmats [B,23,3,3], glob_rotmats [B,1,3,3]).  shape_draws: optional [B,10] N(0,1)
    ('normal') or U[0,1) ('uniform') draws."""
    hipabi.require_gpu_tensor(pose, 'pose', torch.float32)
    hipabi.require_gpu_tensor(global_orients, 'global_orients', torch.float32)
    B, dev = pose.shape[0], pose.device
    p = smpl_augment_params
    mode, std, rng, draws, mean = 0, None, (0.0, 0.0), None, None
    if p['augment_shape']:
        dist = p['delta_betas_distribution']
        assert dist in ['uniform', 'normal']
        hipabi.require_gpu_tensor(mean_shape, 'mean_shape', torch.float32)
        mean = mean_shape.contiguous()
        if dist == 'uniform':
            mode, rng = 2, p['delta_betas_range']
            draws = _draw(device_rng.UNIFORM, (B, 10), dev, shape_draws, 0)
        else:
            assert p['delta_betas_std_vector'] is not None
            mode = 1
            std = torch.as_tensor(p['delta_betas_std_vector'], dtype=torch.float32, device=dev).expand(10).contiguous()
            draws = _draw(device_rng.NORMAL, (B, 10), dev, shape_draws, 1)
    else:
        hipabi.require_gpu_tensor(orig_shape, 'orig_shape', torch.float32)
    rows = _pose_rows(pose, global_orients)
    new_shape = torch.empty(B, 10, device=dev)
    rot = torch.empty(B, 24, 3, 3, device=dev)
    hipabi.check(hipabi.lib().straps_augment_smpl(hipabi.ptr(rows), B, None, hipabi.ptr(orig_shape.contiguous() if mode == 0 else None),
                                                  hipabi.ptr(mean), hipabi.ptr(draws), mode, hipabi.ptr(std), float(rng[0]), float(rng[1]),
                                                  hipabi.ptr(new_shape), hipabi.ptr(rot), None, B, hipabi.stream_ptr()), 'straps_augment_smpl')
    return new_shape, rot[:, 1:], rot[:, :1]


@hipabi.on_tensor_device
def augment_cam_t(mean_cam_t, xy_std=0.05, delta_z_range=(-5, 5), normals_xy=None, uniform_z=None):
    """cam_augmentation.py:4-14"""
    hipabi.require_gpu_tensor(mean_cam_t, 'mean_cam_t', torch.float32)
    B, dev = mean_cam_t.shape[0], mean_cam_t.device
    n = _draw(device_rng.NORMAL, (B, 2), dev, normals_xy, 1)
    u = _draw(device_rng.UNIFORM, (B,), dev, uniform_z, 0)
    out = torch.empty(B, 3, device=dev)
    hipabi.check(hipabi.lib().straps_augment_cam_t(hipabi.ptr(mean_cam_t.contiguous()), hipabi.ptr(n), hipabi.ptr(u), float(xy_std),
                                                   float(delta_z_range[0]), float(delta_z_range[1]), hipabi.ptr(out), B, hipabi.stream_ptr()),
                 'straps_augment_cam_t')
    return out


@hipabi.on_tensor_device
def random_verts2D_deviation(vertices, delta_verts2d_dev_range=(-0.01, 0.01), uniforms=None):
    """proxy_rep_augmentation.py:5-22 (a materialised noisy copy; the training step applies the same noise inside the
    rasteriser instead, NMRRenderer.render_arrays(vert_noise_u=...))."""
    hipabi.require_gpu_tensor(vertices, 'vertices', torch.float32)
    B, N = vertices.shape[0], vertices.shape[1]
    u = _draw(device_rng.UNIFORM, (B, N, 2), vertices.device, uniforms, 0)
    v = vertices.contiguous()
    out = torch.empty_like(v)
    l, h = delta_verts2d_dev_range
    hipabi.check(hipabi.lib().straps_deviate_verts2d(hipabi.ptr(v), hipabi.ptr(u), float(l), float(h), hipabi.ptr(out), B * N,
                                                     hipabi.stream_ptr()), 'straps_deviate_verts2d')
    return out


@hipabi.on_tensor_device
def random_joints2D_deviation(joints2D, delta_j2d_dev_range=(-5, 5), delta_j2d_hip_dev_range=(-15, 15), uniforms=None):
    """proxy_rep_augmentation.py:25-49 (in place on its argument, like the reference)."""
    hipabi.require_gpu_tensor(joints2D, 'joints2D', torch.float32)
    if joints2D.shape[1:] != (17, 2) or not joints2D.is_contiguous():
        raise RuntimeError('random_joints2D_deviation: expected contiguous [B,17,2] COCO joints, got %s' % (tuple(joints2D.shape),))
    B = joints2D.shape[0]
    u = _draw(device_rng.UNIFORM, (B, 17, 2), joints2D.device, uniforms, 0)
    l, h = delta_j2d_dev_range
    hl, hh = delta_j2d_hip_dev_range
    hipabi.check(hipabi.lib().straps_deviate_joints2d(hipabi.ptr(joints2D), hipabi.ptr(u), float(l), float(h), float(hl), float(hh),
                                                      hipabi.ptr(joints2D), B, hipabi.stream_ptr()), 'straps_deviate_joints2d')
    return joints2D


def remove_probabilities(proxy_rep_augment_params, device):
    """per-class removal probability vector [6] (classes 1..6) of random_remove_bodyparts (:52-75)."""
    p = proxy_rep_augment_params
    probs = [0.0] * 6
    if p['remove_appendages']:
        for c, pr in zip(p['remove_appendages_classes'], p['remove_appendages_probabilities']):
            probs[int(c) - 1] = float(pr)
    return torch.tensor(probs, dtype=torch.float32, device=device)


@hipabi.on_tensor_device
def augment_proxy_representation(orig_segs, orig_joints2D, proxy_rep_augment_params, seg_uniforms=None, joint_uniforms=None):
    """proxy_rep_augmentation.py:104-123: body-part removal + box occlusion of the part segmentation (one HIP kernel,
    per-sample decisions from uniforms [B,9]: 6 removal draws, 1 occlusion draw, 2 box-centre draws) and joint jitter.
    Inputs are not modified."""
    hipabi.require_gpu_tensor(orig_segs, 'segmentation', torch.float32)
    p = proxy_rep_augment_params
    B, wh = orig_segs.shape[0], orig_segs.shape[-1]
    probs = remove_probabilities(p, orig_segs.device)
    occl = p['occlude_probability'] if p['occlude_seg'] else 0.0
    u = _draw(device_rng.UNIFORM, (B, 9), orig_segs.device, seg_uniforms, 0)
    new_segs = torch.empty_like(orig_segs, memory_format=torch.contiguous_format)
    hipabi.check(hipabi.lib().straps_augment_seg(hipabi.ptr(orig_segs.contiguous()), hipabi.ptr(u), hipabi.ptr(probs), float(occl),
                                                 int(p['occlude_box_dim']), hipabi.ptr(new_segs), B, wh, hipabi.stream_ptr()),
                 'straps_augment_seg')
    new_joints2D = orig_joints2D.clone()
    if p['deviate_joints2D']:
        new_joints2D = random_joints2D_deviation(new_joints2D.contiguous(), p['delta_j2d_dev_range'], p['delta_j2d_hip_dev_range'],
                                                 uniforms=joint_uniforms)
    return new_segs, new_joints2D
