"""Network-input construction on the GPU (reference utils/label_conversions.py:48-55, 90-127)."""
import torch

from . import hipabi


def build_proxy_input(seg, joints2d, img_wh=256, std=4):
    """train loop :178-182 in one kernel: seg [B,wh,wh] part ids + joints2d [B,17,2] ->
    [B,18,wh,wh] = (seg != 0) ++ 17 Gaussian heat-maps (standard deviation `std` pixels, truncated at 2 std)."""
    hipabi.require_gpu_tensor(seg, 'segmentation', torch.float32)
    hipabi.require_gpu_tensor(joints2d, 'joints2D', torch.float32)
    B, nj = joints2d.shape[0], joints2d.shape[1]
    out = torch.empty(B, nj + 1, img_wh, img_wh, device=seg.device, dtype=torch.float32)
    if int(std) != std or std < 1:
        raise ValueError('heat-map std must be a positive integer (the reference builds a 4 std x 4 std grid with torch.linspace(-2 std, 2 std, 4 std))')
    hipabi.check(hipabi.lib().straps_build_proxy_input_std(hipabi.ptr(seg.contiguous()), hipabi.ptr(joints2d.contiguous()), hipabi.ptr(out), B, nj,
                                                           img_wh, int(std), hipabi.stream_ptr()), 'straps_build_proxy_input_std')
    return out


def convert_2Djoints_to_gaussian_heatmaps_torch(joints2D, img_wh, std=4):
    """utils/label_conversions.py:90-127 (every call site of the reference uses std = 4)."""
    seg = torch.empty(joints2D.shape[0], img_wh, img_wh, device=joints2D.device, dtype=torch.float32)
    hipabi.check(hipabi.lib().straps_memset_zero(hipabi.ptr(seg), seg.numel() * 4, hipabi.stream_ptr()), 'straps_memset_zero')
    return build_proxy_input(seg, joints2D.float(), img_wh, std)[:, 1:]


def convert_multiclass_to_binary_labels_torch(multiclass_labels):
    """utils/label_conversions.py:48-55"""
    B, wh = multiclass_labels.shape[0], multiclass_labels.shape[-1]
    j = torch.full((B, 1, 2), -1000.0, device=multiclass_labels.device)
    return build_proxy_input(multiclass_labels.float(), j, wh)[:, 0]
