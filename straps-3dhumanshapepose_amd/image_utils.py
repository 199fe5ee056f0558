"""On-device crop / resize of the part segmentation: the GPU counterpart of
`batch_crop_seg_to_bounding_box` + `batch_resize` (reference utils/image_utils.py:44-105), which the reference
runs on the host between two device<->host copies every training batch (train loop :161-170)."""
import torch

from . import hipabi


def batch_crop_and_resize(seg, joints2D, img_wh, orig_scale_factor=1.2, delta_scale_range=None, delta_centre_range=None,
                          uniforms=None, out=None, jout=None):
    """seg [B,wh,wh] part ids, joints2D [B,J,2] (GPU fp32) -> (resized seg [B,img_wh,img_wh], resized joints, boxes).
    Random scale / centre jitter is applied when both ranges are given; `uniforms` [B,3] in [0,1) may be supplied
    (otherwise drawn from the module-level device generator, `device_rng.manual_seed`)."""
    hipabi.require_gpu_tensor(seg, 'segmentation', torch.float32)
    hipabi.require_gpu_tensor(joints2D, 'joints2D', torch.float32)
    B, wh, nj = seg.shape[0], seg.shape[-1], joints2D.shape[1]
    jitter = delta_scale_range is not None and delta_centre_range is not None
    if jitter and uniforms is None:
        from . import device_rng
        g = device_rng.default_draws(seg.device)
        uniforms = g.uniform(B, 3)
        g.advance()
    ds, dc = (delta_scale_range or (0.0, 0.0)), (delta_centre_range or (0.0, 0.0))
    out = torch.empty(B, img_wh, img_wh, device=seg.device, dtype=torch.float32) if out is None else out
    jout = torch.empty(B, nj, 2, device=seg.device, dtype=torch.float32) if jout is None else jout
    boxes = torch.empty(B, 6, device=seg.device, dtype=torch.int32)
    u = uniforms.contiguous().float() if jitter else None
    if u is not None and u.numel() != B * 3:
        raise RuntimeError('batch_crop_and_resize: uniforms must hold [B,3] draws')
    hipabi.check(hipabi.lib().straps_crop_resize(hipabi.ptr(seg.contiguous()), hipabi.ptr(joints2D.contiguous()), hipabi.ptr(u),
                                                 float(orig_scale_factor), float(ds[0]), float(ds[1]), float(dc[0]), float(dc[1]),
                                                 hipabi.ptr(out), hipabi.ptr(jout), hipabi.ptr(boxes), B, wh, img_wh, nj,
                                                 hipabi.stream_ptr()), 'straps_crop_resize')
    return out, jout, boxes
