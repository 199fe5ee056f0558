"""HomoscedasticUncertaintyWeightedMultiTaskLoss -- drop-in for reference
losses/multi_task_loss.py:7-119 (same constructor, the five `*_log_var` parameters with the same
names/order so `criterion.state_dict()` and the optimiser parameter order of run_train.py:200 match).

forward(labels, outputs) keeps the reference's dict interface; every task's (row-masked) MSE and
its gradient are HIP kernels (straps_mse_fwd / straps_mse_bwd).  `fused(...)` is the single-call
form used by the build's own train step (straps_loss_fwd_bwd: heads + 5 losses + all gradients).
"""
import numpy as np
import torch
import torch.nn as nn

from . import config, hipabi

TASKS = ('verts', 'joints2D', 'joints3D', 'shape_params', 'pose_params')       # kernel order of log-vars


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, tgt, row_mask, tscale, tshift, reduce_sum=False):
        L = hipabi.lib()
        p = pred.detach().contiguous()
        t = tgt.detach().contiguous().float()
        cols = p.shape[-1]
        rows = p.numel() // cols
        m = row_mask.contiguous().view(-1).to(torch.uint8) if row_mask is not None else None
        out = torch.empty(3, device=p.device, dtype=torch.float32)
        ws = torch.empty(512, device=p.device, dtype=torch.float32)
        hipabi.check(L.straps_mse_fwd(hipabi.ptr(p), hipabi.ptr(t), hipabi.ptr(m), rows, cols, tscale, tshift, hipabi.ptr(out), hipabi.ptr(ws),
                                      hipabi.stream_ptr()), 'straps_mse_fwd')
        ctx.saved = (p, t, m, rows, cols, tscale, tshift, out, reduce_sum)
        return out[0] if reduce_sum else out[2]          # nn.MSELoss(reduction='sum' | 'mean')

    @staticmethod
    def backward(ctx, g):
        p, t, m, rows, cols, tscale, tshift, out, reduce_sum = ctx.saved
        coef = ((g * 2.0) if reduce_sum else (g * 2.0 / out[1])).reshape(1).contiguous()
        grad = torch.empty_like(p)
        hipabi.check(hipabi.lib().straps_mse_bwd(hipabi.ptr(p), hipabi.ptr(t), hipabi.ptr(m), rows, cols, tscale, tshift, hipabi.ptr(coef),
                                                 hipabi.ptr(grad), hipabi.stream_ptr()), 'straps_mse_bwd')
        return grad, None, None, None, None, None


class HomoscedasticUncertaintyWeightedMultiTaskLoss(nn.Module):
    def __init__(self, losses_on, init_loss_weights=None, reduction='mean', eps=1e-6):
        super().__init__()
        self.losses_on = losses_on
        assert reduction in ['mean', 'sum'], "Invalid reduction for loss."
        self.reduction = reduction          # (the fused training step implements 'mean', the value run_train.py:196 uses)
        for name in ('verts', 'joints2D', 'joints3D', 'pose_params', 'shape_params'):        # registration order of the reference (:46-55)
            init = 0.0 if init_loss_weights is None else float(-np.log(init_loss_weights[name] + eps))
            setattr(self, name + '_log_var', nn.Parameter(torch.tensor(init).float(), requires_grad=name in losses_on))

    def log_var_vector(self):
        """the five log-variances in kernel order (verts, joints2D, joints3D, shape_params, pose_params)."""
        return torch.stack([getattr(self, n + '_log_var') for n in TASKS]).detach().contiguous()

    @hipabi.on_tensor_device
    def forward(self, labels, outputs):
        total_loss = 0.
        loss_dict = {}
        wh = float(config.REGRESSOR_IMG_WH)
        spec = {'verts': ('verts', 'verts'), 'joints3D': ('joints3D', 'joints3D'), 'shape_params': ('shape_params', 'shape_params'),
                'pose_params': ('pose_params_rot_matrices', 'pose_params_rot_matrices'), 'joints2D': ('joints2D', 'joints2D')}
        for name in ('verts', 'joints2D', 'joints3D', 'shape_params', 'pose_params'):         # evaluation order of the reference (:78-112)
            if name not in self.losses_on:
                continue
            ko, kl = spec[name]
            pred, lab = outputs[ko], labels[kl]
            hipabi.require_gpu_tensor(pred, "outputs['%s']" % ko, torch.float32)
            if name == 'joints2D':
                mask = labels['vis'] if 'vis' in labels else None
                mse = _MseFn.apply(pred, lab, mask, 2.0 / wh, -1.0, self.reduction == 'sum')  # label normalised 2x/wh - 1 (:92)
            else:
                mse = _MseFn.apply(pred, lab, None, 1.0, 0.0, self.reduction == 'sum')
            s = getattr(self, name + '_log_var')
            total_loss = total_loss + mse * torch.exp(-s) + s
            loss_dict[name] = mse * torch.exp(-s)
        return total_loss, loss_dict
