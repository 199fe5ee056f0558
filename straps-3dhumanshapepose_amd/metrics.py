"""On-device evaluation metrics: the quantities the reference's
TrainingLossesAndMetricsTracker.update_per_batch adds up per batch
(metrics/train_loss_and_metrics_tracker.py:127-213) without copying vertices to the host or running a
per-sample numpy SVD (utils/eval_utils.py:58-63)."""
import torch

from . import config, hipabi


def point_error_sums(pred, target):
    """pred, target [B,N,3] GPU fp32 -> [B,3] per-sample sums of (raw, scale+translation-corrected,
    Procrustes-aligned) point-wise L2 errors."""
    hipabi.require_gpu_tensor(pred, 'pred points', torch.float32)
    hipabi.require_gpu_tensor(target, 'target points', torch.float32)
    assert pred.shape == target.shape and pred.dim() == 3 and pred.shape[2] == 3
    p, t = pred.detach().contiguous(), target.detach().contiguous()
    out = torch.empty(p.shape[0], 3, device=p.device, dtype=torch.float32)
    hipabi.check(hipabi.lib().straps_point_metrics(hipabi.ptr(p), hipabi.ptr(t), hipabi.ptr(out), p.shape[0], p.shape[1], hipabi.stream_ptr()),
                 'straps_point_metrics')
    return out


class BatchMetrics:
    """running sums with the tracker's key names; `update` takes the same dicts as the reference's
    update_per_batch (pred_dict / target_dict with 'verts', 'joints3D', 'joints2D', 'shape_params',
    'pose_params_rot_matrices') and stays on the device -- call `summary()` once per epoch."""

    KEYS = ('pves', 'pves_sc', 'pves_pa', 'pve-ts', 'pve-ts_sc', 'mpjpes', 'mpjpes_sc', 'mpjpes_pa', 'shape_mses', 'pose_mses',
            'joints2D_l2es')

    def __init__(self, device, img_wh=config.REGRESSOR_IMG_WH):
        self.sums = torch.zeros(len(self.KEYS), device=device, dtype=torch.float64)
        self.n = 0
        self.img_wh = img_wh

    def update(self, pred_dict, target_dict, pred_reposed_vertices=None, target_reposed_vertices=None):
        v = point_error_sums(pred_dict['verts'], target_dict['verts']).double().sum(0)
        j = point_error_sums(pred_dict['joints3D'], target_dict['joints3D']).double().sum(0)
        add = torch.zeros_like(self.sums)
        add[0:3] = v
        add[5:8] = j
        if pred_reposed_vertices is not None:
            add[3:5] = point_error_sums(pred_reposed_vertices, target_reposed_vertices).double().sum(0)[:2]
        add[8] = ((pred_dict['shape_params'] - target_dict['shape_params']).double() ** 2).sum()
        add[9] = ((pred_dict['pose_params_rot_matrices'] - target_dict['pose_params_rot_matrices']).double() ** 2).sum()
        p2 = (pred_dict['joints2D'] + 1) * (self.img_wh / 2.0)          # undo_keypoint_normalisation (utils/joints2d_utils.py:5-10)
        add[10] = (p2 - target_dict['joints2D']).double().norm(dim=-1).sum()
        self.sums += add
        self.n += pred_dict['verts'].shape[0]

    def summary(self):
        """per-epoch means with the tracker's normalisers (update_per_epoch :215-251): per vertex (6890), per joint
        (14 / 17), per sample for the parameter MSE sums."""
        s = self.sums.cpu().numpy()
        n = max(self.n, 1)
        per = {'pves': 6890, 'pves_sc': 6890, 'pves_pa': 6890, 'pve-ts': 6890, 'pve-ts_sc': 6890, 'mpjpes': 14, 'mpjpes_sc': 14,
               'mpjpes_pa': 14, 'shape_mses': 10, 'pose_mses': 24 * 9, 'joints2D_l2es': 17}
        return {k: float(s[i]) / (n * per[k]) for i, k in enumerate(self.KEYS)}
