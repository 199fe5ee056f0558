"""Checkpoint and predict-side tooling: the `.tar` dict the reference writes every
`epochs_per_save` epochs (train/train_synthetic_otf_rendering.py:365-377) and reads back in run_predict.py:15-16,
run_train.py:204-209 and utils/checkpoint_utils.py:4-26; and the numpy proxy-representation builder of
predict/predict_3D.py:67-76."""
import copy

import numpy as np
import torch

CHECKPOINT_KEYS = ('epoch', 'best_epoch', 'best_epoch_val_metrics', 'model_state_dict', 'best_model_state_dict', 'optimiser_state_dict',
                   'criterion_state_dict')


def sync_batchnorm_buffers(model, src=0, group=None):
    """data parallel: BatchNorm running statistics (and num_batches_tracked) are per rank (every rank normalises with the statistics of
    ITS shard, the DDP convention); this broadcasts rank `src`'s into every replica.  No-op without an initialised process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return False
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.track_running_stats and m.running_mean is not None:
                for b in (m.running_mean, m.running_var, m.num_batches_tracked):
                    if b is not None:
                        dist.broadcast(b, src=src, group=group)
    return True


def save_checkpoint(path, epoch, regressor, optimiser, criterion, best_epoch=None, best_epoch_val_metrics=None, best_model_wts=None, group=None,
                    sync_bn=False):
    """`optimiser`: a torch optimiser or a train_step.TrainStep (both expose state_dict() in torch.optim.Adam's schema).
    NOT a collective by default: the usual data-parallel idiom `if rank == 0: save_checkpoint(path, ...)` works as it does with the
    reference's single-process writer (train loop :365-377).  BatchNorm running statistics are per rank under data parallel (DESIGN
    section 6); to make the file and every replica agree either call `sync_batchnorm_buffers(regressor)` on EVERY rank first, or pass
    sync_bn=True and call this function on every rank (path = None on the ranks that do not write: they only take part in the broadcast
    and build nothing)."""
    if sync_bn and sync_batchnorm_buffers(regressor, 0, group):
        enc = getattr(regressor, 'image_encoder', None)
        if enc is not None and hasattr(enc, '_bn_epoch'):
            enc._bn_epoch += 1                    # (folded-BatchNorm caches of the eval path must not outlive the new statistics)
    if path is None and sync_bn:
        return None                               # a non-writing rank of a synchronised save: no device-to-host copies, no deepcopy
    sd = {k: v.detach().cpu().clone() for k, v in regressor.state_dict().items()}
    save_dict = {'epoch': epoch,
                 'best_epoch': epoch if best_epoch is None else best_epoch,
                 'best_epoch_val_metrics': dict(best_epoch_val_metrics or {}),
                 'model_state_dict': sd,
                 'best_model_state_dict': copy.deepcopy(sd) if best_model_wts is None else best_model_wts,
                 'optimiser_state_dict': optimiser.state_dict(),
                 'criterion_state_dict': {k: v.detach().cpu().clone() for k, v in criterion.state_dict().items()}}
    if path is not None:
        torch.save(save_dict, path)
    return save_dict


def load_checkpoint(path, map_location='cpu'):
    """reference checkpoints hold numpy scalars in 'best_epoch_val_metrics', so torch >= 2.6 needs weights_only=False."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    missing = [k for k in CHECKPOINT_KEYS if k not in ck]
    if missing:
        raise KeyError('not a STRAPS checkpoint: missing %s' % missing)
    return ck


def load_training_info_from_checkpoint(checkpoint, save_val_metrics):
    """utils/checkpoint_utils.py:4-26"""
    current_epoch = checkpoint['epoch'] + 1
    best = dict(checkpoint['best_epoch_val_metrics'])
    for m in save_val_metrics:
        best.setdefault(m, np.inf)
    best = {k: v for k, v in best.items() if k in save_val_metrics}
    return current_epoch, checkpoint['best_epoch'], checkpoint['best_model_state_dict'], best


def create_proxy_representation(silhouette, joints2D, out_wh, device):
    """predict/predict_3D.py:67-76 + :125-126: [out_wh,out_wh] silhouette + [17,2(+conf)] joints -> GPU tensor
    [1,18,out_wh,out_wh]; the heat-maps are drawn by the same kernel the train step uses (int16 truncation of the joints
    like `.astype(np.int16)` there)."""
    from .label_conversions import build_proxy_input
    seg = torch.from_numpy(np.ascontiguousarray(silhouette, dtype=np.float32))[None].to(device)
    j = torch.from_numpy(np.asarray(joints2D)[:, :2].astype(np.int16).astype(np.float32))[None].to(device)
    return build_proxy_input(seg, j, out_wh)
