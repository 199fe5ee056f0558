"""How much of the stem's gradient tensor does the stem weight gradient read?  stem_wgrad_kernel visits a 2-row x 32-column output tile for
a group of 4 input channels only if the group has a non-zero inside the tile's 9 x 72 input patch; the tile of dy is needed if ANY group is
active.  Prints, for the training step's own batches (B = 64), the fraction of (tile, group) pairs and of tiles that are active -- the share
of the fused stem tail's output (straps_bn_bwd_pooled: 268 MB per step) that is ever read."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import straps_amd  # noqa: E402
from straps_amd.train_step import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
mp = straps_amd.synthetic_mean_params(0)
model = straps_amd.synthetic_smpl_model(0)
B = 64
reg = straps_amd.SingleInputRegressor(18, 18, 3, mean_params=mp).to(dev).train()
smpl = straps_amd.SMPL(model, batch_size=B).to(dev)
crit = straps_amd.HomoscedasticUncertaintyWeightedMultiTaskLoss(
    ['verts', 'shape_params', 'pose_params', 'joints2D', 'joints3D'],
    init_loss_weights={'verts': 1.0, 'joints2D': 0.1, 'pose_params': 0.1, 'shape_params': 0.1, 'joints3D': 1.0}).to(dev)
ts = TrainStep(reg, smpl, crit, B, lr=1e-4, mean_shape=mp['shape'], pipeline_data=False)
for it in range(3):
    with torch.no_grad():
        x = ts.make_batch()['input']                      # [B,18,256,256]
    nz = (x != 0).float()
    nzp = F.pad(nz, (3, 5, 3, 5))                         # patch of tile (ty, tx): rows 4ty-3 .. 4ty+5, cols 64tx-3 .. 64tx+68
    act = F.max_pool2d(nzp, kernel_size=(9, 72), stride=(4, 64))          # [B,18,64,4] per channel
    assert act.shape[-2:] == (64, 4), act.shape
    grp = torch.stack([act[:, g:g + 4].amax(1) for g in range(0, 18, 4)], 1)   # [B,5,64,4]
    anyg = grp.amax(1)
    print('batch %d: non-zero input %.4f | active (tile, group) pairs %.3f | tiles with any active group %.3f'
          % (it, float(nz.mean()), float(grp.mean()), float(anyg.mean())))
