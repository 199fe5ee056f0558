"""SingleInputRegressor -- drop-in for reference models/regressor.py:7-47 (same constructor
arguments, attribute names `image_encoder` / `ief_module`, 132/330 state-dict keys)."""
import torch
import torch.nn as nn

from . import hipabi
from .ief_module import IEFModule
from .resnet import resnet18, resnet50


class SingleInputRegressor(nn.Module):
    def __init__(self, resnet_in_channels=1, resnet_layers=18, ief_iters=3, mean_params=None):
        """`mean_params` (optional, extension): dict/npz with 'pose'[144], 'shape'[10] instead of reading
        config.SMPL_MEAN_PARAMS_PATH from the working directory."""
        super().__init__()
        num_output_params = 3 + 24 * 6 + 10
        # like the reference, other depths construct nothing (models/regressor.py:28-41) and fail at forward
        if resnet_layers == 18:
            self.image_encoder = resnet18(in_channels=resnet_in_channels, pretrained=False)
            self.ief_module = IEFModule([512, 512], 512, num_output_params, iterations=ief_iters, mean_params=mean_params)
        elif resnet_layers == 50:
            self.image_encoder = resnet50(in_channels=resnet_in_channels, pretrained=False)
            self.ief_module = IEFModule([1024, 1024], 2048, num_output_params, iterations=ief_iters, mean_params=mean_params)

    @hipabi.on_tensor_device
    def forward(self, input):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd_ops import regressor_autograd
            return regressor_autograd(self, input)
        feats = self.image_encoder(input)
        return self.ief_module(feats)
