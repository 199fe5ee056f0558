"""MI355X-native STRAPS hot path: ResNet encoder -> IEF regressor -> rot6d -> SMPL forward.

Host side mirrors the reference's nn.Module surface (names, signatures, state-dict keys); all
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in include/straps_hip.h
(loaded with ctypes, see hipabi.py).  There is NO CPU fallback: calling a module without the HIP
library / a GPU raises.
"""
from . import config, hipabi  # noqa: F401
from .synthetic_smpl import synthetic_smpl_model, synthetic_mean_params, load_smpl_model  # noqa: F401
from .resnet import ResNet, BasicBlock, Bottleneck, resnet18, resnet50  # noqa: F401
from .ief_module import IEFModule  # noqa: F401
from .regressor import SingleInputRegressor  # noqa: F401
from .smpl import SMPL, ModelOutput, pack_smpl_model  # noqa: F401
from .rigid_transform_utils import rot6d_to_rotmat, batch_rodrigues  # noqa: F401
from .multi_task_loss import HomoscedasticUncertaintyWeightedMultiTaskLoss  # noqa: F401
from .nmr_renderer import NMRRenderer  # noqa: F401
from . import cam_utils, label_conversions, augmentation, metrics, checkpoint_utils, image_utils, device_rng  # noqa: F401
