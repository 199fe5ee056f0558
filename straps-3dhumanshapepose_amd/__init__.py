"""MI355X-native STRAPS hot path: ResNet encoder -> IEF regressor -> rot6d -> SMPL forward.

Host side mirrors the reference's nn.Module surface (names, signatures, state-dict keys); all
arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in include/straps_hip.h
(loaded with ctypes, see hipabi.py).  There is NO CPU fallback: calling a module without the HIP
library / a GPU raises.
"""
from . import config  # noqa: F401
from .synthetic_smpl import synthetic_smpl_model, synthetic_mean_params, load_smpl_model  # noqa: F401
