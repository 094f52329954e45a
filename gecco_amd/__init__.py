"""gecco_amd — MI355X-native GECCO denoiser (drop-in for the `gecco_torch` module API).

    from gecco_amd.diffusion import EDMPrecond, Diffusion, IdleConditioner, LogUniformSchedule, EDMLoss
    from gecco_amd.models.set_transformer import SetTransformer
    from gecco_amd.models.linear_lift import LinearLift
    from gecco_amd.models.activation import GaussianActivation
    from gecco_amd.reparam import GaussianReparam

The compute path is the hand-written HIP library `libgecco_hip.so` (C ABI: include/gecco_hip.h).  There is no CPU or
eager-PyTorch fallback: if the library is missing or the tensors are not on a HIP device, the operators raise.
"""
__version__ = "0.1.0"

from . import models, reparam  # noqa: E402,F401
from .config import load_config  # noqa: E402,F401
from .diffusion import Diffusion  # noqa: E402,F401
from .hip_ops import frozen_weights, weights_changed  # noqa: E402,F401
