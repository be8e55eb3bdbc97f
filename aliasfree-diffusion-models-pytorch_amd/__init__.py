"""afdm -- MI355X-native alias-free DDPM engine (hot path of MDFahimAnjum/AliasFree-Diffusion-Models-PyTorch).

Public surface = the reference's Python API for the hot path:
UNet, Diffusion, train, custom_upsample, custom_downsample, circularLowpassKernel, argument,
set_seed, plus the engine-side helpers TrainStep / FusedAdamW / GradAllReduce.
Device work runs in libafd_hip.so (hand-written gfx950 kernels, C ABI in include/afd.h).
"""
from ._lib import AfdError, lib  # noqa: F401
from .filters import circularLowpassKernel, custom_downsample, custom_upsample  # noqa: F401
from .blocks import (SelfAttention, DoubleConv, DoubleConv_F, DoubleConv_F4, Down, Down_F, Down_FF, Down_FFF,  # noqa: F401
                     Down_F4, Up, Up_F, Up_FF, Up_FFF, Up_F4)
from .unet import UNet  # noqa: F401
from .diffusion import Diffusion  # noqa: F401
from .training import (argument, set_seed, setup_logging, train, TrainStep, FusedAdamW, FlatParams,  # noqa: F401
                       GradAllReduce)

from .tasks import ddpm_run, rotation_results, shift_results  # noqa: F401
from .data import get_data, get_data_MNIST, save_gen_images, make_collage  # noqa: F401

__version__ = "0.1.0"
