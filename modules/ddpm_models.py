from afdm.unet import UNet  # noqa: F401
from afdm.diffusion import Diffusion  # noqa: F401
from modules.ddpm_utils import *  # noqa: F401,F403
