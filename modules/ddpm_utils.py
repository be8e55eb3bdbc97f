from afdm.blocks import *  # noqa: F401,F403
from afdm.blocks import (SelfAttention, DoubleConv, DoubleConv_F, DoubleConv_F4, Down, Down_F, Down_FF, Down_FFF, Down_F4,  # noqa: F401
                         Up, Up_F, Up_FF, Up_FFF, Up_F4)
from afdm.training import argument, train, setup_logging, set_seed  # noqa: F401
from afdm.filters import circularLowpassKernel, custom_downsample, custom_upsample  # noqa: F401
