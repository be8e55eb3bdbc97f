from afdm.filters import circularLowpassKernel, custom_downsample, custom_upsample  # noqa: F401
