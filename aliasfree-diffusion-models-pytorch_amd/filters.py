"""F1-F3 functional API with the reference's names and argument order (modules/filtrs.py:20,71,79).

`circularLowpassKernel` is host code inside libafd_hip.so (afd_lowpass_kernel); the two resamplers
are the HIP kernels, differentiable w.r.t. `x`.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import lib


def circularLowpassKernel(omega_c=math.pi, N=6, beta=None):
    """N x N jinc * Kaiser low-pass, unit DC gain, FloatTensor[N, N] on the CPU (filtrs.py:20-37)."""
    taps = np.empty((int(N), int(N)), dtype=np.float32)
    lib().afd_lowpass_kernel(float(omega_c), int(N), 0 if beta is None else 1, 0.0 if beta is None else float(beta),
                             taps.ctypes.data)
    return torch.from_numpy(taps)


def _taps(f):
    t = getattr(f, "_afd_taps", None)
    if t is None:
        t = ops.Taps(f)
        if isinstance(f, torch.Tensor):
            try:
                f._afd_taps = t          # cache the host copy on the filter tensor itself
            except Exception:
                pass
    return t


def custom_downsample(x, jinc_filter, factor=2):
    """Low-pass then keep every 2nd row / column (filtrs.py:71-77)."""
    if factor != 2:
        raise ValueError("afdm: custom_downsample supports factor=2 only (the reference never uses another)")
    return ops.FiltDown2.apply(x, _taps(jinc_filter))


def custom_upsample(x, sinc_filter, factor=2):
    """Zero-stuff to 2x then low-pass; DC gain 1/4 like the reference (filtrs.py:79-94)."""
    if factor != 2:
        raise ValueError("afdm: custom_upsample supports factor=2 only (the reference never uses another)")
    return ops.FiltUp2.apply(x, _taps(sinc_filter))
