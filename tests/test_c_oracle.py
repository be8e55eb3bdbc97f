"""Pin the plain-C oracle (oracle/afd_oracle.c) against the reference's golden vectors, and
cross-check it against the Python oracle.  CPU only."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_json, load_golden, rel_l2
from oracle import ref_ops as R

SO = os.path.join(ROOT, "oracle", "libafd_oracle.so")
FP, I64P, U8P = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_uint8))


@pytest.fixture(scope="module")
def C():
    if not os.path.exists(SO):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    return ctypes.CDLL(SO)


def fp(a):
    return a.ctypes.data_as(FP)


def test_c_filter_design_vs_reference(C):
    g = load_golden("filters.npz")
    worst = 0.0
    for key, omega, N, beta in golden_json(g, "grid"):
        out = np.empty((N, N), dtype=np.float32)
        C.orc_lowpass_kernel(ctypes.c_double(omega), N, int(beta >= 0), ctypes.c_double(max(beta, 0.0)), fp(out))
        worst = max(worst, float(np.abs(out - g[key]).max()))
    assert worst < 3e-7      # independent J1 / I0 (Bessel integrals): agreement to fp32 rounding


def test_c_resampling_vs_reference(C):
    g = load_golden("resample.npz")
    for tag in golden_json(g, "cases"):
        x, ku, kd = g[f"x_{tag}"], np.ascontiguousarray(g[f"ku_{tag}"]), np.ascontiguousarray(g[f"kd_{tag}"])
        B, Cc, H, W = x.shape
        N = ku.shape[0]
        x = np.ascontiguousarray(x)
        up = np.empty((B, Cc, 2 * H, 2 * W), dtype=np.float32)
        C.orc_up2(fp(x), fp(up), B * Cc, H, W, fp(ku), N)
        assert rel_l2(up, g[f"up_{tag}_y"]) < 2e-6, tag
        dn = np.empty((B, Cc, (H + 1) // 2, (W + 1) // 2), dtype=np.float32)
        C.orc_down2(fp(x), fp(dn), B * Cc, H, W, fp(ku), N)
        assert rel_l2(dn, g[f"down_{tag}_y"]) < 2e-6, tag
        act = np.empty_like(x)
        C.orc_filt_act(fp(x), fp(act), B * Cc, H, W, fp(ku), fp(kd), N)
        assert rel_l2(act, g[f"act_{tag}_y"]) < 2e-6, tag


def test_c_schedule_noise_denoise_quantise_bit_exact(C):
    g = load_golden("schedule.npz")
    for T in (1000, 300, 101, 11):
        a, ah = (np.empty(T, dtype=np.float32) for _ in range(2))
        C.orc_schedule_from_beta(T, fp(np.ascontiguousarray(g[f"beta_{T}"])), fp(a), fp(ah))
        assert np.array_equal(a, g[f"alpha_{T}"]) and np.array_equal(ah, g[f"alpha_hat_{T}"])
    x, eps, t = (np.ascontiguousarray(g[k]) for k in ("noise_x", "noise_eps", "noise_t"))
    xt = np.empty_like(x)
    C.orc_noise_images(fp(x), fp(eps), t.ctypes.data_as(I64P), fp(np.ascontiguousarray(g["alpha_hat_1000"])), fp(xt),
                       x.shape[0], ctypes.c_long(x[0].size))
    assert np.array_equal(xt, g["noise_xt"])
    xx, e, nz = (np.ascontiguousarray(g[k]) for k in ("den_x", "den_eps", "den_noise"))
    al, ah, be = (np.ascontiguousarray(g[k]) for k in ("alpha_1000", "alpha_hat_1000", "beta_1000"))
    for i in (999, 500, 2, 1):
        out = np.empty_like(xx)
        C.orc_denoise_step(fp(xx), fp(e), fp(nz) if i > 1 else None, fp(al), fp(ah), fp(be), i, fp(out), ctypes.c_long(xx.size))
        assert np.array_equal(out, g[f"den_out_{i}"]), i
    v = np.ascontiguousarray(g["quant_in"])
    q = np.empty(v.size, dtype=np.uint8)
    C.orc_quantize_u8(fp(v), q.ctypes.data_as(U8P), ctypes.c_long(v.size))
    assert np.array_equal(q, g["quant_out"])


def test_c_groupnorm_vs_python_oracle(C):
    gg = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 4, 4, generator=gg) * 2 + 1
    gamma, beta = torch.randn(6, generator=gg), torch.randn(6, generator=gg)
    y = np.empty((2, 6, 4, 4), dtype=np.float32)
    C.orc_groupnorm1(fp(x.numpy()), fp(y), 2, 6, 16, ctypes.c_float(1e-5), fp(gamma.numpy()), fp(beta.numpy()))
    assert rel_l2(y, R.groupnorm1(x, gamma, beta)) < 1e-6
