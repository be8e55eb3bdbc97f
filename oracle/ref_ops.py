"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A functional, state_dict-keyed CPU restatement (numpy / torch-CPU, fp32 or fp64) of the reference's
hot path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module; the product package (`aliasfree-diffusion-models-pytorch_amd/`) never does and raises
if its HIP library is missing.

Pinning: the reference has no tests of its own, so this oracle is pinned by
`tests/golden/*.npz`, produced by running the reference itself on CPU in the build container
(`tests/golden/make_golden.py`); `tests/test_oracle_golden.py` checks every function here against
those vectors.  Third-party arithmetic (torch ATen conv/matmul/softmax, scipy.special.j1,
numpy.kaiser, scipy.ndimage) is used here exactly where the reference calls it.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# F1  filter design                                             modules/filtrs.py:20-37
# --------------------------------------------------------------------------------------------

def lowpass_kernel(omega_c=math.pi, N=6, beta=None):
    """Radial jinc low-pass, optional separable Kaiser window, unit DC gain, fp64 -> fp32.

    filtrs.py:22 (jinc of the distance to the centre (N-1)/2), :23-24 (odd-N centre tap patched
    to omega_c^2/(4 pi) BEFORE windowing), :26-34 (outer-product Kaiser), :36 (sum-normalise),
    :37 (cast)."""
    from scipy.special import j1
    c = (N - 1) / 2.0
    idx = np.arange(N, dtype=np.float64)
    r = np.sqrt((idx[:, None] - c) ** 2 + (idx[None, :] - c) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = omega_c * j1(omega_c * r) / (2 * np.pi * r)
    if N % 2 == 1:
        k[(N - 1) // 2, (N - 1) // 2] = omega_c ** 2 / (4 * np.pi)
    if beta is not None:
        w = np.kaiser(N, beta)
        k = k * np.outer(w, w)
    k = k / np.sum(k)
    return torch.tensor(k, dtype=torch.float32)


# --------------------------------------------------------------------------------------------
# F2 / F3 / F4  filtered resampling, written as explicit tap sums (no conv2d)
# --------------------------------------------------------------------------------------------

def _same_pad(N):
    """torch 'same' padding for a stride-1 N-tap correlation: left (N-1)//2, right the rest."""
    lo = (N - 1) // 2
    return lo, (N - 1) - lo


def _correlate_same(x, k):
    """Depthwise cross-correlation with zero 'same' padding; identical taps on every channel
    (filtrs.py:73-75 / :91-93)."""
    N = k.shape[0]
    lo, hi = _same_pad(N)
    H, W = x.shape[-2:]
    xp = F.pad(x, (lo, hi, lo, hi))
    y = torch.zeros_like(x)
    for a in range(N):
        for b in range(N):
            y = y + k[a, b].to(x.dtype) * xp[..., a:a + H, b:b + W]
    return y


def filt_down2(x, k):
    """custom_downsample (filtrs.py:71-77): filter at full rate, keep even rows / columns."""
    return _correlate_same(x, k)[..., ::2, ::2]


def filt_up2(x, k):
    """custom_upsample (filtrs.py:79-94): zero-stuff onto the 2x grid (samples at even
    positions), then filter.  No x4 gain: DC gain is 1/4."""
    B, C, H, W = x.shape
    z = torch.zeros(B, C, 2 * H, 2 * W, dtype=x.dtype)
    z[..., ::2, ::2] = x
    return _correlate_same(z, k)


def gelu_erf(x):
    """Exact GELU, nn.GELU() default (ddpm_utils.py:114)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def filt_act(x, k_up, k_down):
    """Filtered nonlinearity of DoubleConv_F (ddpm_utils.py:123-125): up2 -> GELU -> down2."""
    return filt_down2(gelu_erf(filt_up2(x, k_up)), k_down)


def filt_act_n3_closed_form(x, ku, kd):
    """The N=3 polyphase closed form the HIP kernel implements,
    restated with slices so tests can check the derivation against `filt_act`."""
    B, C, H, W = x.shape
    xp = F.pad(x, (0, 1, 0, 1))                      # x[H,.] = x[.,W] = 0
    xe, xr, xd, xdr = xp[..., :H, :W], xp[..., :H, 1:], xp[..., 1:, :W], xp[..., 1:, 1:]
    g00 = gelu_erf(ku[1, 1] * xe)
    g01 = gelu_erf(ku[1, 0] * xe + ku[1, 2] * xr)
    g10 = gelu_erf(ku[0, 1] * xe + ku[2, 1] * xd)
    g11 = gelu_erf(ku[0, 0] * xe + ku[0, 2] * xr + ku[2, 0] * xd + ku[2, 2] * xdr)
    G = torch.zeros(B, C, 2 * H, 2 * W, dtype=x.dtype)
    G[..., 0::2, 0::2], G[..., 0::2, 1::2], G[..., 1::2, 0::2], G[..., 1::2, 1::2] = g00, g01, g10, g11
    Gp = F.pad(G, (1, 0, 1, 0))                      # G[-1,.] = G[.,-1] = 0
    y = torch.zeros_like(x)
    for a in range(3):
        for b in range(3):
            y = y + kd[a, b] * Gp[..., a:a + 2 * H:2, b:b + 2 * W:2]
    return y


# --------------------------------------------------------------------------------------------
# F6  GroupNorm(1, C), F5 convs, LayerNorm, attention -- torch ATen where the reference uses it
# --------------------------------------------------------------------------------------------

def groupnorm1(x, weight, bias, eps=1e-5):
    """nn.GroupNorm(1, C) (ddpm_utils.py:85,88,113,116): one group = all C*H*W per sample,
    biased variance, per-channel affine."""
    B = x.shape[0]
    flat = x.reshape(B, -1)
    mean = flat.mean(dim=1)
    var = ((flat - mean[:, None]) ** 2).mean(dim=1)
    xh = (x - mean[:, None, None, None]) * torch.rsqrt(var + eps)[:, None, None, None]
    return xh * weight[None, :, None, None] + bias[None, :, None, None]


def conv3x3(x, w):
    return F.conv2d(x, w, None, padding=1)


def self_attention(sd, p, x, heads=4):
    """SelfAttention.forward (ddpm_utils.py:68-74) on NCHW input; `p` = key prefix in `sd`."""
    B, C, H, W = x.shape
    L, d = H * W, C // heads
    tok = x.reshape(B, C, L).transpose(1, 2)                         # (B, L, C)      :69
    h = F.layer_norm(tok, (C,), sd[p + "ln.weight"], sd[p + "ln.bias"])              # :70
    qkv = h @ sd[p + "mha.in_proj_weight"].T + sd[p + "mha.in_proj_bias"]            # :71
    q, k, v = qkv.split(C, dim=-1)
    sh = lambda z: z.reshape(B, L, heads, d).transpose(1, 2)          # (B, heads, L, d)
    att = torch.softmax((sh(q) / math.sqrt(d)) @ sh(k).transpose(-1, -2), dim=-1) @ sh(v)
    att = att.transpose(1, 2).reshape(B, L, C)
    att = att @ sd[p + "mha.out_proj.weight"].T + sd[p + "mha.out_proj.bias"]
    a = att + tok                                                                     # :72
    f = F.layer_norm(a, (C,), sd[p + "ff_self.0.weight"], sd[p + "ff_self.0.bias"])
    f = gelu_erf(f @ sd[p + "ff_self.1.weight"].T + sd[p + "ff_self.1.bias"])
    f = f @ sd[p + "ff_self.3.weight"].T + sd[p + "ff_self.3.bias"]
    return (f + a).transpose(1, 2).reshape(B, C, H, W)                                # :73-74


# --------------------------------------------------------------------------------------------
# F5 / F7 / F8 / F11  blocks and the UNet, functional over a reference-keyed state_dict
# --------------------------------------------------------------------------------------------

def _dc_keys(sd, p):
    """The two key schemes of a double-conv block: `double_conv.{0,1,3,4}` (DoubleConv,
    ddpm_utils.py:83-89) or `conv1/norm1/conv2/norm2` (DoubleConv_F, :112-116)."""
    if p + "double_conv.0.weight" in sd:
        return (p + "double_conv.0.weight", p + "double_conv.1.", p + "double_conv.3.weight", p + "double_conv.4."), False
    return (p + "conv1.weight", p + "norm1.", p + "conv2.weight", p + "norm2."), True


def double_conv(sd, p, x, residual, filt=None):
    """DoubleConv.forward (ddpm_utils.py:91-95) / DoubleConv_F.forward (:118-143).
    `filt` = (k_up, k_down) selects the filtered form."""
    (c1, n1, c2, n2), _ = _dc_keys(sd, p)
    act = (lambda z: filt_act(z, *filt)) if filt is not None else gelu_erf
    h = groupnorm1(conv3x3(x, sd[c1]), sd[n1 + "weight"], sd[n1 + "bias"])
    h = act(h)
    h = groupnorm1(conv3x3(h, sd[c2]), sd[n2 + "weight"], sd[n2 + "bias"])
    if residual:
        return act(h + x)          # F.gelu(x + dc(x)) :93  |  (dc(x) + x) -> up/gelu/down :128-131
    return h


def time_embedding(t, channels=256):
    """UNet.pos_encoding (ddpm_models.py:261-269) after t.unsqueeze(-1).float() (:272)."""
    tf = t.reshape(-1, 1).to(torch.float32)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, channels, 2).float() / channels))
    return torch.cat([torch.sin(tf.repeat(1, channels // 2) * inv_freq),
                      torch.cos(tf.repeat(1, channels // 2) * inv_freq)], dim=-1)


def emb_add(sd, p, x, temb):
    """emb_layer = SiLU -> Linear, broadcast-added (ddpm_utils.py:208-219)."""
    e = F.silu(temb) @ sd[p + "emb_layer.1.weight"].T + sd[p + "emb_layer.1.bias"]
    return x + e[:, :, None, None].to(x.dtype)


_STAGE_SEQ = {0: "maxpool_conv.", 2: "maxpool_conv."}   # Down / Down_F keep the Sequential with the pool at index 0


def down_stage(sd, p, x, temb, variant, filt):
    """Down (ddpm_utils.py:199-219), Down_F (:253-274), Down_FF (:301-328), Down_FFF (:360-387)."""
    if variant in (0, 2):
        x = F.max_pool2d(x, 2)
        q0, q1 = p + "maxpool_conv.1.", p + "maxpool_conv.2."
    else:
        x = filt_down2(x, filt[1])
        q0, q1 = p + "conv.0.", p + "conv.1."
    f = filt if variant in (2, 3) else None
    x = double_conv(sd, q0, x, True, f)
    x = double_conv(sd, q1, x, False, f)
    return emb_add(sd, p, x, temb)


def up_stage(sd, p, x, skip, temb, variant, filt):
    """Up (ddpm_utils.py:222-245), Up_F (:276-299), Up_FF (:330-358), Up_FFF (:389-417)."""
    if variant in (0, 2):
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        x = filt_up2(x, filt[0])
    x = torch.cat([skip, x], dim=1)
    f = filt if variant in (2, 3) else None
    x = double_conv(sd, p + "conv.0.", x, True, f)
    x = double_conv(sd, p + "conv.1.", x, False, f)
    return emb_add(sd, p, x, temb)


def unet_forward(sd, x, t, variant, f_settings=None, time_dim=256, y=None):
    """UNet.forward (ddpm_models.py:271-298) for variants 0..3; `y` (class labels) adds label_emb(y) to the time
    embedding (:276-277)."""
    filt = None
    if variant:
        filt = (lowpass_kernel(f_settings["omega_c_up"], f_settings["kernel_size"], f_settings["kaiser_beta"]).to(x.dtype),
                lowpass_kernel(f_settings["omega_c_down"], f_settings["kernel_size"], f_settings["kaiser_beta"]).to(x.dtype))
    f = filt if variant in (2, 3) else None
    temb = time_embedding(t, time_dim).to(x.dtype)
    if y is not None:
        temb = temb + sd["label_emb.weight"][y].to(x.dtype)
    x1 = double_conv(sd, "inc.", x, False, f)
    x2 = self_attention(sd, "sa1.", down_stage(sd, "down1.", x1, temb, variant, filt))
    x3 = self_attention(sd, "sa2.", down_stage(sd, "down2.", x2, temb, variant, filt))
    x4 = self_attention(sd, "sa3.", down_stage(sd, "down3.", x3, temb, variant, filt))
    x4 = double_conv(sd, "bot1.", x4, False, f)
    x4 = double_conv(sd, "bot2.", x4, False, f)
    x4 = double_conv(sd, "bot3.", x4, False, f)
    y = self_attention(sd, "sa4.", up_stage(sd, "up1.", x4, x3, temb, variant, filt))
    y = self_attention(sd, "sa5.", up_stage(sd, "up2.", y, x2, temb, variant, filt))
    y = self_attention(sd, "sa6.", up_stage(sd, "up3.", y, x1, temb, variant, filt))
    return F.conv2d(y, sd["outc.weight"], sd["outc.bias"])


# --------------------------------------------------------------------------------------------
# F12-F16  diffusion process
# --------------------------------------------------------------------------------------------

def noise_schedule(T=1000, beta_start=1e-4, beta_end=0.02):
    """Diffusion.__init__ / prepare_noise_schedule (ddpm_models.py:309-315): fp32 linspace,
    alpha = 1 - beta, alpha_hat = sequential fp32 cumprod."""
    beta = torch.linspace(beta_start, beta_end, T)
    alpha = 1.0 - beta
    return beta, alpha, torch.cumprod(alpha, dim=0)


def noise_images(alpha_hat, x, t, eps):
    """Diffusion.noise_images (ddpm_models.py:317-321) with the noise injected."""
    a = torch.sqrt(alpha_hat[t])[:, None, None, None]
    b = torch.sqrt(1 - alpha_hat[t])[:, None, None, None]
    return a * x + b * eps


def denoise_step(beta, alpha, alpha_hat, x, eps_pred, i, noise):
    """One iteration of Diffusion.sample / revert (ddpm_models.py:367-374 / :335-342)."""
    t = torch.full((x.shape[0],), i, dtype=torch.long)
    a, ah, b = alpha[t][:, None, None, None], alpha_hat[t][:, None, None, None], beta[t][:, None, None, None]
    return 1 / torch.sqrt(a) * (x - ((1 - a) / (torch.sqrt(1 - ah))) * eps_pred) + torch.sqrt(b) * noise


def quantize_u8(x):
    """ddpm_models.py:381-382: clamp, shift to [0,1], scale, TRUNCATE to uint8."""
    return (((x.clamp(-1, 1) + 1) / 2) * 255).type(torch.uint8)


def sample_loop(eps_model, T, n, c, S, theta=None, snapshots=True):
    """Diffusion.sample (ddpm_models.py:352-386).  Noise comes from the torch CPU global
    generator in the reference's call order: x_T, then one randn_like per step with i > 1."""
    beta, alpha, alpha_hat = noise_schedule(T)
    x = torch.randn((n, c, S, S))
    result = []
    for i in reversed(range(1, T)):
        t = (torch.ones(n) * i).long()
        eps = eps_model(x, t)
        nz = torch.randn_like(x) if i > 1 else torch.zeros_like(x)
        x = denoise_step(beta, alpha, alpha_hat, x, eps, i, nz)
        if theta is not None:
            x = rotate_wrap(x, theta / T)
        if i % 100 == 0:
            result.append(x)
    result.append(x)
    return x, quantize_u8(x), quantize_u8(torch.cat(result))


def rotate_wrap(x, degrees):
    """Diffusion.rotate_2d_matrix (ddpm_models.py:421-429): scipy order-3 spline, grid-wrap."""
    from scipy import ndimage
    r = ndimage.rotate(input=x.numpy(), angle=degrees, axes=(2, 3), reshape=False, mode="grid-wrap")
    return torch.from_numpy(r)


# --------------------------------------------------------------------------------------------
# F15  train-step body
# --------------------------------------------------------------------------------------------

def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults as the reference constructs it (ddpm_utils.py:489)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def train_step_loss_and_grads(sd, images, t, eps, variant, f_settings, alpha_hat):
    """ddpm_utils.py:500-506 with t and eps injected.  `sd` tensors must require grad.  Parameters the forward does
    not use (label_emb of an unconditional step) get no gradient, as in the reference (grad None -> AdamW skips them)."""
    x_t = noise_images(alpha_hat, images, t, eps)
    pred = unet_forward(sd, x_t, t, variant, f_settings)
    loss = F.mse_loss(eps, pred)
    names = [k for k, v in sd.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
    return loss.detach(), pred.detach(), {k: g for k, g in zip(names, grads) if g is not None}
