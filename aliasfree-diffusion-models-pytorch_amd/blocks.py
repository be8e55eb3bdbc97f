"""F5 / F7 / F8 / F10: the UNet's building blocks as parameter shells over the HIP ops.

Every class keeps the reference's attribute tree (so `state_dict()` keys, shapes, construction
order and therefore seeded initialisation are identical -- SURVEY.md section 5) but none of the
held torch.nn layers is ever *called*: `forward` goes through `ops` (libafd_hip.so) only.

Reference classes covered (modules/ddpm_utils.py): SelfAttention :54-74, DoubleConv :77-95,
DoubleConv_F :97-143, DoubleConv_F4 :145-197, Down/Up :199-245, Down_F/Up_F :253-299,
Down_FF/Up_FF :301-358, Down_FFF/Up_FFF :360-417, Down_F4/Up_F4 :419-480.
"""
import torch
import torch.nn as nn

from . import ops
from .filters import circularLowpassKernel


def _design(f_settings):
    """(down taps, up taps) as CPU tensors + host tap copies, from the f_settings contract
    (ddpm_tasks.py:47-51 -> ddpm_utils.py:102-107)."""
    if f_settings is None:
        raise ValueError("f_settings is empty")
    jinc = circularLowpassKernel(omega_c=f_settings["omega_c_down"], N=f_settings["kernel_size"], beta=f_settings["kaiser_beta"])
    sinc = circularLowpassKernel(omega_c=f_settings["omega_c_up"], N=f_settings["kernel_size"], beta=f_settings["kaiser_beta"])
    return jinc, sinc


class SelfAttention(nn.Module):
    """LN -> 4-head MHA -> +x -> LN -> Linear -> GELU -> Linear -> +   on the NCHW tensor itself:
    tokens are pixels, so every projection is a 1x1 convolution and no transpose ever runs."""

    heads = 4
    fused = True          # the token-wise chains as two fused launches (csrc/tok.hip) where the width is covered

    def __init__(self, channels, size):
        super().__init__()
        self.channels, self.size = channels, size
        self.mha = nn.MultiheadAttention(channels, self.heads, batch_first=True)
        self.ln = nn.LayerNorm([channels])
        self.ff_self = nn.Sequential(nn.LayerNorm([channels]), nn.Linear(channels, channels), nn.GELU(),
                                     nn.Linear(channels, channels))

    def forward(self, x):
        C = self.channels
        x = x.reshape(-1, C, self.size, self.size)
        if self.fused and ops.tok_supported(C):
            m, ff = self.mha, self.ff_self
            qkv, x_res = ops.AttnHead.apply(x, self.ln.weight, self.ln.bias, m.in_proj_weight, m.in_proj_bias)
            att = ops.Attention.apply(qkv, self.heads)
            return ops.AttnTail.apply(att, x_res, m.out_proj.weight, m.out_proj.bias, ff[0].weight, ff[0].bias,
                                      ff[1].weight, ff[1].bias, ff[3].weight, ff[3].bias)
        as1x1 = lambda w: w.reshape(w.shape[0], w.shape[1], 1, 1)
        h, x_res = ops.LayerNormC.apply(x, self.ln.weight, self.ln.bias)         # x_res: x for the residual, via the LN node
        lin = lambda z, w, b, res=None: ops.conv(z, as1x1(w), b, res=res, w_param=w, b_param=b)
        qkv = lin(h, self.mha.in_proj_weight, self.mha.in_proj_bias)
        att = ops.Attention.apply(qkv, self.heads)
        a = lin(att, self.mha.out_proj.weight, self.mha.out_proj.bias, x_res)
        f, a = ops.LayerNormC.apply(a, self.ff_self[0].weight, self.ff_self[0].bias)                 # (a again, via the LN node)
        if torch.is_grad_enabled():
            f = ops.Gelu.apply(lin(f, self.ff_self[1].weight, self.ff_self[1].bias))
        else:
            f = ops.conv_infer(f, as1x1(self.ff_self[1].weight), self.ff_self[1].bias, act=1)
        return lin(f, self.ff_self[3].weight, self.ff_self[3].bias, a)


class _DoubleConvBase(nn.Module):
    """conv3x3 -> GN(1) -> act -> conv3x3 -> GN(1) [-> (+x) -> act]; `emb` (B,C) is the stage's time
    embedding, folded into the last normalisation's store."""

    filtered = False

    def _parts(self):
        raise NotImplementedError

    def forward(self, x, emb=None):
        conv1, norm1, conv2, norm2 = self._parts()
        if self.residual:                                  # x for the residual comes back through the conv node
            h, x = ops.conv(x, conv1.weight, fork=True)
        else:
            h = ops.conv(x, conv1.weight)
        if self.filtered:
            h = ops.GroupNormFiltAct.apply(h, norm1.weight, norm1.bias, None, self._tu, self._td)
        else:
            h = ops.GroupNorm1.apply(h, norm1.weight, norm1.bias, None, None, 1)
        h = ops.conv(h, conv2.weight)
        if not self.residual:
            return ops.GroupNorm1.apply(h, norm2.weight, norm2.bias, None, emb, 0)
        assert emb is None
        if self.filtered:
            return ops.GroupNormFiltAct.apply(h, norm2.weight, norm2.bias, x, self._tu, self._td)
        return ops.GroupNorm1.apply(h, norm2.weight, norm2.bias, x, None, 1)


class DoubleConv(_DoubleConvBase):
    """ddpm_utils.py:77-95 -- parameters under `double_conv.{0,1,3,4}`."""

    def __init__(self, in_channels, out_channels, mid_channels=None, residual=False):
        super().__init__()
        self.residual = residual
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1, bias=False), nn.GroupNorm(1, mid), nn.GELU(),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1, bias=False), nn.GroupNorm(1, out_channels))

    def _parts(self):
        s = self.double_conv
        return s[0], s[1], s[3], s[4]


class DoubleConv_F(_DoubleConvBase):
    """ddpm_utils.py:97-143 -- parameters under `conv1/norm1/conv2/norm2`; activations are the
    fused GroupNorm -> up2 -> GELU -> down2 kernel."""

    filtered = True

    def __init__(self, in_channels, out_channels, mid_channels=None, residual=False, f_settings=None):
        super().__init__()
        self.residual, self.f_settings = residual, f_settings
        self.jinc_filter, self.sinc_filter = _design(f_settings)
        self._td, self._tu = ops.Taps(self.jinc_filter), ops.Taps(self.sinc_filter)
        mid = mid_channels or out_channels
        self.conv1 = nn.Conv2d(in_channels, mid, kernel_size=3, padding=1, bias=False)
        self.norm1 = nn.GroupNorm(1, mid)
        self.gelu = nn.GELU()
        self.conv2 = nn.Conv2d(mid, out_channels, kernel_size=3, padding=1, bias=False)
        self.norm2 = nn.GroupNorm(1, out_channels)

    def _parts(self):
        return self.conv1, self.norm1, self.conv2, self.norm2


class DoubleConv_F4(DoubleConv_F):
    """ddpm_utils.py:145-197 (variant 4, experimental): the normalisation moves to the 2x grid,
    between the upsampler and the GELU.  Composed from the unfused HIP ops."""

    def forward(self, x, emb=None):
        def sandwich(h, norm):
            u = ops.FiltUp2.apply(h, self._tu)
            u = ops.GroupNorm1.apply(u, norm.weight, norm.bias, None, None, 1)      # norm + GELU on the 2x grid
            return ops.FiltDown2.apply(u, self._td)
        h = sandwich(ops.conv(x, self.conv1.weight), self.norm1)
        h = ops.GroupNorm1.apply(ops.conv(h, self.conv2.weight), self.norm2.weight, self.norm2.bias,
                                 x if self.residual else None, None if self.residual else emb, 0)
        return sandwich(h, self.norm2) if self.residual else h


def _emb_layer(emb_dim, out_channels):
    return nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, out_channels))


class _Stage(nn.Module):
    _emb_pre = None       # (t, emb) set by UNet.forward for ONE call: this stage's time embedding of THAT t, computed with the
    #                       other stages' in one launch

    def _emb(self, t):
        pre, self._emb_pre = self._emb_pre, None
        lin = self.emb_layer[1]
        if pre is not None and pre[0] is t:
            # the value is there already; the autograd node is created HERE, in this stage's forward, so that backward
            # reaches it (and this stage's emb_layer gradients) when it reaches the stage -- not at the very end
            return ops.SiluLinearPre.apply(pre[1], t, lin.weight, lin.bias) if torch.is_grad_enabled() else pre[1]
        return ops.SiluLinear.apply(t, lin.weight, lin.bias)


class _DownStage(_Stage):
    """resample(2x down) -> DoubleConv(residual) -> DoubleConv -> + Linear(SiLU(t))."""

    def __init__(self, in_channels, out_channels, emb_dim, f_settings, pooled, filtered_act, v4=False):
        super().__init__()
        self.f_settings, self.pooled = f_settings, pooled
        if not pooled:
            self.jinc_filter, _ = _design(f_settings)
            self._td = ops.Taps(self.jinc_filter)
        if v4:
            mk = lambda i, o, res=False: DoubleConv_F4(i, o, residual=res, f_settings=f_settings)
        elif filtered_act:
            mk = lambda i, o, res=False: DoubleConv_F(i, o, residual=res, f_settings=f_settings)
        else:
            mk = lambda i, o, res=False: DoubleConv(i, o, residual=res)
        pair = [mk(in_channels, in_channels, True), mk(in_channels, out_channels)]
        if pooled:      # reference keeps the pool inside the Sequential -> keys `maxpool_conv.{1,2}.*`
            self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), *pair)
        else:
            self.conv = nn.Sequential(*pair)
        self.emb_layer = _emb_layer(emb_dim, out_channels)
        if v4:
            self.norm1 = nn.GroupNorm(1, in_channels)                 # present, unused (ddpm_utils.py:440,445)

    def forward(self, x, t):
        if self.pooled:
            x = ops.MaxPool2.apply(x)
            a, b = self.maxpool_conv[1], self.maxpool_conv[2]
        else:
            x = ops.FiltDown2.apply(x, self._td)
            a, b = self.conv[0], self.conv[1]
        return b(a(x), self._emb(t))


class _UpStage(_Stage):
    """resample(2x up) -> cat[skip, x] -> DoubleConv(residual) -> DoubleConv(mid=in/2) -> + emb."""

    def __init__(self, in_channels, out_channels, emb_dim, f_settings, bilinear, filtered_act, v4=False):
        super().__init__()
        self.f_settings, self.bilinear = f_settings, bilinear
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self._tu = None
        else:
            _, self.sinc_filter = _design(f_settings)
            self._tu = ops.Taps(self.sinc_filter)
        if v4:
            mk = lambda i, o, m=None, res=False: DoubleConv_F4(i, o, m, residual=res, f_settings=f_settings)
        elif filtered_act:
            mk = lambda i, o, m=None, res=False: DoubleConv_F(i, o, m, residual=res, f_settings=f_settings)
        else:
            mk = lambda i, o, m=None, res=False: DoubleConv(i, o, m, residual=res)
        self.conv = nn.Sequential(mk(in_channels, in_channels, None, True), mk(in_channels, out_channels, in_channels // 2))
        self.emb_layer = _emb_layer(emb_dim, out_channels)
        if v4:
            self.norm1 = nn.GroupNorm(1, in_channels // 2)            # present, unused (ddpm_utils.py:471,476)

    def forward(self, x, skip_x, t):
        x = ops.UpCat.apply(x, skip_x, self._tu, "bilinear" if self.bilinear else "filt")
        return self.conv[1](self.conv[0](x), self._emb(t))


# the reference's class names: real nn.Module subclasses of the two stage shells (isinstance / subclassing work as in the
# reference; the shells hold the one shared implementation)
class Down(_DownStage):
    """ddpm_utils.py:199-219."""

    def __init__(self, in_channels, out_channels, emb_dim=256):
        super().__init__(in_channels, out_channels, emb_dim, None, pooled=True, filtered_act=False)


class Down_F(_DownStage):
    """ddpm_utils.py:253-274."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        _design(f_settings)
        super().__init__(in_channels, out_channels, emb_dim, f_settings, pooled=True, filtered_act=True)


class Down_FF(_DownStage):
    """ddpm_utils.py:301-328."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, pooled=False, filtered_act=False)


class Down_FFF(_DownStage):
    """ddpm_utils.py:360-387."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, pooled=False, filtered_act=True)


class Down_F4(_DownStage):
    """ddpm_utils.py:419-448."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, pooled=False, filtered_act=True, v4=True)


class Up(_UpStage):
    """ddpm_utils.py:222-245."""

    def __init__(self, in_channels, out_channels, emb_dim=256):
        super().__init__(in_channels, out_channels, emb_dim, None, bilinear=True, filtered_act=False)


class Up_F(_UpStage):
    """ddpm_utils.py:276-299."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        _design(f_settings)
        super().__init__(in_channels, out_channels, emb_dim, f_settings, bilinear=True, filtered_act=True)


class Up_FF(_UpStage):
    """ddpm_utils.py:330-358."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, bilinear=False, filtered_act=False)


class Up_FFF(_UpStage):
    """ddpm_utils.py:389-417."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, bilinear=False, filtered_act=True)


class Up_F4(_UpStage):
    """ddpm_utils.py:450-480."""

    def __init__(self, in_channels, out_channels, emb_dim=256, f_settings=None):
        super().__init__(in_channels, out_channels, emb_dim, f_settings, bilinear=False, filtered_act=True, v4=True)
