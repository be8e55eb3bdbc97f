"""Randomised cross-check of the bf16x3 / streaming kernels against the fp32 Winograd / direct kernels at batch sizes and shapes
the unit tests do not pin (odd batches, ragged last tiles, every map size): forward, dgrad, wgrad (3x3) and the 1x1 weight
gradient.  Prints the worst relative L2 difference; exits non-zero above 5e-6."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
from afdm import ops
dev = torch.device("cuda:0"); L = afdm.lib()
g = torch.Generator().manual_seed(5)
worst = (0.0, None)
def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
def modes(ms):
    for m in ms: L.afd_debug_conv_path(m)
cases = []
for S, chans in ((32, [(32, 32), (64, 32), (64, 64), (3, 32)]), (16, [(32, 64), (128, 128), (64, 32), (96, 160)]), (8, [(128, 128), (256, 128), (64, 128), (256, 256)]),
                 (4, [(128, 128), (256, 128), (128, 256)])):
    for (ci, co) in chans:
        for B in (33, 100, 129, 257):
            cases.append((B, ci, co, S))
n = 0
for (B, ci, co, S) in cases:
    x = torch.randn(B, ci, S, S, generator=g).to(dev); w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(dev)
    dy = torch.randn(B, co, S, S, generator=g).to(dev)
    out = {}
    for name, ms in (("new", (80, 84, 88, 92)), ("old", (81, 85, 89, 93))):
        modes(ms)
        ops.bump_param_epoch()
        xd, wd = x.clone().requires_grad_(ci > 3), w.clone().requires_grad_(True)
        y = ops.conv(xd, wd)
        gr = torch.autograd.grad(y, (xd, wd) if ci > 3 else (wd,), dy)
        out[name] = (y.detach(),) + tuple(gr)
    for a, b, what in zip(out["new"], out["old"], ("y", "dx", "dw") if ci > 3 else ("y", "dw")):
        e = rel(a, b); n += 1
        if e > worst[0]: worst = (e, (B, ci, co, S, what))
# 1x1 weight / bias gradients and the output layer
for (B, ci, co, S) in ((33, 32, 96, 32), (100, 64, 192, 16), (129, 128, 384, 8), (257, 64, 64, 16), (100, 32, 3, 32), (33, 64, 1, 16)):
    x = torch.randn(B, ci, S, S, generator=g).to(dev); w = (torch.randn(co, ci, 1, 1, generator=g) / ci ** 0.5).to(dev); b = torch.randn(co, generator=g).to(dev)
    dy = torch.randn(B, co, S, S, generator=g).to(dev)
    out = {}
    for name, ms in (("new", (80, 84, 88, 92)), ("old", (81, 85, 89, 93))):
        modes(ms)
        xd, wd, bd = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv(xd, wd, bd)
        out[name] = (y.detach(),) + tuple(torch.autograd.grad(y, (xd, wd, bd), dy))
    for a, b_, what in zip(out["new"], out["old"], ("y", "dx", "dw", "db")):
        e = rel(a, b_); n += 1
        if e > worst[0]: worst = (e, (B, ci, co, S, "1x1 " + what))
modes((80, 84, 88, 92))
print(f"{n} comparisons, worst relative L2 difference {worst[0]:.2e} at {worst[1]}")
sys.exit(0 if worst[0] < 5e-6 else 1)
