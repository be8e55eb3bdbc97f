import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import afdm
from afdm import ops
from oracle import ref_ops as R
dev = torch.device("cuda:0")
d = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
beta, alpha, ah = R.noise_schedule(1000)
g = torch.Generator().manual_seed(8)
x, e, nz = (torch.randn(4, 3, 32, 32, generator=g) for _ in range(3))
for i in (999, 500, 2, 1):
    ref = R.denoise_step(beta, alpha, ah, x, e, i, nz if i > 1 else torch.zeros_like(x))
    out = ops.denoise_step(x.to(dev), e.to(dev), nz.to(dev) if i > 1 else None, d.alpha, d.alpha_hat, d.beta, i).cpu()
    bad = (out != ref)
    ulp = (out.view(torch.int32) - ref.view(torch.int32)).abs()
    print(i, "mismatch", int(bad.sum()), "of", ref.numel(), "max ulp", int(ulp.max()))
    # constants as torch computes them on CPU vs on the device
    a, h, b = alpha[i], ah[i], beta[i]
    c_cpu = torch.stack([1 / torch.sqrt(a), (1 - a) / torch.sqrt(1 - h), torch.sqrt(b)])
    ad, hd, bd = d.alpha[i], d.alpha_hat[i], d.beta[i]
    c_dev = torch.stack([1 / torch.sqrt(ad), (1 - ad) / torch.sqrt(1 - hd), torch.sqrt(bd)]).cpu()
    print("   consts cpu", [float(v).hex() for v in c_cpu], "\n   consts dev", [float(v).hex() for v in c_dev])
    if bad.any():
        j = bad.flatten().nonzero()[0].item()
        print("   first bad idx", j, "x", float(x.flatten()[j]).hex(), "eps", float(e.flatten()[j]).hex(), "nz", float(nz.flatten()[j]).hex(),
              "got", float(out.flatten()[j]).hex(), "want", float(ref.flatten()[j]).hex())
