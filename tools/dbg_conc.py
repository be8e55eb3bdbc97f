"""Debug: concurrent-stream sampling vs one stream, which images differ (forced Winograd, cold cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
from afdm import ops
dev = torch.device("cuda:0"); L = afdm.lib()
F_SET = {"kernel_size": 12, "kaiser_beta": 8.0, "up_factor": 2, "down_factor": 2, "cutoff": 0.5, "filter_size": 12} if False else None
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_model as T
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(T.F_SET), device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=9, img_size=32, device=dev)
def noise_fn(k, i, shape):
    g = torch.Generator().manual_seed(77 * k + i)
    return torch.randn(shape, generator=g).to(dev)
for bf in (80, 81):
    for force in ((67, 98), (64, 96)):
        L.afd_debug_conv_path(bf)
        for m in force: L.afd_debug_conv_path(m)
        outs = []
        for streams in (3, 1, 3, 1):
            ops.bump_param_epoch(); diff._t_cache.clear()
            outs.append(diff.sample_concurrent(model, n=37, image_channels=3, batch=16, streams=streams, noise_fn=noise_fn)[0].cpu())
            torch.cuda.synchronize()
        for a in range(4):
            for b in range(a + 1, 4):
                d = [(i, int((outs[a][i].int() - outs[b][i].int()).abs().max())) for i in range(37) if not torch.equal(outs[a][i], outs[b][i])]
                print("bf", bf, "force", force, "runs", a, b, "differing images:", d)
L.afd_debug_conv_path(80); L.afd_debug_conv_path(64); L.afd_debug_conv_path(96)
