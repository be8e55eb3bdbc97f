"""Is the eager main-stream chain host-bound?  The chain alone (parameter-gradient launches dropped), eager vs replayed from a
hipGraph (no host in the loop)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
from afdm import ops
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
ops.defer_to_side_stream = lambda fn, *keep, writes=(): None
for tag, kw in (("chain only, eager", dict(graph=False)), ("chain only, graph replay", dict(graph=True))):
    st = afdm.TrainStep(model, diff, lr=3e-4, **kw)
    for _ in range(8):
        st(images)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        st(images)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms/step", flush=True)
