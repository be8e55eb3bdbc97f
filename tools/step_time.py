"""The eager two-stream B=256 train step, ms/step (for environment-variable sweeps: one process per setting)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
for _ in range(10):
    st(images)
res = []
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        st(images)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 40 * 1e3)
print(os.environ.get("TAG", ""), " ".join(f"{r:.3f}" for r in res), "ms/step", flush=True)
