"""Latency regime: sampling at the reference's n = 6 (ms per denoise step, eager and hipGraph replay) and the B = 16 train
step (eager and replay)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
for n in (6, 16, 64):
    for graph in (False, True):
        d = afdm.Diffusion(noise_steps=201, img_size=32, device=dev)
        d.sample(model, n=n, image_channels=3, noise_source="device", graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.sample(model, n=n, image_channels=3, noise_source="device", graph=graph)
        torch.cuda.synchronize()
        print(f"sample n={n:3d} {'graph' if graph else 'eager'}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per denoise step")
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
for B in (16, 64):
    images = torch.randn(B, 3, 32, 32, device=dev)
    for graph in (False, True):
        st = afdm.TrainStep(model, diff, lr=3e-4, graph=graph)
        for _ in range(6):
            st(images)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            st(images)
        torch.cuda.synchronize()
        print(f"train B={B:3d} {'graph' if graph else 'eager'}: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms/step")
