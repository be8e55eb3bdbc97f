"""One 256-image sampling job: one batch on one stream vs the same images split over 2 / 4 streams (device time per denoise step)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
T = 101
diff = afdm.Diffusion(noise_steps=T, img_size=32, device=dev)
n = 256
for tag, fn in (("1 x 256", lambda: diff.sample(model, n=n, image_channels=3)),
                ("2 x 128", lambda: diff.sample_concurrent(model, n=n, image_channels=3, batch=128, streams=2)),
                ("4 x 64", lambda: diff.sample_concurrent(model, n=n, image_channels=3, batch=64, streams=4)),
                ("3 x 86", lambda: diff.sample_concurrent(model, n=n, image_channels=3, batch=86, streams=3)),
                ("1 x 256 graph", lambda: diff.sample(model, n=n, image_channels=3, graph=True)),
                ("2 x 128 graphs", lambda: diff.sample_concurrent(model, n=n, image_channels=3, batch=128, streams=2, graph=True)),
                ("4 x 64 graphs", lambda: diff.sample_concurrent(model, n=n, image_channels=3, batch=64, streams=4, graph=True)),
                ("4 x 256 eager", lambda: diff.sample_concurrent(model, n=4 * n, image_channels=3, batch=256, streams=4)),
                ("4 x 256 graphs", lambda: diff.sample_concurrent(model, n=4 * n, image_channels=3, batch=256, streams=4, graph=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = 4 * n if tag.startswith("4 x 256") else n
    print(f"{tag}: {dt / (T - 1) * 1e3:.3f} ms per denoise step of {m} images -> {m / (dt / (T - 1) * 999):.1f} img/s at T=1000", flush=True)
