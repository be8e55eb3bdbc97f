"""How long the HOST needs to issue one eager train step (no device sync inside the window) vs the device time:
at a tiny batch the device is never the bound, so the eager step time there IS the host's launch time."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
for B in (2, 16, 64, 128, 256):
    st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    images = torch.randn(B, 3, 32, 32, device=dev)
    for _ in range(5):
        st(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st(images)
    t_host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 20
    print(f"B={B:3d}: host issue {t_host*1e3:6.2f} ms/step, with device {t_all*1e3:6.2f} ms/step")
