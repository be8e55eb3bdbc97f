"""Host-side cost of one eager train step: cProfile over 20 steps at a small batch (the GPU then never blocks the host),
top functions by own time.   python tools/host_profile.py [batch=16]"""
import sys, os, math, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
images = torch.randn(B, 3, 32, 32, device=dev)
for _ in range(5):
    st(images)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    st(images)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step eager")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    st(images)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(28)
