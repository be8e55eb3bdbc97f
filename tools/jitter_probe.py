"""Window-to-window variation of the eager B=256 train step: 40 windows of 20 steps, Python's cyclic GC on / off, hipGraph replay."""
import sys, os, math, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); B = 256
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
def windows(st, n=40, k=20):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            st(images)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / k * 1e3)
    return out
def show(tag, w):
    s = sorted(w)
    print(f"{tag}: min {s[0]:.3f} median {s[len(s)//2]:.3f} p90 {s[int(len(s)*0.9)]:.3f} max {s[-1]:.3f} mean {sum(w)/len(w):.3f} | " + " ".join(f"{x:.2f}" for x in w), flush=True)
st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
for _ in range(10):
    st(images)
show("eager, gc on ", windows(st))
gc.collect(); gc.disable()
show("eager, gc off", windows(st))
gc.enable()
show("eager, gc on ", windows(st))
stg = afdm.TrainStep(model, diff, lr=3e-4, graph=True)
for _ in range(5):
    stg(images)
show("graph        ", windows(stg))
