"""Median over windows of the B=256 train step of ONE TrainStep in this process (configured by environment variables; run the
variants as separate processes, alternating).   python tools/step_median.py [--graph | --lanes] [--windows 30]"""
import sys, os, math, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); B = int(os.environ.get("AFD_B", 256))
graph = "lanes" if "--lanes" in sys.argv else "--graph" in sys.argv
W = int(sys.argv[sys.argv.index("--windows") + 1]) if "--windows" in sys.argv else 30
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
st = afdm.TrainStep(model, diff, lr=3e-4, graph=graph)
for _ in range(30):
    st(images)
gc.collect(); gc.disable()
w = []
for _ in range(W):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        loss = st(images)
    torch.cuda.synchronize()
    w.append((time.perf_counter() - t0) / 20 * 1e3)
s_ = sorted(w)
mode = "lanes" if graph == "lanes" else "graph" if graph else "eager"
extra = f" {st.lanes_counts}" if graph == "lanes" else ""
print(f"{mode} {os.environ.get('TAG', '')}{extra}: median {s_[len(s_)//2]:.3f} min {s_[0]:.3f} p90 {s_[int(len(s_)*0.9)]:.3f} loss {float(loss):.4f}", flush=True)
