"""Interleaved windows of the eager B=256 train step with the weight-gradient stream at different priorities."""
import sys, os, math, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); B = 256
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
steps = {}
for pr in [int(a) for a in sys.argv[1:]] or [0, 1, -1]:
    os.environ["AFD_WGRAD_PRIO"] = str(pr if pr <= 0 else 0)
    os.environ["AFD_MAIN_PRIO"] = "1" if pr in (9, 19) else "0"     # 9: the main chain on a high-priority stream, side normal; 10 / 19: hipGraph replay without / with
    st = afdm.TrainStep(model, diff, lr=3e-4, graph=pr >= 10)
    for _ in range(10):
        st(images)
    steps[pr] = st
gc.collect(); gc.disable()
res = {k: [] for k in steps}
for rnd in range(6):
    for k, st in steps.items():
        for _ in range(3):
            st(images)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            st(images)
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / 40 * 1e3)
for k, v in res.items():
    print(f"side priority {k}: " + " ".join(f"{x:.3f}" for x in v) + f"  median {sorted(v)[len(v)//2]:.3f}", flush=True)
