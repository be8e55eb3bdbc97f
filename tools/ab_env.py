"""Same-box, same-process A/B of the B=256 train step over values of an environment variable that the step reads on every
call (AFD_FOLD_EVERY, ...), interleaved.   python tools/ab_env.py VAR v1 v2 [v3 ...] [--batch 256] [--rounds 4]"""
import sys, os, math, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
args = sys.argv[1:]
B, R = 256, 4
if "--batch" in args:
    i = args.index("--batch"); B = int(args[i + 1]); del args[i:i + 2]
if "--rounds" in args:
    i = args.index("--rounds"); R = int(args[i + 1]); del args[i:i + 2]
var, vals = args[0], args[1:]
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
for _ in range(10):
    st(images)
res = {v: [] for v in vals}
gc.collect(); gc.disable()          # (a full collection costs ~30 ms: one window in forty; tools/jitter_probe.py)
for rnd in range(R):
    for v in vals:
        os.environ[var] = v
        for _ in range(3):
            st(images)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            st(images)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 40 * 1e3)
for v in vals:
    print(f"{var}={v}: " + " ".join(f"{r:.3f}" for r in res[v]) + f"  best {min(res[v]):.3f} median {sorted(res[v])[len(res[v]) // 2]:.3f} ms/step", flush=True)
