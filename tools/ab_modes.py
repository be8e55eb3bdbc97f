"""Same-box A/B of the B=256 train step between two sets of afd_debug_conv_path settings, interleaved; each set has its own
TrainStep (weight images are built per setting).   python tools/ab_modes.py "76,78" "77,79" [--batch 256]"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
args = sys.argv[1:]
B = 256
if "--batch" in args:
    i = args.index("--batch"); B = int(args[i + 1]); del args[i:i + 2]
SETS = [tuple(int(v) for v in a.split(",")) for a in args]
dev = torch.device("cuda:0"); L = afdm.lib()
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
steps = {}
def apply(ms):
    for m in ms:
        L.afd_debug_conv_path(m)
for ms in SETS:
    apply(ms)
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
    steps[ms] = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    for _ in range(8):
        steps[ms](images)
torch.cuda.synchronize()
res = {ms: [] for ms in SETS}
for rnd in range(4):
    for ms in SETS:
        apply(ms)
        st = steps[ms]
        for _ in range(3):
            st(images)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            st(images)
        torch.cuda.synchronize()
        res[ms].append((time.perf_counter() - t0) / 40 * 1e3)
for ms in SETS:
    print(f"modes {ms}: " + " ".join(f"{r:.3f}" for r in res[ms]) + f"  best {min(res[ms]):.3f} ms/step", flush=True)
apply(SETS[0])
