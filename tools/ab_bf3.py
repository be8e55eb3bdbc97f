"""Same-box A/B of the B=256 train step between two afd_debug_conv_path settings, interleaved.
   python tools/ab_bf3.py [B] [modeA modeB]     default 80 81: bf16x3 forward / dgrad by rule vs fp32 Winograd only; 84 85: the wgrad"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0"); L = afdm.lib()
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MODES = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (80, 81)
afdm.set_seed(42)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(B, 3, 32, 32, device=dev)
steps = {}
for mode in MODES:
    L.afd_debug_conv_path(mode)
    afdm.set_seed(42)
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
    steps[mode] = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    for _ in range(8):
        steps[mode](images)
torch.cuda.synchronize()
for rnd in range(4):
    for mode in MODES:
        L.afd_debug_conv_path(mode)
        st = steps[mode]
        for _ in range(3):
            st(images)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            st(images)
        torch.cuda.synchronize()
        print(f"round {rnd} mode {mode}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)
L.afd_debug_conv_path(MODES[0])
