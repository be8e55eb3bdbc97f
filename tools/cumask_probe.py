"""Experiment: does partitioning the CUs between the main stream and the weight-gradient stream (hipExtStreamCreateWithCUMask)
beat sharing them?  Diagnostic only."""
import ctypes, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc={rc}")
    return torch.cuda.ExternalStream(st.value)

F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
images = torch.randn(256, 3, 32, 32, device=dev)
ALL = (1 << 256) - 1

def run(tag, main_bits, side_bits):
    main = masked_stream(main_bits) if main_bits is not None else torch.cuda.current_stream()
    st = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
    if side_bits is not None:
        st.wgrad_stream = [masked_stream(side_bits)]
    with torch.cuda.stream(main):
        for _ in range(8):
            st(images)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            st(images)
        torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms/step", flush=True)

run("baseline (no masks)", None, None)
for n_side in (32, 64, 96, 128):
    lo = (1 << n_side) - 1                       # the first n_side mask bits
    run(f"side = {n_side} mask bits, main = all", None, lo)
    run(f"side = {n_side} mask bits, main = the other {256 - n_side}", ALL ^ lo, lo)
# interleaved partition: every 4th bit to the side stream
ev = sum(1 << i for i in range(0, 256, 4))
run("side = every 4th bit (64), main = all", None, ev)
run("side = every 4th bit (64), main = the rest", ALL ^ ev, ev)
