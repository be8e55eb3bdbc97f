import sys, os, math, io, contextlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
fs = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi/2, "omega_c_up": math.pi/2}
with contextlib.redirect_stdout(io.StringIO()):
    model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=fs, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
step = afdm.TrainStep(model, diff, lr=3e-4, graph=False)
x = torch.rand(64, 3, 32, 32, device=dev) * 2 - 1
for _ in range(2): step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(x)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::zeros_like", "aten::zero_", "aten::fill_", "aten::_to_copy", "aten::mul", "aten::detach"):
        st = [s for s in (e.stack or []) if "afdm" in s or "aliasfree" in s or "autograd" in s][:2]
        shp = str(e.input_shapes)[:60]
        cnt[(e.name, tuple(st), shp)] += 1
for (name, st, shp), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]:
    print(c, name, shp, " | ".join(s.split("/")[-1] for s in st))
