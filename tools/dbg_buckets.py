"""When does each data-parallel bucket become ready during backward?  One process, B = 2: GradAllReduce with its collective
replaced by a recorder (world forced to 2), prints (bucket, flushes so far, launches so far, backward finished?)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm
from afdm import ops
dev = torch.device("cuda:0")
F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
afdm.set_seed(42)
model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=F_SET, device=dev, variant=3).to(dev)
diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
step = afdm.TrainStep(model, diff, lr=3e-4, graph=False, distributed=True)
ddp = step.ddp
ddp.world = 2
log = []
state = {"flush": 0, "done": False}
orig_flush = ops.flush_wgrads
def flush(final=False):
    state["flush"] += 1
    return orig_flush(final)
ops.flush_wgrads = flush
def fake_launch(k):
    if k in ddp._launched:
        return
    ddp._launched.add(k)
    log.append((k, state["flush"], state["done"], len(ops._GradMode.pending), sum(len(f) for f in ops._GradMode.folds)))
ddp._launch = fake_launch
orig_done = ddp.backward_done
def bdone():
    state["done"] = True
    orig_done()
ddp.backward_done = bdone
ddp.finish = lambda: (setattr(ddp, "overlapped_last_step", ddp._in_backward), 0.5)[1]
x = torch.randn(B, 3, 32, 32, device=dev)
names = {id(p): n for n, p in model.named_parameters()}
step(x)
print("slices", ddp.slices, "overlapped", ddp.overlapped_last_step)
for k, fl, done, pend, nf in log:
    print(f"bucket {k}: started at flush {fl}, backward done={done}, pending {pend}, queued folds {nf}")
left = [names[i] for i in ddp._bucket_of if i not in ddp._seen]
print("never reported:", left[:10])
