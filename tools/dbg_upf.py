import sys, math, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_golden, rel_l2
import afdm
from afdm import ops
dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.asarray(a))
g = load_golden('blocks.npz')
name = 'upF_8_4'
fs = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
mod = afdm.Up_F(8, 4, f_settings=fs)
p = name + '.sd.'
mod.load_state_dict({k[len(p):]: T(g[k]) for k in g.files if k.startswith(p)})
mod = mod.to(dev)
ins = []
j = 0
while f"{name}.in{j}" in g.files:
    ins.append(T(g[f"{name}.in{j}"]).to(dev).requires_grad_(True)); j += 1
rec = []
orig = ops.GroupNormFiltAct.forward
def fwd(ctx, *a):
    y = orig(ctx, *a)
    rec.append((y, y.clone()))
    return y
ops.GroupNormFiltAct.forward = staticmethod(fwd)
y = mod(*ins)
torch.cuda.synchronize()
params = list(mod.named_parameters())
grads = torch.autograd.grad(y, ins + [q for _, q in params], T(g[f"{name}.dy"]).to(dev), allow_unused=True)
torch.cuda.synchronize()
for i, (a, b) in enumerate(rec):
    print('F4 output', i, tuple(a.shape), 'changed after backward:', not torch.equal(a, b), float((a - b).abs().max()))
for (kn, _), gr in zip(params, grads[len(ins):]):
    print(kn, rel_l2(gr.cpu(), g[f"{name}.dsd.{kn}"]))
