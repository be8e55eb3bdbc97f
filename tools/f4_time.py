"""Times the filtered-GELU family (22 Config-D sites, B = 256): forward and backward, per plane size and in total."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
from afdm import ops
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tk = ops.Taps(afdm.circularLowpassKernel(math.pi / 2, 3, 2)); P = lambda t: t.data_ptr()
tf = tb = 0; seen = {}
for (C, S) in bench.ACT_SITES:
    if (C, S) not in seen:
        x = torch.randn(B, C, S, S, device=dev); y = torch.empty_like(x); dv = torch.empty_like(x)
        st = torch.zeros(B, 2, device=dev); st[:, 1] = 1; g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev); part = torch.empty(B * C * 2, device=dev)
        a = bench.ev_time(lambda: L.afd_filt_act_fwd(P(x), P(y), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, s), reps=10)
        b = bench.ev_time(lambda: L.afd_filt_act_bwd(P(x), P(y), P(dv), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, P(part), s), reps=10)
        seen[(C, S)] = (a, b)
        e = B * C * S * S
        print(f"  C={C:3d} S={S:2d}: fwd {a*1e3:6.1f} us ({8*e/a/1e6:5.0f} GB/s)  bwd {b*1e3:6.1f} us ({12*e/b/1e6:5.0f} GB/s)")
    tf += seen[(C, S)][0]; tb += seen[(C, S)][1]
e = sum(B * C * S * S for C, S in bench.ACT_SITES)
print(f"F4 fwd {tf:.3f} ms ({8*e/tf/1e6:.0f} GB/s)  bwd {tb:.3f} ms ({12*e/tb/1e6:.0f} GB/s)")
