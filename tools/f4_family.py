"""Runs the filtered-GELU family of the Config-D train step in isolation (the 22 sites, forward and backward through the
fused GroupNorm prologue, B = 256) `reps` times, for a rocprofv3 --pmc pass over this process.

  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ... --kernel-trace --output-format csv -d out -o f4 -- python3 tools/f4_family.py 3
"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afdm
import bench
from afdm import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
tk = ops.Taps(afdm.circularLowpassKernel(math.pi / 2, 3, 2))
P = lambda t: t.data_ptr()
bufs = {}
for (C, S) in bench.ACT_SITES:
    if (C, S) in bufs:
        continue
    x = torch.randn(B, C, S, S, device=dev); y = torch.empty_like(x); dv = torch.empty_like(x)
    st = torch.zeros(B, 2, device=dev); st[:, 1] = 1
    bufs[(C, S)] = (x, y, dv, st, torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.empty(B * C * 2, device=dev))
torch.cuda.synchronize()
for _ in range(reps):
    for (C, S) in bench.ACT_SITES:
        x, y, dv, st, g, be, part = bufs[(C, S)]
        L.afd_filt_act_fwd(P(x), P(y), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, s)
        L.afd_filt_act_bwd(P(x), P(y), P(dv), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, P(part), s)
torch.cuda.synchronize()
print(f"f4: {reps} x {len(bench.ACT_SITES)} sites fwd + bwd")
