"""3x3 weight gradient (f16x2 kernels + fold) per Config-D layer shape at B = 256, one line (for AFD_LIBPATH variant libraries)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
out, tot = [], 0.0
for (ci, co, S) in shapes:
    if ci < 32:
        continue
    if L.afd_conv_wgrad_form(B, ci, co, S, S, 3) != 4:
        continue
    x = torch.randn(B, ci, S, S, device=dev); y = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    ws = torch.empty(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4 + 1, device=dev)
    f = lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
    f(); t = bench.ev_time(f, reps=20)
    tot += t * bench.CONV3.count((ci, co, S))
    out.append(f"{ci}>{co}@{S}:{t*1e3:.1f}")
print(" ".join(out), f"| total {tot:.3f} ms", flush=True)
