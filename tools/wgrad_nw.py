"""wgrad: 4 vs 8 waves per workgroup, per 3x3 layer shape (B=256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
tot = {32: 0.0, 33: 0.0}
for (ci, co, S) in shapes:
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); y = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
    r = {}
    for mode in (32, 33):
        L.afd_debug_conv_path(mode)
        r[mode] = bench.ev_time(lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s), reps=10)
        tot[mode] += r[mode] * cnt
    print(f"{ci:4d}->{co:4d} @{S:2d} x{cnt}:  4 waves {r[32]*1e3:7.1f} us   8 waves {r[33]*1e3:7.1f} us")
L.afd_debug_conv_path(34)
print("per-step totals (ms): 4 waves %.3f  8 waves %.3f" % (tot[32], tot[33]))
