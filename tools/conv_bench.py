"""Per-layer conv timing (Config D shapes, B=256): fwd / dgrad / wgrad TFLOP/s for each distinct shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afdm
import bench

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2] if len(sys.argv) > 2 else None
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
print(f"{'shape':>22} {'count':>5} {'GFLOP':>7} | {'fwd us':>8} {'TF':>6} | {'dgrad us':>8} {'TF':>6} | {'wgrad us':>8} {'TF':>6}")
tot = [0, 0, 0]
for (ci, co, S) in shapes:
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
    fl = 2.0 * B * S * S * ci * co * 9
    tf = bench.ev_time(lambda: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s), reps=10)
    td = bench.ev_time(lambda: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s), reps=10)
    tw = bench.ev_time(lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s), reps=10)
    for i, t in enumerate((tf, td, tw)):
        tot[i] += t * cnt
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} B{B:<4d} {cnt:5d} {fl/1e9:7.2f} | {tf*1e3:8.1f} {fl/tf/1e9:6.1f} | {td*1e3:8.1f} {fl/td/1e9:6.1f} | {tw*1e3:8.1f} {fl/tw/1e9:6.1f}")
print("per-step totals (ms): fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
