"""Winograd vs direct 3x3 wgrad per Config-D layer shape (B=256): microseconds (kernel + slab reduce)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
tot = [0.0, 0.0, 0.0]
print(f"{'shape':>20} {'n':>2} {'GFLOP':>6} | {'direct':>8} {'wino':>8} {'auto':>8}  (us)")
for (ci, co, S) in shapes:
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); y = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    ts = []
    for mode in (97, 98, 96):
        L.afd_debug_conv_path(mode)
        ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
        ts.append(bench.ev_time(lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s), reps=10))
    L.afd_debug_conv_path(96)
    for i in range(3): tot[i] += ts[i] * cnt
    fl = 2.0 * B * S * S * ci * co * 9
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} {cnt:2d} {fl / 1e9:6.2f} | {ts[0] * 1e3:8.1f} {ts[1] * 1e3:8.1f} {ts[2] * 1e3:8.1f}")
print("wgrad per step: direct %.3f ms, winograd %.3f ms, auto %.3f ms" % tuple(tot))
