"""3x3 weight gradient per Config-D layer shape (B = 256): the bf16x3 kernel (csrc/bf3_wgrad.hip) against the fp32 Winograd
form, microseconds per call (kernel + slab reduction) and the per-step totals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
tot = {"w": 0.0, "b": 0.0, "a": 0.0}
for (ci, co, S) in shapes:
    if ci < 32:
        continue
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); dy = torch.randn(B, co, S, S, device=dev); dw = torch.empty(co, ci, 3, 3, device=dev)
    r = {}
    for tag, mode in (("w", 85), ("b", 86), ("a", 84)):
        L.afd_debug_conv_path(mode)
        ws = torch.empty(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4 + 1, device=dev)
        f = lambda: L.afd_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
        f(); r[tag] = bench.ev_time(f, reps=10)
    L.afd_debug_conv_path(84)
    fl = 2.0 * B * S * S * ci * co * 9
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} x{cnt}: wgrad wino {r['w']*1e3:7.1f}  bf3 {r['b']*1e3:7.1f} ({fl/r['b']/1e9:5.0f} TF)  rule {r['a']*1e3:7.1f} us")
    for k in tot:
        tot[k] += cnt * r[k]
print("per step (ms): wgrad wino %.3f  bf3 (where covered) %.3f  rule %.3f" % (tot["w"], tot["b"], tot["a"]))
