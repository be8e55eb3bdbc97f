"""1x1 convolution (token projections of the attention blocks) per shape: time, GB/s (algorithmic), TF."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
if len(sys.argv) > 1:
    L.afd_debug_conv_path(int(sys.argv[1]))           # 0 by rule, 1 big tile, 2 split-K small tile, 9 streaming kernel off
shapes = []
for (C, S) in bench.ATTN:
    shapes += [(C, 3 * C, S, "in_proj"), (C, C, S, "out/ff")]
tot = [0, 0, 0]
for (ci, co, S, name) in shapes:
    cnt = 1 if name == "in_proj" else 3
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 1, 1, device=dev) * 0.1; bias = torch.randn(co, device=dev)
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(bias)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 1) // 4, 1), device=dev)
    tf = bench.ev_time(lambda: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, ci, co, S, S, 1, 0, s), reps=10)
    td = bench.ev_time(lambda: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 1, s), reps=10)
    tw = bench.ev_time(lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), B, ci, co, S, S, 1, 0, ws.data_ptr(), s), reps=10)
    by = 4.0 * B * S * S * (ci + co)
    for i, t in enumerate((tf, td, tw)):
        tot[i] += t * cnt
    print(f"{name:8s} {ci:4d}->{co:4d} @{S:2d}x{S:<2d} x{cnt} | fwd {tf*1e3:7.1f}us {by/tf/1e6:7.0f} GB/s | dgrad {td*1e3:7.1f}us {by/td/1e6:7.0f} GB/s | wgrad+dbias {tw*1e3:7.1f}us {by/tw/1e6:7.0f} GB/s")
print("per-step totals (ms): fwd %.3f dgrad %.3f wgrad %.3f" % tuple(tot))
