"""3x3 forward / dgrad per Config-D layer shape (B = 256): the direct bf16x3 kernel (csrc/bf3.hip) against the fp32 Winograd
kernels, microseconds per launch (weights ready) and the per-step totals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
tot = {"wf": 0, "bf": 0, "af": 0, "wd": 0, "bd": 0, "ad": 0}
for (ci, co, S) in shapes:
    if ci < 32:
        continue
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x)
    u = torch.empty(16 * ci * co, device=dev)
    r = {}
    for tag, mode in (("w", 81), ("b", 82), ("a", 80)):
        L.afd_debug_conv_path(mode)
        nf = L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 0); nd = L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 1)
        r[tag + "f"] = r[tag + "d"] = float("nan")
        if nf:
            L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
            r[tag + "f"] = bench.ev_time(lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s), reps=10)
        if nd:
            L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s)
            r[tag + "d"] = bench.ev_time(lambda: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 1, L.afd_conv3x3_weight_kinds(B, ci, co, S, S), s), reps=10)
    L.afd_debug_conv_path(80)
    fl = 2.0 * B * S * S * ci * co * 9
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} x{cnt}: fwd wino {r['wf']*1e3:7.1f}  bf3 {r['bf']*1e3:7.1f} ({fl/r['bf']/1e9:5.0f} TF)  rule {r['af']*1e3:7.1f} | "
          f"dgrad wino {r['wd']*1e3:7.1f}  bf3 {r['bd']*1e3:7.1f}  rule {r['ad']*1e3:7.1f}  us")
    for k in tot:
        if r[k] == r[k]:
            tot[k] += cnt * r[k]
print("per step (ms): fwd wino %.3f bf3 %.3f rule %.3f | dgrad wino %.3f bf3 %.3f rule %.3f" % (tot["wf"], tot["bf"], tot["af"], tot["wd"], tot["bd"], tot["ad"]))
