"""3x3 forward / dgrad / wgrad per Config-D layer shape (B = 256): the f16x2 kernels (csrc/h2.hip, h2_wgrad.hip: two fp16 pieces,
online scaling) against round 2's bf16x3 kernels (csrc/bf3.hip, bf3_wgrad.hip), microseconds per launch (weights ready; wgrad =
kernel + fold) and the per-step totals over the layers the rule gives to the direct forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, afdm, bench
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
shapes = sorted(set(bench.CONV3), key=lambda t: (-t[2], t[0], t[1]))
tot = {k: 0.0 for k in ("hf", "bf", "hd", "bd", "hw", "bw")}
for (ci, co, S) in shapes:
    if ci < 32:
        continue
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    u = torch.empty(16 * ci * co, device=dev)
    r = {}
    for tag, m_dir, m_wg in (("h", 76, 78), ("b", 77, 79)):
        L.afd_debug_conv_path(m_dir); L.afd_debug_conv_path(m_wg)
        kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
        r[tag + "f"] = r[tag + "d"] = r[tag + "w"] = float("nan")
        if kinds & 1:
            L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, kinds, s)
            r[tag + "f"] = bench.ev_time(lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, kinds, s), reps=10)
        if kinds & 2:
            L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, kinds, s)
            r[tag + "d"] = bench.ev_time(lambda: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 1, kinds, s), reps=10)
        if L.afd_conv_wgrad_form(B, ci, co, S, S, 3) in (2, 4):
            ws = torch.empty(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4 + 1, device=dev)
            f = lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s)
            f(); r[tag + "w"] = bench.ev_time(f, reps=10)
    L.afd_debug_conv_path(76); L.afd_debug_conv_path(78)
    fl = 2.0 * B * S * S * ci * co * 9
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} x{cnt}: fwd bf16x3 {r['bf']*1e3:7.1f} f16x2 {r['hf']*1e3:7.1f} ({fl/r['hf']/1e9:5.0f} TF) | "
          f"dgrad {r['bd']*1e3:7.1f} -> {r['hd']*1e3:7.1f} | wgrad {r['bw']*1e3:7.1f} -> {r['hw']*1e3:7.1f} ({fl/r['hw']/1e9:5.0f} TF)  us", flush=True)
    for k in tot:
        if r[k] == r[k]:
            tot[k] += cnt * r[k]
print("per step (ms), direct-form layers only: fwd bf16x3 %.3f f16x2 %.3f | dgrad %.3f -> %.3f | wgrad %.3f -> %.3f" %
      (tot["bf"], tot["hf"], tot["bd"], tot["hd"], tot["bw"], tot["hw"]))
# ---- the small maps: f16x2 split-K kernel (conv_h2_sk, afd_debug_conv_path 74) against round 1's fp32 Winograd split-K kernel (75)
tot = {k: 0.0 for k in ("sf", "wf", "sd", "wd")}
for (ci, co, S) in shapes:
    if ci < 32 or S > 8:
        continue
    cnt = bench.CONV3.count((ci, co, S))
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x)
    u = torch.empty(16 * ci * co, device=dev)
    r = {}
    L.afd_debug_conv_path(74)
    k74 = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
    L.afd_debug_conv_path(75)
    k75 = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
    if k74 == k75:
        L.afd_debug_conv_path(74)
        continue                                                  # the rule gives this layer to the same kernel either way
    for tag, mode in (("s", 74), ("w", 75)):
        L.afd_debug_conv_path(mode)
        kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
        r[tag + "f"] = r[tag + "d"] = float("nan")
        if (k74 ^ k75) & 1:
            L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, kinds, s)
            r[tag + "f"] = bench.ev_time(lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, kinds, s), reps=10)
        if (k74 ^ k75) & 2:
            L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, kinds, s)
            r[tag + "d"] = bench.ev_time(lambda: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 1, kinds, s), reps=10)
    L.afd_debug_conv_path(74)
    print(f"{ci:4d}->{co:4d} @{S:2d}x{S:<2d} x{cnt}: fwd wino_sk {r['wf']*1e3:6.1f} -> h2_sk {r['sf']*1e3:6.1f} | dgrad {r['wd']*1e3:6.1f} -> {r['sd']*1e3:6.1f} us", flush=True)
    for k in tot:
        if r[k] == r[k]:
            tot[k] += cnt * r[k]
print("small maps per step (ms): fwd wino_sk %.3f -> h2_sk %.3f | dgrad %.3f -> %.3f" % (tot["wf"], tot["sf"], tot["wd"], tot["sd"]))
