"""Per-launch time of the fused token-chain kernels of the attention blocks (csrc/tok.hip) at the Config-D shapes,
B = 256, next to the unfused chain (LayerNorm, 1x1 convolutions, GELU) they replace.  Run on the MI355X box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import afdm
import bench
from afdm import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
tot = {"hf": 0, "tf": 0, "tb": 0, "hb": 0, "uf": 0, "ub": 0}
for name, (C, S) in [("sa1", (64, 16)), ("sa2", (128, 8)), ("sa3", (128, 4)), ("sa4", (64, 8)), ("sa5", (32, 16)), ("sa6", (32, 32))]:
    HW = S * S
    x = torch.randn(B, C, S, S, device=dev)
    mk = lambda *sh: torch.randn(*sh, device=dev) * 0.1
    g1, b1_, w_in, b_in = 1 + mk(C), mk(C), mk(3 * C, C), mk(3 * C)
    wo, bo, g2, be2, w1, b1, w2, b2 = mk(C, C), mk(C), 1 + mk(C), mk(C), mk(C, C), mk(C), mk(C, C), mk(C)
    h, qkv, st = torch.empty_like(x), torch.empty(B, 3 * C, S, S, device=dev), torch.empty(B, HW, 2, device=dev)
    a, f, u, g, out = (torch.empty_like(x) for _ in range(5))
    st2 = torch.empty(B, HW, 2, device=dev)
    att, dout = torch.randn_like(x), torch.randn_like(x)
    du, df, da, datt, dh, dx = (torch.empty_like(x) for _ in range(6))
    dqkv = torch.randn_like(qkv)
    t_hf = bench.ev_time(lambda: L.afd_tok_head_fwd(P(x), P(g1), P(b1_), P(w_in), P(b_in), P(h), P(st), P(qkv), B, C, HW, 1e-5, s), reps=10)
    t_tf = bench.ev_time(lambda: L.afd_tok_tail_fwd(P(att), P(x), P(wo), P(bo), P(g2), P(be2), P(w1), P(b1), P(w2), P(b2), P(a), P(st2), P(f), P(u), P(g), P(out), B, C, HW, 1e-5, s), reps=10)
    t_tb = bench.ev_time(lambda: L.afd_tok_tail_bwd(P(dout), P(u), P(a), P(st2), P(g2), P(w2), P(w1), P(wo), P(du), P(df), P(da), P(datt), B, C, HW, s), reps=10)
    t_hb = bench.ev_time(lambda: L.afd_tok_head_bwd(P(dqkv), P(x), P(st), P(g1), P(w_in), P(da), P(dh), P(dx), B, C, HW, s), reps=10)
    # the unfused chain on the same shapes (forward: LN, in_proj | out_proj+res, LN, FF1, GELU, FF2+res; backward: the dgrads,
    # GELU', the two LayerNorm dx; weight gradients excluded on both sides)
    as1 = lambda w: w.reshape(w.shape[0], w.shape[1], 1, 1)
    def unf_fwd():
        L.afd_layernorm_c_fwd(P(x), P(h), P(st), B, C, HW, 1e-5, P(g1), P(b1_), s)
        L.afd_conv_fwd(P(h), P(w_in), P(b_in), None, P(qkv), B, C, 3 * C, S, S, 1, 0, s)
        L.afd_conv_fwd(P(att), P(wo), P(bo), P(x), P(a), B, C, C, S, S, 1, 0, s)
        L.afd_layernorm_c_fwd(P(a), P(f), P(st2), B, C, HW, 1e-5, P(g2), P(be2), s)
        L.afd_conv_fwd(P(f), P(w1), P(b1), None, P(u), B, C, C, S, S, 1, 0, s)
        L.afd_gelu_fwd(P(u), P(g), x.numel(), s)
        L.afd_conv_fwd(P(g), P(w2), P(b2), P(a), P(out), B, C, C, S, S, 1, 0, s)
    def unf_bwd():
        L.afd_conv_dgrad(P(dout), P(w2), P(du), B, C, C, S, S, 1, s)
        L.afd_gelu_bwd(P(u), P(du), P(du), x.numel(), s)
        L.afd_conv_dgrad(P(du), P(w1), P(df), B, C, C, S, S, 1, s)
        L.afd_layernorm_c_bwd(P(a), P(df), P(st2), B, C, HW, P(g2), P(da), P(dout), None, None, None, 0, s)
        L.afd_conv_dgrad(P(da), P(wo), P(datt), B, C, C, S, S, 1, s)
        L.afd_conv_dgrad(P(dqkv), P(w_in), P(dh), B, C, 3 * C, S, S, 1, s)
        L.afd_layernorm_c_bwd(P(x), P(dh), P(st), B, C, HW, P(g1), P(dx), P(da), None, None, None, 0, s)
    t_uf = bench.ev_time(unf_fwd, reps=10)
    t_ub = bench.ev_time(unf_bwd, reps=10)
    e = 4.0 * B * C * HW                                  # bytes of one C-channel activation tensor
    px = B * HW
    fl_c = 2.0 * px * C * C
    print(f"{name} C={C:3d} {S:2d}x{S:<2d}: head fwd {t_hf*1e3:6.1f} us ({5*e/t_hf/1e6:5.0f} GB/s, {3*fl_c/t_hf/1e9:5.1f} TF) | tail fwd {t_tf*1e3:6.1f} us "
          f"({7*e/t_tf/1e6:5.0f} GB/s, {3*fl_c/t_tf/1e9:5.1f} TF) | tail bwd {t_tb*1e3:6.1f} us ({7*e/t_tb/1e6:5.0f} GB/s) | head bwd {t_hb*1e3:6.1f} us "
          f"({7*e/t_hb/1e6:5.0f} GB/s) || unfused fwd {t_uf*1e3:6.1f} bwd {t_ub*1e3:6.1f} us")
    for k, v in zip(tot, (t_hf, t_tf, t_tb, t_hb, t_uf, t_ub)):
        tot[k] += v
print("totals (us): fused fwd %.0f (head %.0f + tail %.0f), fused bwd %.0f (tail %.0f + head %.0f) | unfused fwd %.0f, bwd %.0f" % (
    (tot["hf"] + tot["tf"]) * 1e3, tot["hf"] * 1e3, tot["tf"] * 1e3, (tot["tb"] + tot["hb"]) * 1e3, tot["tb"] * 1e3, tot["hb"] * 1e3,
    tot["uf"] * 1e3, tot["ub"] * 1e3))
