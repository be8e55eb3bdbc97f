#!/usr/bin/env python3
"""bench.py -- train-step images/sec (+ sample images/sec) of the Config-D UNet on MI355X.

Workload (BASELINE.json configs[2]/[3]): CIFAR-10-shaped synthetic batches 3x32x32, UNet variant=3
(filtered GELU + filtered resampling, f_settings {3, 2, pi/2, pi/2}), 256 images per GPU, T=1000,
lr 3e-4, fp32.  One "step" = the reference's per-batch body (ddpm_utils.py:499-507): timestep draw,
noise_images, UNet forward, MSE, backward, [gradient all-reduce], AdamW.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel, timed live with HIP events
on the launch stream), `kernels` (every kernel family timed the same way), `cpu_baseline`
(the CPU oracle's train step on the host cores, bounded sample), `sample` (Diffusion.sample).
"""
import argparse
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

F_SET = {"kernel_size": 3, "kaiser_beta": 2, "omega_c_down": math.pi / 2, "omega_c_up": math.pi / 2}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: bf16 MFMA dense peak (the bf16x3 kernels issue 6 bf16 products per fp32 product)

# Config D layer table: (Cin, Cout, HW side) of every 3x3 conv, forward order
CONV3 = ([(3, 32, 32), (32, 32, 32)] + [(32, 32, 16)] * 2 + [(32, 64, 16), (64, 64, 16)] + [(64, 64, 8)] * 2 +
         [(64, 128, 8), (128, 128, 8)] + [(128, 128, 4)] * 4 + [(128, 256, 4), (256, 256, 4), (256, 256, 4), (256, 256, 4),
         (256, 128, 4), (128, 128, 4)] + [(256, 256, 8)] * 2 + [(256, 128, 8), (128, 64, 8)] + [(128, 128, 16)] * 2 +
         [(128, 64, 16), (64, 32, 16)] + [(64, 64, 32)] * 2 + [(64, 32, 32), (32, 32, 32)])
# filtered-GELU sites (C, side)
ACT_SITES = ([(32, 32)] + [(32, 16)] * 2 + [(64, 16)] + [(64, 8)] * 2 + [(128, 8)] + [(128, 4)] * 3 + [(256, 4), (256, 4), (128, 4)] +
             [(256, 8)] * 2 + [(128, 8)] + [(128, 16)] * 2 + [(64, 16)] + [(64, 32)] * 2 + [(32, 32)])
ATTN = [(64, 16), (128, 8), (128, 4), (64, 8), (32, 16), (32, 32)]      # (C, side), 4 heads


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """CPU threads this process may really use (cgroup / affinity aware), capped at 16 = one GPU's share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def ev_time(fn, reps=5, warm=2):
    """Average device time of fn() in ms, HIP events on the current (= launch) stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


GN_PLAIN = [(32, 32), (64, 16), (128, 8), (128, 4), (256, 4), (256, 4), (128, 4), (64, 8), (32, 16), (32, 32)]   # GroupNorm sites without F4 (C, side)
RESAMPLE = [(32, 32), (64, 16), (128, 8)]                # block-level F3 inputs (C, side); F2 runs the same shapes in reverse
EMB = [64, 128, 128, 64, 32, 32]                         # time-embedding Linear widths (256 -> C)
N_PARAMS = 5897155


def kernel_table(dev, B, mode="train"):
    """Times EVERY kernel family of one Config-D train step (mode "train": forward + backward + optimiser) or of one denoise
    step of Diffusion.sample (mode "sample": the UNet forward in its inference form -- no saved activations in the token
    chains -- plus the DDPM update; ddpm_models.py:352-386) over its real shapes, each through the C-ABI entry point the
    autograd shells call, with HIP events on the launch stream.  Per family: launches, ms per step, algorithmic flops or
    bytes, the bound it is priced against and the fraction reached.

    `frac` of the matrix-pipe families is the EXECUTED fraction: the time the pipe would need, at the peak of the
    instruction each launch really issues, for the products it issues (bf16x3: 6 bf16 products per fp32 product against
    the 2500 TFLOP/s bf16 peak; Winograd: 16/36 of the products against the 157.3 TFLOP/s fp32 peak; direct fp32 MFMA:
    all of them against 157.3) over the measured time -- never above 1.  The algorithmic direct-form rate against the
    fp32 matrix peak is kept beside it as `frac_algorithmic_vs_fp32_peak` (it exceeds 1 where bf16x3 pays off)."""
    import afdm
    from afdm import ops
    L, s = afdm.lib(), torch.cuda.current_stream().cuda_stream
    train = mode == "train"
    rows = {}
    P = lambda t: None if t is None else t.data_ptr()

    def add(name, ms, flops=0.0, bytes_=0.0, launches=1, bound=None, **extra):
        r = rows.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "bound": bound})
        r["ms"] += ms; r["flops"] += flops; r["bytes"] += bytes_; r["launches"] += launches
        for k, v in extra.items():
            r[k] = r.get(k, 0) + v

    def cached(cache, key, fn):
        if key not in cache:
            cache[key] = fn()
        return cache[key]

    # ---- 3x3 convolutions (F5) -----------------------------------------------------------------------------------
    seen = {}

    def pipe_s(fl, form):
        """Seconds the matrix pipe needs AT PEAK for the multiplies the kernel form issues for `fl` algorithmic flops:
        f16x2 = 3 fp16 products per fp32 product on the fp16 MFMA (same 2500 TFLOP/s dense peak as bf16), bf16x3 = 6 bf16 products,
        Winograd = 16/36 of the products on the fp32 MFMA."""
        return {"h2": 3.0 * fl / (MFMA_BF16_PEAK_TF * 1e12), "bf3": 6.0 * fl / (MFMA_BF16_PEAK_TF * 1e12),
                "wino": 16.0 / 36.0 * fl / (MFMA_F32_PEAK_TF * 1e12), "direct": fl / (MFMA_F32_PEAK_TF * 1e12)}[form]

    for (ci, co, S) in CONV3:
        tf, td, tw, ff, fd, fw = cached(seen, (ci, co, S), lambda: conv_layer_times(L, s, dev, B, ci, co, S, fwd_only=not train))
        fl = 2.0 * B * S * S * ci * co * 9              # algorithmic (direct-form) flops of one pass
        add("conv3x3_fwd", tf, fl, bound="mfma", pipe_s=pipe_s(fl, ff), **{ff: 1})
        if not train:
            continue
        if ci > 3:
            add("conv3x3_dgrad", td, fl, bound="mfma", pipe_s=pipe_s(fl, fd), **{fd: 1})
        add("conv3x3_wgrad", tw, fl, bound="mfma", pipe_s=pipe_s(fl, fw), **{fw: 1})
    if train:
        # the weight images of all 3x3 layers (row scales + two-piece splits, or the Winograd transform): ONE batched call at the
        # start of every step (ops.WinoStepPlan); sampling builds them once per trajectory
        reqs = {}
        for i, (ci, co, S) in enumerate(CONV3):
            nf, nd = L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 0), L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 1)
            if nf or nd:
                reqs[i] = [torch.randn(co, ci, 3, 3, device=dev) * 0.05, nf, nd, L.afd_conv3x3_weight_kinds(B, ci, co, S, S)]
        plan = ops.WinoStepPlan(reqs)
        add("conv3x3_weight_images", ev_time(plan.launch), bytes_=sum(4.0 * 9 * r[0].shape[0] * r[0].shape[1] + r[1] + r[2] for r in reqs.values()),
            launches=2, bound="hbm")
    # ---- filtered GELU (F4) + the GroupNorm kernels around it (F6) ------------------------------------------------
    seen = {}
    tk = ops.Taps(afdm.circularLowpassKernel(math.pi / 2, 3, 2))

    def act_site(C, S):
        x = torch.randn(B, C, S, S, device=dev); y = torch.empty_like(x); dv = torch.empty_like(x); dx = torch.empty_like(x)
        st = torch.zeros(B, 2, device=dev); st[:, 1] = 1
        g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
        part = torch.empty(B * C * 2, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        if L.afd_filt_act_fwd_gn_supported(C, S, S, 3):
            # small samples: the statistics and GroupNorm's backward ride in the activation's own launches (as ops.py calls them)
            tf = ev_time(lambda: L.afd_filt_act_fwd_gn(P(x), P(y), B, C, S, S, 1e-5, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, s))
            tb = ev_time(lambda: L.afd_filt_act_bwd_gn(P(x), P(y), P(dx), None, B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, P(part), s)) if train else 0.0
            return tf, tb, None, None
        tf = ev_time(lambda: L.afd_filt_act_fwd(P(x), P(y), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, s))
        tg = ev_time(lambda: L.afd_groupnorm1_fwd(P(x), None, P(st), B, C, S * S, 1e-5, None, None, None, 0, None, s))
        if not train:
            return tf, 0.0, tg, 0.0
        tb = ev_time(lambda: L.afd_filt_act_bwd(P(x), P(y), P(dv), B, C, S, S, P(st), P(g), P(be), None, tk.ptr, tk.ptr, 3, None, P(part), s))
        ta = ev_time(lambda: L.afd_groupnorm1_bwd(P(x), P(dv), P(st), B, C, S * S, P(g), P(be), None, 0, P(dx), None, P(part), None, 1, P(dg), P(db), 1, s))
        return tf, tb, tg, ta
    for (C, S) in ACT_SITES:
        tf, tb, tg, ta = cached(seen, (C, S), lambda: act_site(C, S))
        e = float(B) * C * S * S
        add("filt_act_fwd_n3", tf, bytes_=8 * e, bound="hbm")
        if train:
            add("filt_act_bwd_n3", tb, bytes_=12 * e, bound="hbm")
        if tg is not None:
            add("groupnorm1_stats", tg, bytes_=4 * e, bound="hbm")
            if train:
                add("groupnorm1_bwd_apply", ta, bytes_=12 * e, bound="hbm")
    seen = {}

    def gn_site(C, S):
        x = torch.randn(B, C, S, S, device=dev); y = torch.empty_like(x); dx = torch.empty_like(x)
        st = torch.empty(B, 2, device=dev); g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
        emb = torch.randn(B, C, device=dev); demb = torch.empty(B, C, device=dev)
        part = torch.empty(B * C * 2, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        tf = ev_time(lambda: L.afd_groupnorm1_fwd(P(x), P(y), P(st), B, C, S * S, 1e-5, P(g), P(be), None, 0, P(emb), s))
        tb = ev_time(lambda: L.afd_groupnorm1_bwd(P(x), P(y), P(st), B, C, S * S, P(g), P(be), None, 0, P(dx), None, P(part), P(demb), 0, P(dg), P(db), 1, s)) if train else 0.0
        return tf, tb
    for (C, S) in GN_PLAIN:
        tf, tb = cached(seen, (C, S), lambda: gn_site(C, S))
        e = float(B) * C * S * S
        add("groupnorm1_fwd_full", tf, bytes_=8 * e, bound="hbm")
        if train:
            add("groupnorm1_bwd_full", tb, bytes_=12 * e, launches=1, bound="hbm")          # sample-resident: x, dy read once, dx written (+ the small partials)
    # ---- block-level filtered resampling (F2 / F3) and the concat copy --------------------------------------------
    for (C, S) in RESAMPLE:
        x = torch.randn(B, C, S, S, device=dev); y = torch.empty(B, C, S // 2, S // 2, device=dev)
        e_hi, e_lo = float(B) * C * S * S, float(B) * C * S * S / 4
        add("filt_down2_fwd", ev_time(lambda: L.afd_filt_down2_fwd(P(x), P(y), B, C, S, S, 0, 0, tk.ptr, 3, s)), bytes_=4 * (e_hi + e_lo), bound="hbm")
        add("filt_up2_fwd", ev_time(lambda: L.afd_filt_up2_fwd(P(y), P(x), B, C, S // 2, S // 2, 0, 0, tk.ptr, 3, s)), bytes_=4 * (e_hi + e_lo), bound="hbm")
        if train:
            add("filt_down2_bwd", ev_time(lambda: L.afd_filt_down2_bwd(P(y), P(x), B, C, S, S, 0, 0, tk.ptr, 3, s)), bytes_=4 * (e_hi + e_lo), bound="hbm")
            add("filt_up2_bwd", ev_time(lambda: L.afd_filt_up2_bwd(P(x), P(y), B, C, S // 2, S // 2, 0, 0, tk.ptr, 3, s)), bytes_=4 * (e_hi + e_lo), bound="hbm")
        # (forward only: the backward hands the skip's gradient on as a view of the concat gradient)
        add("concat_copy", ev_time(lambda: L.afd_copy_batched(P(x), P(x), B, C * S * S, 0, 0, s)), bytes_=8 * e_hi, launches=1, bound="hbm")
        del x, y
    # ---- attention blocks (F10): core + the fused token-wise chains + their parameter gradients --------------------
    for (C, S) in ATTN:
        Lq = S * S
        qkv = torch.randn(B, 3 * C, S, S, device=dev); o = torch.empty(B, C, S, S, device=dev)
        lse = torch.empty(B, 4, Lq, device=dev); dq = torch.empty_like(qkv); dl = torch.empty_like(lse)
        tf = ev_time(lambda: L.afd_attn_fwd(P(qkv), P(o), P(lse), B, 4, C // 4, Lq, s), reps=3, warm=1)
        fl = 4.0 * B * Lq * Lq * C
        add("attn_fwd", tf, fl, bound="mfma")
        if train:
            tb = ev_time(lambda: L.afd_attn_bwd(P(qkv), P(o), P(o), P(lse), P(dq), P(dl), B, 4, C // 4, Lq, s), reps=3, warm=1)
            add("attn_bwd", tb, 2.5 * fl, launches=2, bound="mfma")
        # token-wise chains
        x = torch.randn(B, C, S, S, device=dev)
        mk = lambda *sh: torch.randn(*sh, device=dev) * 0.1
        g1, b1_, w_in, b_in = 1 + mk(C), mk(C), mk(3 * C, C), mk(3 * C)
        wo, bo, g2, be2, w1, b1, w2, b2 = mk(C, C), mk(C), 1 + mk(C), mk(C), mk(C, C), mk(C), mk(C, C), mk(C)
        h, st = torch.empty_like(x), torch.empty(B, Lq, 2, device=dev)
        a, f, u, g, out = (torch.empty_like(x) for _ in range(5))
        st2 = torch.empty(B, Lq, 2, device=dev)
        e, flc = 4.0 * B * C * Lq, 2.0 * B * Lq * C * C
        if not train:
            # inference forms (ops.AttnHead / AttnTail under no_grad): nothing is saved for a backward -- head reads x, writes qkv;
            # tail reads att and x, writes out
            assert L.afd_tok_supported(C)
            add("tok_head_fwd", ev_time(lambda: L.afd_tok_head_fwd(P(x), P(g1), P(b1_), P(w_in), P(b_in), None, None, P(qkv), B, C, Lq, 1e-5, s)),
                3 * flc, 4 * e, bound="hbm")
            add("tok_tail_fwd", ev_time(lambda: L.afd_tok_tail_fwd(P(o), P(x), P(wo), P(bo), P(g2), P(be2), P(w1), P(b1), P(w2), P(b2), None, None, None, None,
                                                                    None, P(out), B, C, Lq, 1e-5, s)), 3 * flc, 3 * e, bound="hbm")
            del qkv, o, lse, dq, x, h, a, f, u, g, out
            continue
        du, df, da, datt, dh, dx = (torch.empty_like(x) for _ in range(6))
        if L.afd_tok_supported(C):
            add("tok_head_fwd", ev_time(lambda: L.afd_tok_head_fwd(P(x), P(g1), P(b1_), P(w_in), P(b_in), P(h), P(st), P(qkv), B, C, Lq, 1e-5, s)),
                3 * flc, 5 * e, bound="hbm")
            add("tok_tail_fwd", ev_time(lambda: L.afd_tok_tail_fwd(P(o), P(x), P(wo), P(bo), P(g2), P(be2), P(w1), P(b1), P(w2), P(b2), P(a), P(st2), P(f), P(u),
                                                                    P(g), P(out), B, C, Lq, 1e-5, s)), 3 * flc, 7 * e, bound="hbm")
            add("tok_tail_bwd", ev_time(lambda: L.afd_tok_tail_bwd(P(out), P(u), P(a), P(st2), P(g2), P(w2), P(w1), P(wo), P(du), P(df), P(da), P(datt), B, C, Lq, s)),
                3 * flc, 7 * e, bound="hbm")
            add("tok_head_bwd", ev_time(lambda: L.afd_tok_head_bwd(P(dq), P(x), P(st), P(g1), P(w_in), P(da), P(dh), P(dx), B, C, Lq, s)), 3 * flc, 7 * e, bound="hbm")
        ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, C, 3 * C, S, S, 1), L.afd_conv_wgrad_workspace_bytes(B, C, C, S, S, 1), 4) // 4, device=dev)
        dw3, db3, dwc, dbc = torch.empty(3 * C, C, device=dev), torch.empty(3 * C, device=dev), torch.empty(C, C, device=dev), torch.empty(C, device=dev)
        t_w = ev_time(lambda: L.afd_conv_wgrad(P(h), P(dq), P(dw3), P(db3), B, C, 3 * C, S, S, 1, 0, P(ws), s))
        t_w += 3 * ev_time(lambda: L.afd_conv_wgrad(P(f), P(du), P(dwc), P(dbc), B, C, C, S, S, 1, 0, P(ws), s))
        add("linear_wgrad", t_w, 6 * flc, (5 + 3 * 2) * e, launches=4, bound="hbm")
        part = torch.empty(B, 2, C, device=dev)
        add("layernorm_params", 2 * ev_time(lambda: L.afd_layernorm_c_bwd_params(P(x), P(dh), P(st), B, C, Lq, P(part), P(dbc), P(dbc), 0, s)), bytes_=2 * 2 * e,
            launches=2, bound="hbm")
        del qkv, o, lse, dq, x, h, a, f, u, g, out, du, df, da, datt, dh, dx
    # ---- time embedding, loss, optimiser / DDPM update ---------------------------------------------------------------
    temb = torch.randn(B, 256, device=dev)
    n_img = B * 3 * 32 * 32
    xi, ei = torch.randn(n_img, device=dev), torch.randn(n_img, device=dev)
    if train:
        for C in EMB:
            w, b = torch.randn(C, 256, device=dev), torch.randn(C, device=dev)
            o, dw, db = torch.empty(B, C, device=dev), torch.empty(C, 256, device=dev), torch.empty(C, device=dev)
            add("silu_linear_fwd", ev_time(lambda: L.afd_silu_linear_fwd(P(temb), P(w), P(b), P(o), B, 256, C, s)), 2.0 * B * 256 * C, bound="hbm",
                bytes_=4.0 * (B * 256 + 256 * C + B * C))
            add("silu_linear_bwd", ev_time(lambda: L.afd_silu_linear_bwd(P(temb), P(w), P(o), P(dw), P(db), None, B, 256, C, 0, s)), 2.0 * B * 256 * C, bound="hbm",
                bytes_=4.0 * (B * 256 + 256 * C + B * C))
        loss, wsl = torch.empty(1, device=dev), torch.empty(4096, device=dev)
        add("mse_fwd_bwd", ev_time(lambda: (L.afd_mse_fwd(P(xi), P(ei), P(loss), P(wsl), n_img, s), L.afd_mse_bwd(P(xi), P(ei), P(loss), P(xi), n_img, s))),
            bytes_=4.0 * 5 * n_img, launches=2, bound="hbm")
        pf, gf, mf, vf = (torch.zeros(N_PARAMS, device=dev) for _ in range(4))
        stt = torch.zeros(4, device=dev)

        def adam():
            L.afd_adamw_tick(P(stt), 0.9, 0.999, s)
            L.afd_adamw_step(P(pf), P(gf), P(mf), P(vf), N_PARAMS, P(stt), 3e-4, 0.9, 0.999, 1e-8, 0.01, 1.0, s)
        add("adamw", ev_time(adam), bytes_=28.0 * N_PARAMS, launches=2, bound="hbm")
    else:
        import ctypes, struct
        tabs = [torch.randn(1000, C, device=dev) for C in EMB]           # emb_layer(pos_encoding(t)) of every timestep: built once per trajectory
        os_ = [torch.empty(B, C, device=dev) for C in EMB]
        tix = torch.randint(0, 1000, (B,), device=dev)
        desc = b"".join(struct.pack("<QQQi4x", P(w), 0, P(o), w.shape[1]) for w, o in zip(tabs, os_))
        buf = ctypes.create_string_buffer(desc, len(desc))
        add("timestep_embedding_gather", ev_time(lambda: L.afd_gather_rows_batched(P(tix), ctypes.addressof(buf), len(EMB), B, 1000, s)),
            bytes_=4.0 * 2 * B * sum(EMB), bound="hbm")
        x32 = torch.randn(B, 32, 32, 32, device=dev); w_out = torch.randn(3, 32, 1, 1, device=dev); b_out = torch.randn(3, device=dev)
        y3 = torch.empty(B, 3, 32, 32, device=dev)
        add("outc_1x1", ev_time(lambda: L.afd_conv_fwd(P(x32), P(w_out), P(b_out), None, P(y3), B, 32, 3, 32, 32, 1, 0, s)), bytes_=4.0 * B * 35 * 1024, bound="hbm")
        al = torch.linspace(0.9, 0.99, 1000, device=dev)
        nz, xo = torch.empty(n_img, device=dev), torch.empty(n_img, device=dev)
        add("denoise_step", ev_time(lambda: L.afd_denoise_step(P(xi), P(ei), P(nz), P(al), P(al), P(al), 500, P(xo), n_img, s)), bytes_=16.0 * n_img, bound="hbm")
        add("randn (torch device generator: the per-step noise draw)", ev_time(lambda: nz.normal_()), bytes_=4.0 * n_img, bound="hbm")
    # ---- Config E: the per-step rotation of sampling (not part of the train step) -----------------------------------
    xr = torch.randn(B, 3, 32, 32, device=dev)
    t_rot = ev_time(lambda: ops.rotate_spline3_wrap(xr, 0.09))
    torch.cuda.synchronize()
    pmc, pmc_path = {}, "profiles/pmc_families.json" if train else "profiles/pmc_sample_families.json"
    try:
        pmc = json.load(open(os.path.join(ROOT, pmc_path)))
    except Exception:
        pass
    out = []
    for name, r in rows.items():
        sec = r["ms"] * 1e-3
        tf = r["flops"] / sec / 1e12 if r["flops"] else None
        gb = r["bytes"] / sec / 1e9 if r["bytes"] else None
        row = {"kernel": name, "launches_per_step": r["launches"], "ms_per_step": round(r["ms"], 4), "bound": r["bound"],
               "tflops": round(tf, 3) if tf else None, "gbs": round(gb, 1) if gb else None}
        if r["bound"] == "mfma" and "pipe_s" in r:
            # executed fraction: matrix-pipe time at the peak of the instruction each launch issues / measured time
            row["frac"] = round(r["pipe_s"] / sec, 4)
            row["frac_basis"] = "executed: issued products at the peak of the instruction issued (f16x2: 3 fp16 products, bf16x3: 6 bf16 products per fp32 product on the 2500 TFLOP/s 16-bit MFMA; Winograd 16/36 and direct 1x on the 157.3 TFLOP/s fp32 MFMA)"
            row["launches_by_form"] = {k: r[k] for k in ("h2", "bf3", "wino", "direct") if r.get(k)}
            row["frac_algorithmic_vs_fp32_peak"] = round(tf / MFMA_F32_PEAK_TF, 4)
        elif r["bound"] == "mfma":
            row["frac"] = round(tf / MFMA_F32_PEAK_TF, 4)          # attention: algorithmic flops against the fp32 matrix peak
        else:
            row["frac"] = round(gb / HBM_PEAK_GBS, 4) if gb else None
        if name in pmc and pmc[name].get("mfma_busy_frac") is not None:
            # NOT measured by this run: merged from the committed rocprofv3 PMC pass of an earlier profiling run
            row["mfma_busy_pmc_from_profile"] = pmc[name]["mfma_busy_frac"]
            if pmc[name].get("valu_busy_frac") is not None:
                row["valu_busy_pmc_from_profile"] = pmc[name]["valu_busy_frac"]
            row["pmc_source"] = pmc[name].get("source", pmc_path)
        out.append(row)
    out.sort(key=lambda z: -z["ms_per_step"])
    out.append({"kernel": "affine_spline3_wrap (Config E sampling, per denoise step" + ("; not in the train step)" if train else "; only with theta)"),
                "launches_per_step": 0, "ms_per_step": round(t_rot, 4), "bound": "fp64 vector", "gbs": round(8.0 * xr.numel() / (t_rot * 1e-3) / 1e9, 1)})
    return out, rows


def conv_layer_times(L, s, dev, B, ci, co, S, reps=5, fwd_only=False):
    """(fwd, dgrad, wgrad) ms of one 3x3 layer through the same dispatch the autograd shells use (ops.py): the
    transformed-weight entry points where the library covers the shape (non-zero workspace: the bf16x3 direct kernel or
    the fp32 Winograd kernels, the library's choice), else the direct kernels.  Also returns the form each pass took
    ("bf3" | "wino" | "direct")."""
    x = torch.randn(B, ci, S, S, device=dev); w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = torch.randn(B, co, S, S, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    ws = torch.empty(max(L.afd_conv_wgrad_workspace_bytes(B, ci, co, S, S, 3) // 4, 1), device=dev)
    nf, nd = L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 0), L.afd_conv3x3_wino_workspace_bytes(B, ci, co, S, S, 1)
    kinds = L.afd_conv3x3_weight_kinds(B, ci, co, S, S)
    u = torch.empty(max(nf, nd, 4) // 4, device=dev)
    # (weights_ready = 1 in the timed calls: a step builds every layer's weight image in ONE batched launch at its start,
    # ops.WinoStepPlan; sampling builds them once per trajectory)
    if nf:
        L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 0, kinds, s)
        tf = ev_time(lambda: L.afd_conv3x3_wino_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 0, u.data_ptr(), 1, kinds, s), reps)
    else:
        tf = ev_time(lambda: L.afd_conv_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, ci, co, S, S, 3, 0, s), reps)
    form = lambda n, bit: "direct" if not n else (("bf3" if kinds & 4 else "h2") if kinds & bit else "wino")
    if fwd_only:
        return tf, 0.0, 0.0, form(nf, 1), None, None
    if ci <= 3:
        td = 0.0
    elif nd:
        L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 0, kinds, s)
        td = ev_time(lambda: L.afd_conv3x3_wino_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), None, B, ci, co, S, S, u.data_ptr(), 1, kinds, s), reps)
    else:
        td = ev_time(lambda: L.afd_conv_dgrad(y.data_ptr(), w.data_ptr(), dx.data_ptr(), B, ci, co, S, S, 3, s), reps)
    tw = ev_time(lambda: L.afd_conv_wgrad(x.data_ptr(), y.data_ptr(), dw.data_ptr(), None, B, ci, co, S, S, 3, 0, ws.data_ptr(), s), reps)
    return tf, td, tw, form(nf, 1), form(nd, 2), ("direct", "wino", "bf3", "direct", "h2")[L.afd_conv_wgrad_form(B, ci, co, S, S, 3)]


def pmc_traffic(family, launches, path="profiles/pmc_families.json"):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/pmc_families.json for the
    train step, profiles/pmc_sample_families.json for the sampling forward; produced by tools/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as the guide's gfx950 correction prescribes).  None when the family
    has no committed pass."""
    try:
        d = json.load(open(os.path.join(ROOT, path)))[family]
        return int(d["hbm_bytes_per_step"] / max(1, launches)), d.get("source", path)
    except Exception:
        return None, None


def roofline_of(table, rows, pmc_path="profiles/pmc_families.json"):
    """The `roofline` object of the arg-max family of a kernel table (ms per step over ALL timed families)."""
    trow = table[0]
    name, r = trow["kernel"], rows[trow["kernel"]]
    traffic, tsrc = pmc_traffic(name, r["launches"], pmc_path)
    sec = r["ms"] * 1e-3
    common = {"kernel": name, "launches_per_step": r["launches"], "ms_per_step": round(r["ms"], 3),
              "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2), "traffic": traffic, "traffic_source": tsrc}
    if r["bound"] == "mfma" and "pipe_s" in r:
        # 3x3 convolution families: priced against the pipe they execute on.  achieved = the bf16-MFMA-equivalent rate of the
        # products the launches issue (frac x the bf16 peak; the few Winograd / direct launches of the family are converted at
        # their own instruction's peak, so frac is exactly matrix-pipe-time-at-peak / measured time)
        frac = r["pipe_s"] / sec
        ach_alg = r["flops"] / sec / 1e12
        return dict(common, bound="mfma", achieved=round(frac * MFMA_BF16_PEAK_TF, 1), peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=round(frac, 4),
                    achieved_algorithmic=round(ach_alg, 2), frac_algorithmic_vs_fp32_peak=round(ach_alg / MFMA_F32_PEAK_TF, 4),
                    launches_by_form=trow.get("launches_by_form"), mfma_busy_pmc_from_profile=trow.get("mfma_busy_pmc_from_profile"),
                    pmc_source=trow.get("pmc_source"),
                    basis="executed: the layers run direct matrix-core kernels on two fp16 pieces per operand under power-of-two scales that follow "
                          "the data (3 fp16 products per fp32 product, fp32-class results: csrc/h2_common.h; launches_by_form: h2; bf3 = round 2's "
                          "three-piece bf16 form, 6 products; wino = fp32 Winograd, 16/36 of the products; direct = fp32 MFMA); achieved = issued "
                          "products priced at the peak of the instruction that issues them, expressed on the fp16 / bf16 scale (2500 TFLOP/s dense); "
                          "achieved_algorithmic = direct-form fp32 flops (2*9*Cin*Cout per output pixel, SURVEY 8d) / HIP-event time")
    if r["bound"] == "mfma":
        ach = r["flops"] / sec / 1e12
        return dict(common, bound="mfma", achieved=round(ach, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(ach / MFMA_F32_PEAK_TF, 4),
                    mfma_busy_pmc_from_profile=trow.get("mfma_busy_pmc_from_profile"), valu_busy_pmc_from_profile=trow.get("valu_busy_pmc_from_profile"),
                    pmc_source=trow.get("pmc_source"),
                    basis="algorithmic attention flops (4*L*L*C forward, 2.5x that backward, SURVEY 8d) / HIP-event time, priced against the fp32 "
                          "matrix peak (fp32-class results).  Executed: S / dP as 3-piece bf16 splits on the matrix cores; at head dim 8 (90 % of the "
                          "flops) P V, dV, dK, dQ as two-piece fp16 products on the matrix cores too (one-pass backward), the softmax algebra and the "
                          "splits on the vector pipe, which bounds the kernels (PMC: valu_busy_pmc_from_profile)")
    ach = r["bytes"] / sec / 1e9
    return dict(common, bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))


def _cpu_train_leg(variant, c, B, steps, warm=1):
    """The CPU oracle's train step (fwd + autograd bwd + AdamW) on the host cores: (images/s, s/step) over `steps` timed steps
    after `warm` untimed ones."""
    import contextlib, io
    import afdm
    from oracle import ref_ops as R
    afdm.set_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        net = afdm.UNet(c_in=c, c_out=c, image_size=32, f_settings=dict(F_SET) if variant else None, device="cpu", variant=variant)
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    _, _, ah = R.noise_schedule(1000)
    g = torch.Generator().manual_seed(42)
    images = torch.rand(B, c, 32, 32, generator=g) * 2 - 1
    times = []
    for it in range(steps + warm):
        t0 = time.perf_counter()
        t = torch.randint(1, 1000, (B,))
        eps = torch.randn(images.shape)
        _, _, grads = R.train_step_loss_and_grads(sd, images, t, eps, variant, F_SET, ah)
        with torch.no_grad():
            for kname in sd:
                p, m[kname], v2[kname] = R.adamw_step(sd[kname], grads[kname], m[kname], v2[kname], it + 1, 3e-4)
                sd[kname].copy_(p)
        times.append(time.perf_counter() - t0)
    dt = sum(times[warm:]) / steps
    return B / dt, dt


def _cpu_sample_leg(n, steps):
    """10 denoise steps of the CPU oracle's sampling loop (Config D), scaled x 99.9 to the 999 of a trajectory
    (SURVEY 8d): (images/s for a full trajectory, s per denoise step)."""
    import contextlib, io
    import afdm
    from oracle import ref_ops as R
    afdm.set_seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        net = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET), device="cpu", variant=3)
    sd = net.state_dict()
    beta, alpha, ah = R.noise_schedule(1000)
    x = torch.randn(n, 3, 32, 32)
    with torch.no_grad():
        t0 = None
        for k, i in enumerate(reversed(range(999 - steps, 1000))):          # 1 warm-up + `steps` timed
            if k == 1:
                t0 = time.perf_counter()
            t = torch.full((n,), i, dtype=torch.long)
            x = R.denoise_step(beta, alpha, ah, x, R.unet_forward(sd, x, t, 3, F_SET), i, torch.randn_like(x))
        per = (time.perf_counter() - t0) / steps
    return n / (per * 999), per


def cpu_baseline(variant=3, batch=256):
    """The CPU oracle (oracle/ref_ops.py, torch-CPU fp32: the reference's own ATen path restated) on the GPU box's host
    cores, bounded samples of the same workloads (~60 s of CPU work in all): `value` is ONE train step of the benched
    configuration at the benched batch size (B=256: the size the GPU number is quoted on; one timed step after a B=16
    warm-up of the same graph), `legs` adds the same variant at B=64, BASELINE config 2 (variant 1, Config B) at B=64,
    BASELINE config 1 (variant 0, 1 channel, B=16) and the sampling loop (configs 1-3)."""
    torch.set_num_threads(host_threads())
    cores = torch.get_num_threads()
    _cpu_train_leg(variant, 3, 16, 1)                                    # warm the allocator / thread pool
    vB, dtB = _cpu_train_leg(variant, 3, batch, 1, warm=0)
    legs = [{"config": f"variant {variant} (Config {'ABCDE'[variant]}) train step, 3x32x32, B={batch}: 1 timed step", "value": round(vB, 2), "unit": "images/s",
             "s_per_step": round(dtB, 3)}]
    v, dt = _cpu_train_leg(variant, 3, 64, 2)
    legs.append({"config": f"variant {variant} train step, 3x32x32, B=64: 2 timed steps after 1 warm-up", "value": round(v, 2), "unit": "images/s",
                 "s_per_step": round(dt, 3)})
    if variant != 1:
        vb, dtb = _cpu_train_leg(1, 3, 64, 2)
        legs.append({"config": "BASELINE config 2: variant 1 (Config B, filtered resampling only), 3x32x32, B=64 train step (Train.ipynb:106)",
                     "value": round(vb, 2), "unit": "images/s", "s_per_step": round(dtb, 3)})
    v1, dt1 = _cpu_train_leg(0, 1, 16, 3)
    legs.append({"config": "BASELINE config 1: variant 0 (Config A), MNIST-shaped 1x32x32, B=16 train step", "value": round(v1, 2),
                 "unit": "images/s", "s_per_step": round(dt1, 3)})
    vs, per = _cpu_sample_leg(16, 10)
    legs.append({"config": "Config D sample, n=16: 10 timed denoise steps x 99.9 -> one 999-step trajectory", "value": round(vs, 4),
                 "unit": "images/s", "s_per_denoise_step": round(per, 3)})
    return {"value": round(vB, 2), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (oracle/ref_ops.py, torch-CPU fp32) train step, variant {variant}, B={batch} (the benched size), 1 timed step after a "
                      f"B=16 warm-up, {dtB:.2f} s/step",
            "legs": legs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--variant", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="eager launches only")
    ap.add_argument("--graph", action="store_true", help="hipGraph replay only (default: probe both after warm-up, keep the faster)")
    ap.add_argument("--no-sample", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--sample-n", type=int, default=256)
    ap.add_argument("--sample-steps", type=int, default=1000, help="noise_steps T of the sampling run (999 forwards at 1000)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("AFD_DIST_BACKEND", "nccl"))     # nccl = RCCL over xGMI
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    local = local % max(1, torch.cuda.device_count())        # (tests may stack ranks on one GPU with gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import afdm
    afdm.lib()                                    # raises if libafd_hip.so is missing: no fallback
    afdm.set_seed(42)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = afdm.UNet(c_in=3, c_out=3, image_size=32, f_settings=dict(F_SET) if args.variant else None,
                          device=dev, variant=args.variant).to(dev)
    diff = afdm.Diffusion(noise_steps=1000, img_size=32, device=dev)
    g = torch.Generator().manual_seed(42 + rank)
    images = (torch.rand(args.batch, 3, 32, 32, generator=g) * 2 - 1).to(dev)
    torch.manual_seed(42 + rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def make_step(use_graph):
        """(step callable, graph really in use): a failed capture is reported and falls back to eager launches."""
        st = afdm.TrainStep(model, diff, lr=3e-4, graph=use_graph, distributed=(world > 1))
        try:
            for _ in range(args.warmup):
                st(images)
            return st, use_graph
        except Exception as e:
            if not use_graph:
                raise
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
            st = afdm.TrainStep(model, diff, lr=3e-4, graph=False, distributed=(world > 1))
            for _ in range(args.warmup):
                st(images)
            return st, False

    def probe(st, n=12):
        gc.collect()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            st(images)
        sync()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)        # every rank takes the same decision
        return dt.item() / n

    # launch mode: the same step launched eagerly (with the weight-gradient kernels overlapped on a second stream), replayed from a
    # captured hipGraph, or re-issued from the captured graph on two real streams by a C++ loop ("lanes": csrc/replay.hip -- the
    # eager step's stream semantics without the interpreter).  Which is fastest depends on how quickly the host issues ~330
    # launches per step (5.6 ms of interpreter time against 6.7-7.0 ms of device time), so by default all three are probed
    # after warm-up (untimed) and the fastest one is measured.
    if args.no_graph or args.graph:
        step, graph_ok = make_step(args.graph)
        mode = "hipGraph" if graph_ok else "eager"
        log(f"rank {rank}/{world}: model ready ({mode})")
    else:
        probes = {}
        for mode, g in (("eager", False), ("hipGraph", True), ("lanes", "lanes")):
            step, ok = make_step(g)
            probes[mode] = probe(step) if (ok or not g) else float("inf")
            del step
        log(f"rank {rank}/{world}: probe " + ", ".join(f"{k} {v * 1e3:.2f} ms/step" for k, v in probes.items()))
        mode = min(probes, key=probes.get)
        step, graph_ok = make_step({"eager": False, "hipGraph": True, "lanes": "lanes"}[mode])
        if not graph_ok:
            mode = "eager"
    # Python's cyclic collector is run before and switched off during the timed steps (timeit's convention): a full collection
    # walks every module / tensor object of the process (~30 ms: one 20-step window in forty read 8.6 instead of 7.04 ms/step,
    # tools/jitter_probe.py); the steps themselves create no reference cycles that need it
    gc.collect()
    gc.disable()
    sync()
    log("timing train steps")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(images)
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    final_loss = float(loss.item())
    ms = dt / args.steps * 1e3
    value = args.batch * world / (dt / args.steps)

    result = {
        "metric": "train_step_images_per_sec", "value": round(value, 1), "unit": "images/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"CIFAR-10 32x32x3 synthetic, UNet variant={args.variant} (Config {'ABCDE'[args.variant]}), "
                               f"batch {args.batch}/GPU, T=1000, AdamW lr 3e-4, random-init weights (seed 42)",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "hip_graph": bool(graph_ok), "launch_mode": mode,
                   "host_gc": "collected before, disabled during the timed steps (timeit's convention)"},
        "final_loss": round(final_loss, 5),
    }

    log(f"train: {ms:.2f} ms/step, {value:.0f} img/s, loss {final_loss:.4f}")
    if not args.no_sample:
        log("sampling")
        n = args.sample_n
        ds = afdm.Diffusion(noise_steps=args.sample_steps, img_size=32, device=dev)
        sync()
        t0 = time.perf_counter()
        xq, _ = ds.sample(model, n=n, image_channels=3, noise_source="device")
        sync()
        sdt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([sdt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sdt = tt.item()
        result["sample"] = {"metric": "sample_images_per_sec", "value": round(n * world / sdt, 3), "unit": "images/s",
                            "n_per_gpu": n, "denoise_steps": args.sample_steps - 1, "seconds": round(sdt, 3)}
        # the same loop with four independent 256-image trajectories in flight per GPU (four HIP streams): throughput form
        # (measured, tools/sample_streams.py: 77 / 96 / 103 / 107 images/s with 1 / 2 / 3 / 4 trajectories in flight)
        NS = 4
        sync()
        t0 = time.perf_counter()
        ds.sample_concurrent(model, n=NS * n, image_channels=3, batch=n, streams=NS)
        sync()
        cdt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([cdt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cdt = tt.item()
        result["sample"]["trajectories_in_flight"] = {"value": round(NS * n * world / cdt, 3), "unit": "images/s",
                                                      "n_per_gpu": NS * n, "streams": NS, "seconds": round(cdt, 3)}

    if rank == 0 and world == 1:
        if not args.no_kernels and args.variant == 3:       # (the family tables are written over Config D's layer shapes)
            log("per-kernel timing: train step")
            table, rows = kernel_table(dev, args.batch, "train")
            result["roofline"] = roofline_of(table, rows)
            result["kernels"] = table
            if "sample" in result:
                log("per-kernel timing: sampling forward")
                stable, srows = kernel_table(dev, args.sample_n, "sample")
                result["sample"]["roofline"] = roofline_of(stable, srows, "profiles/pmc_sample_families.json")
                result["sample"]["kernels"] = stable
                result["sample"]["kernel_ms_sum_per_denoise_step"] = round(sum(r["ms"] for r in srows.values()), 4)
                result["sample"]["ms_per_denoise_step"] = round(result["sample"]["seconds"] * 1e3 / result["sample"]["denoise_steps"], 4)
        if not args.no_cpu_baseline:
            log(f"CPU baseline on {host_threads()} host threads")
            result["cpu_baseline"] = cpu_baseline(args.variant, args.batch)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
