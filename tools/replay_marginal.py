"""What each kernel family costs the B=256 train step IN SITU: the two-lane replay (TrainStep(graph="lanes")) with the family's
nodes launched TWICE (AFD_REPLAY_DUP: every kernel still sees the data it saw) or, with --skip, dropped from the launch list
(AFD_REPLAY_SKIP: results are garbage and everything downstream runs on stale / zero data -- an upper bound) against the full
list, one process per variant, medians over windows.   python tools/replay_marginal.py [--skip]      (~10 s per family)"""
import os, subprocess, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMS = [("(full list)", ""), ("attn_bwd_fused + delta", "attn_bwd_fused,attn_delta"), ("attn_fwd_pv", "attn_fwd_pv"),
        ("attention, small blocks", "attn_fwd_k,attn_fwd_mfma,attn_bwd_dq,attn_bwd_dkv"),
        ("conv 3x3 tile kernels (fwd + dgrad)", "conv_h2I,conv_h2l"), ("conv 3x3 small-map kernels", "conv_h2_sk"),
        ("filt_act fwd", "filt_act_fwd"), ("filt_act bwd", "filt_act_bwd"), ("tok_* chains", "tok_"),
        ("GroupNorm kernels", "gn_"), ("wgrad 3x3 (side)", "wgrad_h2"), ("wgrad 1x1 (side)", "pw_wgrad"),
        ("folds (side)", "fold_batched"), ("LayerNorm params (side)", "ln_c_bwd_plane"), ("silu_linear dw (side)", "silu_linear_dw"),
        ("ALL side-lane kernels", "wgrad,fold_batched,ln_c_bwd_plane,silu_linear_dw,colsum")]
base = None
for name, pats in FAMS:
    env = dict(os.environ, TAG="x", **{("AFD_REPLAY_SKIP" if "--skip" in sys.argv else "AFD_REPLAY_DUP"): pats})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_median.py"), "--lanes", "--windows", "8"], env=env,
                         capture_output=True, text=True).stdout
    m = re.search(r"median ([0-9.]+)", out)
    t = float(m.group(1)) if m else float("nan")
    if base is None:
        base = t
    print(f"{name:40s} {t:7.3f} ms/step   {t - base:+.3f}", flush=True)
