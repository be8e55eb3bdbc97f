#!/usr/bin/env python3
"""profiles/pmc_families.json from the counter passes of tools/profile_all.sh: per bench.py kernel family, the HBM
bytes per step (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), MFMA instructions and MFMA-busy fraction.
The three conv families come from their isolated runs (pmcf/), the others from the whole-step passes (pmc/) by kernel
name.    python tools/build_pmc_families.py gpurun_out/final [whole_step_steps=4] [family_reps=3]"""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "afd::" in k:
                kn = k.split("afd::")[1].split("<")[0].split("(")[0]
            elif k.startswith("_ZN3afd"):                      # (kernels with _Float16 vector arguments stay mangled in the trace)
                m = re.match(r"_ZN3afd(\d+)", k)
                kn = k[m.end():m.end() + int(m.group(1))]
            else:
                continue
            per[kn][r["Counter_Name"]] += float(r["Counter_Value"])
            n[kn].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in n.items()}


def summarise(pers, div, kernels=None):
    tot = collections.defaultdict(float)
    for per in pers:
        for kn, c in per.items():
            if kernels is None or kn in kernels:
                for name, v in c.items():
                    tot[name] += v
    fetch, write = tot["FETCH_SIZE"] * 1024 / div, tot["WRITE_SIZE"] * 1024 / div
    gui = tot["GRBM_GUI_ACTIVE"]
    return {"hbm_bytes_per_step": int(2 * fetch + write), "fetch_bytes_raw": int(fetch), "fetch_bytes_x2": int(2 * fetch),
            "write_bytes": int(write), "mfma_insts_per_step": int(tot["SQ_INSTS_MFMA"] / div),
            "valu_insts_per_step": int(tot["SQ_INSTS_VALU"] / div),
            "mfma_busy_frac": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 4) if gui else None,
            "valu_busy_frac": round(tot["SQ_ACTIVE_INST_VALU"] * 4 / (gui / 8 * 1024), 4) if gui and tot.get("SQ_ACTIVE_INST_VALU") else None}


out = {}
src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES passes (tools/r03_final.sh), FETCH_SIZE x2 per the gfx950 correction"
tag = os.path.basename(os.path.normpath(root))
for fam in ("fwd", "dgrad", "wgrad"):
    pers = [load(os.path.join(root, "pmcf", f"{fam}_{k}"))[0] for k in ("fetch", "write", "sq")]
    d = summarise(pers, reps)
    d["source"] = f"profiles/pmc_families.json ({tag}): {src}, over tools/conv_family.py {fam} (one step's launches of the family, B=256)"
    out["conv3x3_" + fam] = d
whole = [load(os.path.join(root, "pmc", k))[0] for k in ("fetch", "write", "sq")]
groups = {"attn_bwd": {"attn_bwd_dq_k", "attn_bwd_dkv_k", "attn_bwd_dq_mfma", "attn_bwd_dkv_mfma", "attn_bwd_fused", "attn_delta_k"},
          "attn_fwd": {"attn_fwd_k", "attn_fwd_mfma", "attn_fwd_pv"},
          "filt_act_fwd_n3": {"filt_act_fwd_n3"}, "filt_act_bwd_n3": {"filt_act_bwd_n3"},
          "groupnorm1_stats": {"gn_fwd_reg", "gn_fwd_loop"}, "groupnorm1_bwd_apply": {"gn_bwd_apply"}, "groupnorm1_bwd_full": {"gn_bwd_reg"},
          "tok_head_fwd": {"tok_head_fwd", "tok_head_fwd_wide"}, "tok_tail_fwd": {"tok_tail_fwd", "tok_tail_fwd_wide"},
          "tok_tail_bwd": {"tok_tail_bwd", "tok_tail_bwd_wide"}, "tok_head_bwd": {"tok_head_bwd", "tok_head_bwd_wide"},
          "linear_wgrad": {"conv_wgrad_mfma", "pw_wgrad_bf3", "pw_wgrad_h2"}, "layernorm_params": {"ln_c_bwd_plane"},
          "param_grad_folds": {"fold_batched_k"}}
for fam, ks in groups.items():
    d = summarise(whole, steps, ks)
    d["source"] = f"profiles/pmc_families.json ({tag}): {src}, whole train step (bench.py --no-graph), kernels {sorted(ks)}"
    out[fam] = d
outp = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_families.json"
json.dump(out, open(outp, "w"), indent=1)
for k, v in out.items():
    print(f"{k:20s} HBM {v['hbm_bytes_per_step'] / 1e6:9.1f} MB/step  (fetch x2 {v['fetch_bytes_x2'] / 1e6:8.1f}, write {v['write_bytes'] / 1e6:8.1f})  MFMA busy {v['mfma_busy_frac']}")
