#!/usr/bin/env python3
"""profiles/pmc_sample_families.json from the counter passes over tools/fwd_profile.py (30 denoise steps of Diffusion.sample at
n = 256): per bench.py sampling-table family, HBM bytes per denoise step (FETCH_SIZE doubled per the gfx950 correction +
WRITE_SIZE), MFMA / VALU instructions and the MFMA-busy fraction.   python tools/build_pmc_sample.py gpurun_out/DIR/fwd [steps=30]"""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30


def load(d):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "afd::" in k:
                kn = k.split("afd::")[1].split("<")[0].split("(")[0]
            elif k.startswith("_ZN3afd"):
                m = re.match(r"_ZN3afd(\d+)", k)
                kn = k[m.end():m.end() + int(m.group(1))]
            else:
                continue
            per[kn][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


pers = [load(os.path.join(root, k)) for k in ("fetch", "write", "sq1")]
groups = {"conv3x3_fwd": {"conv_h2", "conv_h2l", "conv_h2_sk", "conv_wino", "conv_wino_sk", "conv_mfma", "conv_bf3"},
          "attn_fwd": {"attn_fwd_pv", "attn_fwd_mfma", "attn_fwd_k"},
          "filt_act_fwd_n3": {"filt_act_fwd_n3"}, "groupnorm1_fwd_full": {"gn_fwd_reg", "gn_fwd_loop"},
          "tok_head_fwd": {"tok_head_fwd", "tok_head_fwd_wide"}, "tok_tail_fwd": {"tok_tail_fwd", "tok_tail_fwd_wide"}}
out = {}
src = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* passes over tools/fwd_profile.py (tools/r03_pmc.sh), FETCH_SIZE x2 per the gfx950 correction"
for fam, ks in groups.items():
    tot = collections.defaultdict(float)
    for per in pers:
        for kn, c in per.items():
            if kn in ks:
                for name, v in c.items():
                    tot[name] += v
    fetch, write = tot["FETCH_SIZE"] * 1024 / steps, tot["WRITE_SIZE"] * 1024 / steps
    gui = tot["GRBM_GUI_ACTIVE"]
    out[fam] = {"hbm_bytes_per_step": int(2 * fetch + write), "fetch_bytes_x2": int(2 * fetch), "write_bytes": int(write),
                "mfma_insts_per_step": int(tot["SQ_INSTS_MFMA"] / steps), "valu_insts_per_step": int(tot["SQ_INSTS_VALU"] / steps),
                "mfma_busy_frac": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024), 4) if gui else None,
                "valu_busy_frac": round(tot["SQ_ACTIVE_INST_VALU"] * 4 / (gui / 8 * 1024), 4) if gui else None,
                "source": f"profiles/pmc_sample_families.json ({os.path.basename(os.path.dirname(os.path.normpath(root)))}): {src}, kernels {sorted(ks)}"}
json.dump(out, open("profiles/pmc_sample_families.json", "w"), indent=1)
for k, v in out.items():
    print(f"{k:22s} HBM {v['hbm_bytes_per_step'] / 1e6:8.1f} MB/denoise step  MFMA busy {v['mfma_busy_frac']}  VALU busy {v['valu_busy_frac']}")
