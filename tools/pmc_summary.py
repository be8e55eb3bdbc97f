#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (csv output) per kernel family.

The passes are collected separately, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit
one pass; counters never share a run with the tracing domains other than --kernel-trace):

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR/fetch -o f -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR/write -o w -- python3 bench.py ...
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE ...

  python tools/pmc_summary.py DIR --steps N [--json out.json] [--txt out.txt]

Units / corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B as reported by
rocprofv3; on gfx950 FETCH_SIZE tallies a 128-B request as 64 B, so wide streaming reads are under-reported by
exactly 2x.  Both the raw figure and the 2x-corrected upper bound are printed; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def family(name):
    m = re.match(r"(?:void )?(?:afd::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = family(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, cnt, dur


def main():
    d = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    tout = sys.argv[sys.argv.index("--txt") + 1] if "--txt" in sys.argv else None
    acc, cnt, dur = load(d)
    rows = []
    for k in acc:
        a = acc[k]
        n = max(cnt[k].values()) / steps
        fetch = a.get("FETCH_SIZE", 0.0) * 1024 / steps
        write = a.get("WRITE_SIZE", 0.0) * 1024 / steps
        mf, ins = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("SQ_INSTS_MFMA", 0.0)
        busy = a.get("SQ_BUSY_CU_CYCLES", 0.0) or a.get("SQ_BUSY_CYCLES", 0.0)
        rows.append({"kernel": k, "launches_per_step": round(n, 2),
                     "fetch_bytes_per_step_raw": int(fetch), "fetch_bytes_per_step_x2": int(2 * fetch),
                     "write_bytes_per_step": int(write),
                     "mfma_insts_per_step": int(ins / steps), "mfma_busy_cycles_per_step": int(mf / steps),
                     "valu_insts_per_step": int(a.get("SQ_INSTS_VALU", 0.0) / steps),
                     "gui_active_cycles_per_step": int(a.get("GRBM_GUI_ACTIVE", 0.0) / steps),
                     "sq_busy_cycles_per_step": int(busy / steps)})
    rows.sort(key=lambda r: -(r["fetch_bytes_per_step_x2"] + r["write_bytes_per_step"]))
    lines = [f"{'kernel family':28s} {'n/step':>7s} {'fetch MB raw':>13s} {'fetch MB x2':>12s} {'write MB':>9s} {'MFMA insts':>11s} {'MFMA busy / (GUI/8*1024 SIMD)':>28s}"]
    for r in rows:
        gui = r["gui_active_cycles_per_step"]
        frac = r["mfma_busy_cycles_per_step"] / (gui / 8.0 * 1024.0) if gui else 0.0   # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
        r["mfma_busy_frac"] = round(frac, 4)
        lines.append(f"{r['kernel']:28s} {r['launches_per_step']:7.1f} {r['fetch_bytes_per_step_raw'] / 1e6:13.1f} "
                     f"{r['fetch_bytes_per_step_x2'] / 1e6:12.1f} {r['write_bytes_per_step'] / 1e6:9.1f} "
                     f"{r['mfma_insts_per_step']:11d} {frac:28.3f}")
    txt = "\n".join(lines)
    print(txt)
    if tout:
        open(tout, "w").write(txt + "\n")
    if jout:
        json.dump({"steps": steps, "source": d, "families": rows}, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
