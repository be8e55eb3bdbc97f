#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this turns its `kernels` view into the same
per-kernel summary the CSV `--stats` output has, optionally grouped into kernel families.

  python tools/rocpd_stats.py gpurun_out/x/prof/x_results.db [out.csv] [--steps N]
"""
import csv
import math
import re
import sqlite3
import sys


def family(name):
    m = re.match(r"(?:void )?(?:afd::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = None
    if "--steps" in sys.argv:
        steps = int(sys.argv[sys.argv.index("--steps") + 1])
        args = [a for a in args if a != str(steps)]
    db = args[0]
    out = args[1] if len(args) > 1 else None
    con = sqlite3.connect(db)
    rows = {}
    for name, dur, vg, ag, lds in con.execute("select name, duration, vgpr_count, accum_vgpr_count, lds_size from kernels"):
        r = rows.setdefault(name, {"n": 0, "sum": 0, "sq": 0.0, "min": 1 << 62, "max": 0, "vgpr": vg, "agpr": ag, "lds": lds})
        r["n"] += 1; r["sum"] += dur; r["sq"] += float(dur) * dur; r["min"] = min(r["min"], dur); r["max"] = max(r["max"], dur)
    total = sum(r["sum"] for r in rows.values()) or 1
    table = []
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["sum"]):
        avg = r["sum"] / r["n"]
        sd = math.sqrt(max(r["sq"] / r["n"] - avg * avg, 0.0))
        table.append([name, r["n"], r["sum"], round(avg, 3), round(100.0 * r["sum"] / total, 2), r["min"], r["max"], round(sd, 3),
                      r["vgpr"], r["agpr"], r["lds"]])
    hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "VGPRs", "AGPRs", "LDSBytes"]
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(hdr)
            w.writerows(table)
    fam = {}
    for t in table:
        f = fam.setdefault(family(t[0]), [0, 0])
        f[0] += t[1]; f[1] += t[2]
    div = steps or 1
    print(f"total kernel time {total / 1e6 / div:.3f} ms" + (f" per step ({steps} steps)" if steps else ""))
    for k, (n, s) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:32s} calls {n / div:8.1f}  ms {s / 1e6 / div:8.3f}  {100.0 * s / total:5.1f}%")


if __name__ == "__main__":
    main()
