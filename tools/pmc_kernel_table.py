"""Per-kernel sums of every counter found under the given rocprofv3 --pmc output directories (one CSV per pass):
  python tools/pmc_kernel_table.py reps dir1 [dir2 ...]  -> JSON {kernel: {counter: value per rep, "dispatches": n per rep, "ns": kernel time per rep}}"""
import collections
import csv
import glob
import json
import os
import re
import sys

reps = int(sys.argv[1])
out = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "afd::" in k:
                kn = k.split("afd::")[1].split("(")[0]
            elif k.startswith("_ZN3afd"):                      # (kernels with _Float16 vector arguments stay mangled in the trace)
                m = re.match(r"_ZN3afd(\d+)", k)
                n = int(m.group(1)); name = k[m.end():m.end() + n]; rest = k[m.end() + n:]
                targs = re.findall(r"L[ib](\d+)E", rest.split("EEv")[0]) if rest.startswith("I") else []
                kn = name + ("<" + ", ".join(targs) + ">" if targs else "")
            else:
                continue
            out[kn][r["Counter_Name"]] += float(r["Counter_Value"]) / reps
            if (kn, r["Dispatch_Id"]) not in seen and d == sys.argv[2]:
                seen.add((kn, r["Dispatch_Id"]))
                out[kn]["dispatches"] += 1.0 / reps
                out[kn]["ns"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / reps
print(json.dumps({k: dict(v) for k, v in out.items()}, indent=1))
