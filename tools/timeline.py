"""Main-stream / side-stream timeline of one train step from a rocprofv3 --kernel-trace CSV: busy time, idle gaps and
the dependent-launch gaps per queue, for the last `steps` steps (a step starts at each noise_images_k launch)."""
import csv
import sys
import collections

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [r["s"] for r in rows if "noise_images" in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("no step markers")
lo, hi = starts[-3], starts[-1]               # two full steps
sel = [r for r in rows if lo <= r["s"] < hi]
nstep = 2
span = (hi - lo) / nstep
print(f"step span {span/1e6:.3f} ms, kernels/step {len(sel)/nstep:.0f}")
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs) / nstep
    gaps = []
    for a, b in zip(rs, rs[1:]):
        gaps.append(b["s"] - a["e"])
    pos = [g for g in gaps if g > 0]
    small = [g for g in pos if g < 20000]
    print(f"queue {q}: {len(rs)/nstep:.0f} kernels/step, busy {busy/1e6:.3f} ms/step, gaps>0: {len(pos)/nstep:.0f}/step sum {sum(pos)/nstep/1e6:.3f} ms "
          f"(of which <20us: {len(small)/nstep:.0f}, sum {sum(small)/nstep/1e6:.3f} ms, median {sorted(small)[len(small)//2]/1e3 if small else 0:.2f} us)")
# union busy (any queue) and overlap
ev = []
for r in sel:
    ev.append((r["s"], 1)); ev.append((r["e"], -1))
ev.sort()
cur = 0; last = lo; t1 = t2 = 0
for t, d in ev:
    if cur >= 1: t1 += t - last
    if cur >= 2: t2 += t - last
    cur += d; last = t
print(f"any-queue busy {t1/nstep/1e6:.3f} ms/step, >=2 kernels concurrently {t2/nstep/1e6:.3f} ms/step")
# main-queue kernel time by family
mainq = max(byq, key=lambda q: len(byq[q]))
fam = collections.Counter()
for r in byq[mainq]:
    n = r["Kernel_Name"]
    n = n.split("(")[0].replace("void afd::", "").replace("afd::", "")
    fam[n.split("<")[0]] += (r["e"] - r["s"]) / nstep
for k, v in fam.most_common(25):
    print(f"   main {k:32s} {v/1e6:.3f} ms/step")
