"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals, and how much of the wall span the
device sat idle between consecutive kernels (diagnostic for launch-gap / host-bound time)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # drop the first N kernels (warm-up)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
tot = collections.defaultdict(lambda: [0, 0])
busy = 0
gaps = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0][:90]
    tot[n][0] += 1; tot[n][1] += e - s
    busy += e - s
    if prev_end is not None: gaps.append(max(0, s - prev_end))
    prev_end = max(prev_end or 0, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {sum(gaps)/1e6:.3f} ms")
gaps.sort()
print("gap us: median %.1f  p90 %.1f  max %.1f  count>20us %d" % (gaps[len(gaps)//2]/1e3, gaps[int(len(gaps)*.9)]/1e3, gaps[-1]/1e3, sum(g > 20000 for g in gaps)))
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t/1e6:9.3f} ms {c:6d} x {t/c/1e3:9.1f} us  {n}")
