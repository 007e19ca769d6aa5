"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, the mean counter values of
the LAST dispatch group (steady state), one line per (kernel, dispatch-shape)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0][:48]
    key = (n, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    d = acc.setdefault(key, collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (n, g, l), d in acc.items():
    if not re.search(sys.argv[2] if len(sys.argv) > 2 else ".", n):
        continue
    print(f"{n} grid={g} lds={l} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in d.items()))
