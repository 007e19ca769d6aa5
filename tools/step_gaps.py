"""Idle gaps between consecutive kernels of one step (rocprofv3 kernel trace): where the GPU waits for the host."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
idx = [i for i, r in enumerate(rows) if "hexplane_pe_forward_kernel" in r["Kernel_Name"] or "hexplane_forward_kernel" in r["Kernel_Name"]]
a, b = idx[step], idx[step + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, kernels {b - a}")
prev = None
for r in rows[a:b + 1]:
    if prev is not None:
        gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3
        if gap > thr:
            print(f"  gap {gap:6.1f} us  after {prev['Kernel_Name'][:45]:45s} before {r['Kernel_Name'][:45]}")
    prev = r
