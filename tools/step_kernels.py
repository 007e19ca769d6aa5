"""Print the duration of every kernel of one training step from a rocprofv3 kernel trace (diagnostic)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pat = sys.argv[3] if len(sys.argv) > 3 else "."
idx = [i for i, r in enumerate(rows) if "hexplane_pe_forward_kernel" in r["Kernel_Name"] or "hexplane_forward_kernel" in r["Kernel_Name"]]
for r in rows[idx[step]:idx[step + 1]]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0]
    if re.search(pat, n):
        print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  {n[:60]}  grid={r.get('Grid_Size_X','')}x{r.get('Grid_Size_Y','')}")
