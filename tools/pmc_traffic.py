"""Merge the FETCH_SIZE and WRITE_SIZE pass summaries (tools/pmc_summary.py output) into one JSON:
{"kernels": {"<kernel> grid=<n>": {"FETCH_SIZE_KB": .., "WRITE_SIZE_KB": ..}}} (KB per launch as reported)."""
import json, re, sys
out = {}
for path, field in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    for line in open(path):
        m = re.match(r"(.*) grid=(\d+) lds=\d+ .*%s=([0-9.eE+-]+)" % field, line.strip())
        if m:
            out.setdefault(f"{m.group(1)} grid={m.group(2)}", {})[field + "_KB"] = float(m.group(3))
print(json.dumps({
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 2` (C2), mean per launch; tools/profile_round.sh",
    "unit": "KB per launch as reported; gfx950 correction (MI355X_MICROARCH.md, HBM): bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
    "kernels": out}, indent=1))
