#!/bin/bash
# usage (here, repo root): tools/copy_profiles.sh <tag> <round prefix, e.g. r02>
tag=$1; r=$2
src=gpurun_out/prof_$tag
cp $src/kernel_stats.csv profiles/${r}_bench_c2_train_kernel_stats.csv
for n in p1 p2 p3 pf pw; do cp $src/pmc_$n.txt profiles/${r}_pmc_$n.txt; done
cp $src/pmc_lds.json profiles/${r}_pmc_lds.json
cp $src/pmc_traffic.json profiles/${r}_pmc_traffic.json
for c in 2 3 4 5; do cp $src/bench_c$c.json profiles/${r}_bench_c${c}_train.json; done
python3 - $r <<'PY'
import json, sys
r = sys.argv[1]
for c in (2, 3, 4, 5):
    d = json.load(open(f"profiles/{r}_bench_c{c}_train.json"))
    print(f"C{c}", d["value"], d["ms_per_step"], (d.get("render_fps") or {}).get("value"), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"],
          (d.get("with_optimizer") or {}).get("ms_per_step"), (d.get("full_fine_stage") or {}).get("ms_per_step"))
PY
