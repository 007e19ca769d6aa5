"""usage: tools/ab_table.py <tag> <suffix ...>: the stage times of the bench lines tools/ab_libs.sh left, side by side."""
import json
import sys

tag, names = sys.argv[1], sys.argv[2:]
rows = {}
for n in names:
    d = json.loads([l for l in open(f"gpurun_out/r3/ab_{tag}_{n}.json") if l.startswith("{")][-1])
    rows[n] = dict(d["stages_ms"], STEP=d["ms_per_step"], FPS_ms=(d.get("render_fps") or {}).get("ms_per_frame") or 0.0)
print(f"{'stage':20s}" + "".join(f"{n:>10s}" for n in names))
for k in rows[names[0]]:
    print(f"{k:20s}" + "".join(f"{rows[n].get(k, 0):10.4f}" for n in names))
