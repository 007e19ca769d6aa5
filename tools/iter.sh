#!/bin/bash
# usage (GPU box, repo root): tools/iter.sh <tag> [pytest -k expr]   -- quick parity run + per-kernel step listing
set -e
tag=$1; sel=${2:-deformation or fullsize}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "$sel" > gpurun_out/${tag}_t.log 2>&1 || { tail -30 gpurun_out/${tag}_t.log; exit 1; }
tail -2 gpurun_out/${tag}_t.log
bash tools/prof_kernels.sh $tag --steps 10 --warmup 3
python tools/step_kernels.py "$(find gpurun_out/$tag -name '*kernel_trace.csv' | head -1)" 8 > gpurun_out/${tag}_step.txt
timeout -k 10 100 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/${tag}_bench.json 2>/dev/null
python -c "import json;d=json.load(open('gpurun_out/${tag}_bench.json'));print(d['value'],d['ms_per_step'],d['stages_ms'])"
