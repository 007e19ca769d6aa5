#!/bin/bash
# usage (GPU box, repo root): tools/pmc_run.sh <outname> "<counters>" [bench args...]
# one rocprofv3 --pmc pass over a short bench.py run; result CSV summarised into gpurun_out/<outname>.txt
set -e
name=$1; ctrs=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$root/gpurun_out/$name"
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc $ctrs -d "$root/gpurun_out/$name" -o pmc --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-stage-events --no-optimizer-pass --no-fps-pass --steps 3 --warmup 2 "$@" > "$root/gpurun_out/$name.log" 2>&1
cd "$root"
python tools/pmc_summary.py "$(find gpurun_out/$name -name '*counter_collection.csv' | head -1)" > "gpurun_out/$name.txt"
