#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_kernels.sh <outname> [bench args...]
# rocprofv3 kernel trace of bench.py, summarised per kernel into gpurun_out/<outname>.txt
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$root/gpurun_out/$name"
timeout -k 10 300 rocprofv3 --kernel-trace -d "$root/gpurun_out/$name" -o tr --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-stage-events "$@" > "$root/gpurun_out/$name.log" 2>&1
cd "$root"
python tools/trace_summary.py "$(find gpurun_out/$name -name '*kernel_trace.csv' | head -1)" 0 > "gpurun_out/$name.txt"
