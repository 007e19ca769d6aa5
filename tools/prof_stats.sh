#!/bin/bash
# usage (GPU box, repo root): tools/prof_stats.sh <name> [bench args...]  -> gpurun_out/r2/<name>_kernel_stats.csv
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $out/${name}_stats.log 2>&1
rc=$?
f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${name}_kernel_stats.csv
tail -n 2 $out/${name}_stats.log | cut -c1-300
exit $rc
