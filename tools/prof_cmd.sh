#!/bin/bash
# usage (GPU box, repo root): tools/prof_cmd.sh <name> <python script and args...> -> gpurun_out/r3/<name>_kernel_stats.csv
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
s=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- python3 $root/$s "$@" > $out/${name}_stats.log 2>&1
rc=$?
f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $out/${name}_kernel_stats.csv
grep -v rocprofv3 $out/${name}_stats.log | tail -n 3 | cut -c1-300
exit $rc
