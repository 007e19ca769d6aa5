#!/bin/bash
# usage (GPU box, repo root): tools/pmc_cmd.sh <outname> "<counters>" <python script and args...>
# one rocprofv3 --pmc pass over a script; per-kernel summary in gpurun_out/r3/<outname>.txt
name=$1; ctrs=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
s=$1; shift
timeout -s KILL 150 rocprofv3 --pmc $ctrs -d /tmp/pmc_$name -o pmc --output-format csv -- python3 $root/$s "$@" > $root/gpurun_out/r3/$name.log 2>&1
cd $root
python tools/pmc_summary.py "$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)" "${PMC_FILTER:-.}" > gpurun_out/r3/$name.txt
