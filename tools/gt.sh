#!/bin/bash
# usage (GPU box, repo root): tools/gt.sh <name> <seconds> <command...>
# Runs one GPU step under its own time limit with output in gpurun_out/r3/<name>.log; exits with the step's status so
# that steps chained with && stop at the first failure.
name=$1; secs=$2; shift 2
mkdir -p gpurun_out/r3
timeout -k 10 "$secs" "$@" > "gpurun_out/r3/$name.log" 2>&1
rc=$?
echo "rc=$rc" >> "gpurun_out/r3/$name.log"
tail -n 4 "gpurun_out/r3/$name.log"
exit $rc
