#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh <tag> <VAR=value|base> ...
# The C2 train bench (no extras) once per environment setting, back to back on one box; lines in
# gpurun_out/r3/ab_<tag>_<n>.json in the order given (compare the stages with tools/ab_table.py).
tag=$1; shift
mkdir -p gpurun_out/r3
n=0
for kv in "$@"; do
  out=gpurun_out/r3/ab_${tag}_$n
  if [ "$kv" = base ]; then
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimizer-pass $BENCH_ARGS > $out.json 2> $out.err
  else
    env "$kv" timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimizer-pass $BENCH_ARGS > $out.json 2> $out.err
  fi || { echo "$kv failed"; tail -3 $out.err; exit 1; }
  n=$((n+1))
done
