#!/bin/bash
# usage (GPU box, repo root): tools/ab_libs.sh <tag> <lib suffix ...>   ("" = the regular libgs4d.so, written as "base")
# Runs the C2 train bench (no extras) once per library variant built by tools/build_variant_lib.sh, back to back on one box,
# and leaves the lines in gpurun_out/r3/ab_<tag>_<suffix>.json (compare the stages with tools/ab_table.py).
tag=$1; shift
mkdir -p gpurun_out/r3
for sfx in "$@"; do
  lib=libgs4d_$sfx.so; [ "$sfx" = base ] && lib=libgs4d.so
  GS4D_LIB_NAME=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-optimizer-pass $BENCH_ARGS > gpurun_out/r3/ab_${tag}_$sfx.json 2> gpurun_out/r3/ab_${tag}_$sfx.err || { echo "$sfx failed"; tail -3 gpurun_out/r3/ab_${tag}_$sfx.err; exit 1; }
done
