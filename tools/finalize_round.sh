#!/bin/bash
# usage (GPU box, repo root): tools/finalize_round.sh <tag> profile|bench
# The round's evidence set, in two calls on two fresh boxes (a box that has just run the rocprofv3 passes measures the
# step ~5 % slower than a fresh one: 1.14 vs 1.08 ms at C2, so the bench lines are not taken behind the profiler):
#   profile: tools/profile_round.sh <tag> -- kernel stats + 5 PMC passes + HBM traffic of the default bench command under
#            gpurun_out/prof_<tag>/; copy pmc_traffic.json into profiles/r02_pmc_traffic.json HERE before the second call
#            (the bench lines quote the traffic of this tree's kernels from that file)
#   bench:   the bench lines of C2 (default command, with the CPU baseline), C3, C4, C5 under
#            gpurun_out/prof_<tag>/bench_c{2,3,4,5}.json
# Copy the files into profiles/ afterwards (tools/copy_profiles.sh).
tag=$1; what=$2
out=gpurun_out/prof_$tag
mkdir -p $out
if [ "$what" = profile ]; then
  bash tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/profile_round_$tag.log; exit 1; }
  ls $out
  exit 0
fi
timeout -k 10 600 python bench.py > $out/bench_c2.log 2>&1 || { echo "bench C2 failed"; tail -5 $out/bench_c2.log; exit 1; }
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline > $out/bench_c3.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > $out/bench_c4.log 2>&1 || exit 1
timeout -k 10 500 python bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_c5.log 2>&1 || exit 1
for c in 2 3 4 5; do grep '^{' $out/bench_c$c.log | tail -1 > $out/bench_c$c.json; done
ls $out
