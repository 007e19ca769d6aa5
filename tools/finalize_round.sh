#!/bin/bash
# usage (GPU box, repo root): tools/finalize_round.sh <tag>
# The round's evidence set in one go: tools/profile_round.sh <tag> (kernel stats + 5 PMC passes + traffic of the default
# bench command) and the bench lines of C2 (default command, with the CPU baseline), C3, C4, C5 under
# gpurun_out/prof_<tag>/bench_c{2,3,4,5}.json.  Copy the files into profiles/ afterwards (tools/copy_profiles.sh).
tag=$1
out=gpurun_out/prof_$tag
bash tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/profile_round_$tag.log; exit 1; }
# the bench lines quote the HBM traffic of THIS tree's kernels: the fresh PMC summary replaces the committed one first
cp $out/pmc_traffic.json profiles/r02_pmc_traffic.json
timeout -k 10 600 python bench.py > $out/bench_c2.log 2>&1 || { echo "bench C2 failed"; tail -5 $out/bench_c2.log; exit 1; }
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline > $out/bench_c3.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > $out/bench_c4.log 2>&1 || exit 1
timeout -k 10 500 python bench.py --config C5 --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_c5.log 2>&1 || exit 1
for c in 2 3 4 5; do grep '^{' $out/bench_c$c.log | tail -1 > $out/bench_c$c.json; done
ls $out
