#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag> [bench args, e.g. --config C5 --steps 5 --warmup 2]
# Produces the evidence set bench.py / DESIGN.md cite, under gpurun_out/prof_<tag>/ (copy into profiles/):
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats of the default bench.py command (C2 train)
#   pmc_{p1,p2,p3,pf,pw}.txt  one rocprofv3 --pmc pass each (separate runs), summarised per kernel
#   pmc_traffic.json        FETCH_SIZE / WRITE_SIZE per kernel (KB per launch as reported)
set -e
tag=$1; shift
ARGS="$@"
STEPS="--steps 30 --warmup 5"
case "$ARGS" in *--steps*) STEPS="";; esac
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out/stats" -o r02 --output-format csv -- python3 "$root/bench.py" $STEPS --no-cpu-baseline $ARGS > "$out/stats.log" 2>&1
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
cd "$root"
run_pmc() {
  name=$1; ctrs=$2
  bash tools/pmc_run.sh prof_$tag/$name "$ctrs" $PMC_ARGS
  mv gpurun_out/prof_$tag/$name.txt $out/pmc_$name.txt
}
# the counter passes run 3 timed steps of the same configuration (a --steps in the arguments is dropped)
PMC_ARGS=$(echo "$ARGS" | sed -E 's/--steps +[0-9]+//; s/--warmup +[0-9]+//')
run_pmc p1 "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
run_pmc p2 "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
run_pmc p3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum"
run_pmc pf "FETCH_SIZE"
run_pmc pw "WRITE_SIZE"
python tools/pmc_traffic.py $out/pmc_pf.txt $out/pmc_pw.txt > $out/pmc_traffic.json
python tools/pmc_lds.py $out/pmc_p2.txt > $out/pmc_lds.json
rm -rf $out/stats $out/p1 $out/p2 $out/p3 $out/pf $out/pw
ls $out
