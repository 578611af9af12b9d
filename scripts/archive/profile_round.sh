#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py and separate PMC passes for HBM traffic of K1.
# Outputs land in gpurun_out/ (merged back); summaries are then copied into profiles/ by hand.
set -x
export TMPDIR=/tmp
ROUND=${1:-r01}
OUT=gpurun_out/prof_$ROUND
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# PMC: separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2) — kernel-trace only, no other trace domains
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python scripts/k1_only.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python scripts/k1_only.py > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
