#!/bin/bash
# Round 3, first GPU call: the new failure-contract test, a bench line carrying roofline_mg, kernel stats of a multigrid solve (baseline of the round).
export TMPDIR=/tmp
OUT=gpurun_out/r03_first
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_failure_contract.py -x -q > $OUT/failure_test.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace_mg -o mg -- python scripts/gpu_mg_profile.py > $OUT/mg_profile.log 2>&1
python scripts/rocpd_summary.py stats $(find $OUT/trace_mg -name "*.db" | head -1) > $OUT/mg_kernel_stats.txt
rm -rf $OUT/trace_mg
tail -5 $OUT/failure_test.log; cat $OUT/bench.json | head -c 600
