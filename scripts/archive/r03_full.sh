#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03_full
mkdir -p $OUT
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/gpu_tests.log 2>&1
tail -15 $OUT/gpu_tests.log
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/bench_under_rocprof.json 2> $OUT/bench_rocprof.err
python scripts/rocpd_summary.py stats $(find $OUT/trace -name "*.db" | head -1) > $OUT/bench_kernel_stats.txt
rm -rf $OUT/trace
head -32 $OUT/bench_kernel_stats.txt
