#!/bin/bash
# per-step times with the preconditioner log, then the kernel-trace gaps of one 20-step C3 solve
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
python scripts/gpu_step_times.py C3 20 > $O/step_times.txt 2>&1
VERB=1 python scripts/research/c3_twenty_steps.py > $O/verbose.txt 2>&1
rocprofv3 --kernel-trace -d $O/trace -o t -- python scripts/research/c3_twenty_steps.py > $O/trace.log 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_summary.py stats $DB > $O/c3_20_kernel_stats.txt
python scripts/research/trace_gaps.py $DB 15 40 > $O/c3_20_gaps.txt
rm -rf $O/trace
