#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace_c4 -o t -- python scripts/research/c5_three_steps.py C4 > $O/trace_c4.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/trace_c4 -name "*.db" | head -1) > $O/c4_kernel_stats.txt
rm -rf $O/trace_c4
tail -1 $O/trace_c4.log; head -24 $O/c4_kernel_stats.txt | cut -c1-150
grep "multigrid:" $O/trace_c4.log | head -3
