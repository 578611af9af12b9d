#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
for V in 0 1; do
rocprofv3 --kernel-trace --stats -d $O/trace_c5_$V -o t -- python scripts/research/c5_three_steps.py C5 mg_smoothed_levels=$V > $O/trace_c5_$V.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/trace_c5_$V -name "*.db" | head -1) > $O/c5_smoothed${V}_kernel_stats.txt
rm -rf $O/trace_c5_$V
tail -1 $O/trace_c5_$V.log; head -22 $O/c5_smoothed${V}_kernel_stats.txt | cut -c1-150
done
