#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
python scripts/research/session_step_times.py 400,1000,3000 2>&1 | grep -v "^\[pgo\] it" > $O/session_steps.txt
for n in 400 3000; do
rocprofv3 --kernel-trace --stats -d $O/trace_s$n -o t -- python scripts/research/session_one_solve.py $n > $O/trace_s$n.log 2>&1
DB=$(find $O/trace_s$n -name "*.db" | head -1)
python scripts/rocpd_summary.py stats $DB > $O/session_${n}_kernel_stats.txt
python scripts/research/trace_gaps.py $DB 10 25 > $O/session_${n}_gaps.txt
rm -rf $O/trace_s$n
done
cat $O/session_steps.txt; head -30 $O/session_400_kernel_stats.txt | cut -c1-160; cat $O/session_400_gaps.txt | head -20
