#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04_s17; mkdir -p $O
for n in 3000 400; do
  rocprofv3 --kernel-trace -d $O/tr_$n -o t -- python scripts/research/session_one_solve.py $n > $O/trace_$n.log 2>&1
  python scripts/rocpd_summary.py gaps $(find $O/tr_$n -name "*.db" | head -1) > $O/gaps_$n.txt
  rm -rf $O/tr_$n
  python scripts/research/session_one_solve.py $n > $O/plain_$n.txt 2>&1
done
python scripts/research/session_step_times.py 400,3000 2>&1 | grep -v "^\[pgo\]" > $O/step_times.txt
head -45 $O/gaps_3000.txt; cat $O/plain_3000.txt; head -30 $O/gaps_400.txt; cat $O/step_times.txt
