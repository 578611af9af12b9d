#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04_s19; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr -o t -- python scripts/dev/c3_twenty.py C3 > $O/trace.log 2>&1
python scripts/rocpd_summary.py gaps $(find $O/tr -name "*.db" | head -1) > $O/gaps_c3.txt
rm -rf $O/tr
python scripts/dev/c3_twenty.py C3 > $O/plain.txt 2>&1
cat $O/gaps_c3.txt | cut -c1-150 | head -40; tail -2 $O/trace.log; cat $O/plain.txt | tail -1
