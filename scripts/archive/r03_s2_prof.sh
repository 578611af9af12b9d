#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace_mg -o t -- python scripts/gpu_mg_profile.py > $O/trace_mg.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/trace_mg -name "*.db" | head -1) > $O/mg_kernel_stats.txt
rm -rf $O/trace_mg
head -30 $O/mg_kernel_stats.txt | cut -c1-200
