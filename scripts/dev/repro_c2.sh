#!/bin/bash
# one gpurun call: (A) the driver's order that failed in GPUTEST_r03, (B) N fresh processes of repro_c2.py
mkdir -p gpurun_out/repro
( timeout 900 python -m pytest tests/test_gpu_c5.py tests/test_gpu_coarse.py -x -q -m gpu -k "c5 or small_configs" 2>&1 | tail -30 ) > gpurun_out/repro/A_driver_order.log
for i in $(seq 1 ${1:-12}); do
  timeout 300 python scripts/dev/repro_c2.py C2 1 > gpurun_out/repro/B_$i.log 2>&1
done
grep -h "^default:\|^plain" gpurun_out/repro/B_*.log | sort | uniq -c
tail -5 gpurun_out/repro/A_driver_order.log
