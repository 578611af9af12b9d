#!/bin/bash
mkdir -p gpurun_out/r04_s16
for v in 5 3.5 7 10 14; do
  for c in C3 C4; do
    echo "blocks per lane group <= $v:" $(PGO_DEBUG_SEG_BLOCKS=$v python scripts/dev/mg_iteration_time.py $c 2>/dev/null)
  done
done > gpurun_out/r04_s16/seg_scan.txt
cat gpurun_out/r04_s16/seg_scan.txt
