#!/bin/bash
mkdir -p gpurun_out/r04_s9
python scripts/dev/mg_timeline.py C3 > gpurun_out/r04_s9/mg_timeline.txt 2>&1
python scripts/dev/ab_variant.py dense8 "-DPGO_DENSE_UNROLL8" 2 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s9/ab_dense8.txt 2>&1
cat gpurun_out/r04_s9/mg_timeline.txt gpurun_out/r04_s9/ab_dense8.txt
