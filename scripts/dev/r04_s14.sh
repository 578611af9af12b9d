#!/bin/bash
mkdir -p gpurun_out/r04_s14
python scripts/dev/ab_variant.py wide "-DPGO_MG_WIDE" 3 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s14/ab_wide.txt 2>&1
python scripts/dev/ab_variant.py wide "-DPGO_MG_WIDE" 1 -- scripts/dev/mg_iteration_time.py C4 > gpurun_out/r04_s14/ab_wide_c4.txt 2>&1
cat gpurun_out/r04_s14/ab_wide.txt gpurun_out/r04_s14/ab_wide_c4.txt
