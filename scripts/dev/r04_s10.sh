#!/bin/bash
mkdir -p gpurun_out/r04_s10
python scripts/dev/ab_variant.py precol "-DPGO_MG_PRECOL" 3 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s10/ab_precol.txt 2>&1
python scripts/dev/ab_variant.py precol "-DPGO_MG_PRECOL" 1 -- scripts/dev/mg_iteration_time.py C4 > gpurun_out/r04_s10/ab_precol_c4.txt 2>&1
cat gpurun_out/r04_s10/ab_precol.txt gpurun_out/r04_s10/ab_precol_c4.txt
