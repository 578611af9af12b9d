#!/bin/bash
mkdir -p gpurun_out/r04_s6
python scripts/dev/ab_variant.py nodesc "-DPGO_MG_NO_DESC" 3 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s6/ab_desc_c3.txt 2>&1
python scripts/dev/ab_variant.py nodesc "-DPGO_MG_NO_DESC" 1 -- scripts/dev/mg_iteration_time.py C4 > gpurun_out/r04_s6/ab_desc_c4.txt 2>&1
python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "build_graph\|hierarchy (host)" | head -30 > gpurun_out/r04_s6/build_phases.txt
( timeout 900 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_fullsize.py tests/test_gpu_c5.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | head -4 ) > gpurun_out/r04_s6/tests.log
cat gpurun_out/r04_s6/ab_desc_c3.txt gpurun_out/r04_s6/ab_desc_c4.txt gpurun_out/r04_s6/build_phases.txt gpurun_out/r04_s6/tests.log
