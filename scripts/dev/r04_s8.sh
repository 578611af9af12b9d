#!/bin/bash
mkdir -p gpurun_out/r04_s8
python scripts/dev/ab_variant.py serialhead "-DPGO_SERIAL_HEAD" 3 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s8/ab_head.txt 2>&1
python scripts/dev/ab_variant.py row8 "-DPGO_MG_ROW8" 2 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s8/ab_row8.txt 2>&1
python scripts/dev/mf_timeline.py C3 > gpurun_out/r04_s8/mf_timeline.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_determinism.py tests/test_gpu_coarse.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep "passed\|failed" ) > gpurun_out/r04_s8/tests.log
cat gpurun_out/r04_s8/ab_head.txt gpurun_out/r04_s8/ab_row8.txt gpurun_out/r04_s8/mf_timeline.txt gpurun_out/r04_s8/tests.log
