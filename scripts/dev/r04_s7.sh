#!/bin/bash
mkdir -p gpurun_out/r04_s7
python scripts/dev/ab_variant.py desc "-DPGO_MG_DESC" 2 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s7/ab_desc.txt 2>&1
python scripts/dev/ab_variant.py hoist "-DPGO_MF_HOIST_IDX" 2 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s7/ab_hoist.txt 2>&1
python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "build_graph" | head -30 > gpurun_out/r04_s7/build_phases.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers')})" > gpurun_out/r04_s7/bench.txt
cat gpurun_out/r04_s7/ab_desc.txt gpurun_out/r04_s7/ab_hoist.txt gpurun_out/r04_s7/build_phases.txt gpurun_out/r04_s7/bench.txt
