#!/bin/bash
mkdir -p gpurun_out/r04_s12
python scripts/dev/ab_variant.py colfirst "-DPGO_MG_COLFIRST" 3 -- scripts/dev/mg_iteration_time.py C3 > gpurun_out/r04_s12/ab_colfirst.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers','chi2_rel_diff','chi2_converged_rel_diff')})" > gpurun_out/r04_s12/bench.txt
cat gpurun_out/r04_s12/ab_colfirst.txt gpurun_out/r04_s12/bench.txt
