#!/bin/bash
mkdir -p gpurun_out/r04_s11
( timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 ) > gpurun_out/r04_s11/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers','chi2_rel_diff','chi2_converged_rel_diff')}); print({k: d['roofline_pcg'][k] for k in ('frac','avg_iteration_ms')}, {k: d['roofline_mg'][k] for k in ('frac','avg_iteration_ms')})" > gpurun_out/r04_s11/bench.txt
python scripts/dev/verbose_solve.py C3 3 2>&1 | grep "build_graph\|installed at its first use" | head -20 > gpurun_out/r04_s11/build_phases.txt
cat gpurun_out/r04_s11/tests.log gpurun_out/r04_s11/bench.txt gpurun_out/r04_s11/build_phases.txt
