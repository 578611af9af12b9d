#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03_sa
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_multigrid.py -x -q --durations=5 > $OUT/mg_tests.log 2>&1
tail -25 $OUT/mg_tests.log
for sm in 0 1 2; do echo "smoothed levels $sm"; python scripts/gpu_mg_profile.py mg_smoothed_levels=$sm 2>&1 | tail -1; done
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
for k in ['value','cg_iterations_total','cg_iterations_per_step','chi2_rel_diff']: print(k, d.get(k))
print({k: d['roofline_mg'][k] for k in ('avg_iteration_ms','frac','iterations_in_timed_region','share_of_timed_region')})
PY
