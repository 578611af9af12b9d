#!/bin/bash
for i in 1 2 3; do timeout 300 python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "hierarchy (host): pooled" ; done
for g in C3 G12000 C4; do
  timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
timeout 900 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_determinism.py tests/test_gpu_two_ranks_one_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['lm_iters_per_s_including_transfers'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['lm_iters_per_s_including_transfers'])"
