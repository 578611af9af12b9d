#!/bin/bash
for g in C3 C4 G6000 C2; do
  timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_operator.py -q -m gpu -x 2>&1 | tail -2
