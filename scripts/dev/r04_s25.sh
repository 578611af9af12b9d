#!/bin/bash
O=gpurun_out/r04_s25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_coarse.py tests/test_gpu_multigrid.py tests/test_gpu_determinism.py tests/test_gpu_parity.py tests/test_gpu_host_shim.py tests/test_gpu_breakdown_retry.py -q -m gpu -x 2>&1 | tail -3
for g in C2 G6000 P9000 G12000; do
  timeout 300 python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 300 python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
echo "new  $(timeout 600 python scripts/gpu_session_replay.py 3000 600 100 2 2>/dev/null | tail -1)"
echo "prev $(PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 600 python scripts/gpu_session_replay.py 3000 600 100 2 2>/dev/null | tail -1)"
echo "new  $(timeout 600 python scripts/gpu_session_replay.py 3000 600 100 2 2>/dev/null | tail -1)"
