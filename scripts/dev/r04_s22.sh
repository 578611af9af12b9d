#!/bin/bash
O=gpurun_out/r04_s22; mkdir -p $O
for g in C3 C4 G12000 C2; do
  python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done > $O/digests.txt 2>&1
cat $O/digests.txt
