#!/bin/bash
O=gpurun_out/r04_s29; mkdir -p $O
python scripts/dev/ab_variant.py packed "-DPGO_MF_PACKED_INDEX" 3 -- scripts/dev/mg_iteration_time.py C3 > $O/ab.txt 2>&1
cat $O/ab.txt
for g in C3 G6000; do
  timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_packed.so timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packed ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
