#!/bin/bash
O=gpurun_out/r04_s31; mkdir -p $O
python scripts/dev/ab_variant.py dlds "-DPGO_UPD_D_LDS" 3 -- scripts/dev/mg_iteration_time.py C3 > $O/ab_dlds.txt 2>&1
cat $O/ab_dlds.txt | cut -c1-175
python scripts/dev/ab_variant.py dpred "-DPGO_UP_D_PRED" 3 -- scripts/dev/mg_iteration_time.py C3 > $O/ab_dpred.txt 2>&1
cat $O/ab_dpred.txt | cut -c1-175
for v in dlds dpred; do
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_$v.so timeout 300 python -m tests.solve_digest C3 max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
