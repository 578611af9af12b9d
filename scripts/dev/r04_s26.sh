#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04_s26; mkdir -p $O
for g in C3 C4 G12000; do
  timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 300 python -m tests.solve_digest $g max_num_iterations=12 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python scripts/gpu_mg_profile.py > $O/trace.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/tr -name "*.db" | head -1) | grep "galerkin\|psTw\|mg_w_\|val_f32\|mg_ps_\|scatter" | cut -c1-150
rm -rf $O/tr
for r in 1 2; do
  echo "new  $(timeout 600 python scripts/dev/setup_time.py 2>/dev/null)"
  echo "prev $(PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 600 python scripts/dev/setup_time.py 2>/dev/null)"
done
timeout 900 python -m pytest tests/test_gpu_multigrid.py -q -m gpu -x 2>&1 | tail -2
