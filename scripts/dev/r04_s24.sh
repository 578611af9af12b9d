#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04_s24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_coarse.py -q -m gpu -x 2>&1 | tail -3
for g in C2 G6000 C3 C1F5; do
  timeout 300 python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 300 python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done > $O/digests.txt 2>&1
cat $O/digests.txt
for r in 1 2 3; do
  echo "new  $(timeout 600 python scripts/dev/setup_time.py 2>/dev/null)"
  echo "prev $(PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so timeout 600 python scripts/dev/setup_time.py 2>/dev/null)"
done > $O/ab.txt
cat $O/ab.txt
timeout 600 python scripts/research/session_step_times.py 400,3000 2>&1 | grep -v "^\[pgo\]"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python scripts/research/session_one_solve.py 3000 > $O/trace.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/tr -name "*.db" | head -1) | grep "gj_" | cut -c1-150
rm -rf $O/tr
