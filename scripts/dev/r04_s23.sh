#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04_s23; mkdir -p $O
for g in C2 G6000 P9000 C1F5; do
  python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
  PGO_LIBPGO_OVERRIDE=build/variants/libpgo_prev.so python -m tests.solve_digest $g 2>/dev/null | grep DIGEST | sed 's/^DIGEST //' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['graph'], d['sha256'][:16], d['final_cost'], d['cg_iterations'])"
done > $O/digests.txt 2>&1
cat $O/digests.txt
python scripts/research/session_step_times.py 400,3000 2>&1 | grep -v "^\[pgo\]"
rocprofv3 --kernel-trace --stats -d $O/tr -o t -- python scripts/research/session_one_solve.py 3000 > $O/trace.log 2>&1
python scripts/rocpd_summary.py stats $(find $O/tr -name "*.db" | head -1) | grep "coarse_\|gj_" | cut -c1-150
rm -rf $O/tr
python -m pytest tests/test_gpu_coarse.py -q -m gpu -x 2>&1 | tail -2
