#!/bin/bash
# round 5, call 7: end game of the PCG (no chunk in flight near convergence) and the single-reduction form of the two-level method's fused iteration
export TMPDIR=/tmp
OUT=gpurun_out/r05_s7
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
timeout 1200 python -m pytest tests/test_gpu_single_reduction.py tests/test_gpu_coarse.py tests/test_gpu_breakdown_retry.py tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider > $OUT/tests_new.txt 2>&1
tail -5 $OUT/tests_new.txt
for n in S400 S1000 S3000; do timeout 600 python scripts/dev/r05/ab_options.py $n 10 3 "" "cg_end_game=0" "cg_single_reduction=0" "cg_single_reduction=0,cg_end_game=0"; done > $OUT/ab_sessions.txt 2>&1
cat $OUT/ab_sessions.txt | cut -c1-130
timeout 600 python scripts/dev/r05/ab_options.py C3 20 3 "" "cg_end_game=0" > $OUT/ab_c3.txt 2>&1
cat $OUT/ab_c3.txt | cut -c1-130
timeout 600 python scripts/dev/r05/ab_options.py C2 10 2 "" "cg_end_game=0" "cg_single_reduction=0,cg_end_game=0" > $OUT/ab_c2.txt 2>&1
cat $OUT/ab_c2.txt | cut -c1-130
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/tests_all.txt 2>&1
tail -6 $OUT/tests_all.txt
