#!/bin/bash
# round 5, call 5: multigrid operators built beside the first block-Jacobi iterations — test, A/B on C3 / C4, and where the single-reduction form stops paying (C4, C5)
export TMPDIR=/tmp
OUT=gpurun_out/r05_s5
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
timeout 900 python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -p no:cacheprovider > $OUT/tests_new.txt 2>&1
tail -5 $OUT/tests_new.txt
timeout 600 python scripts/dev/r05/ab_options.py C3 20 3 "" "mg_overlap_build=0" > $OUT/ab_c3.txt 2>&1
cat $OUT/ab_c3.txt
timeout 900 python scripts/dev/r05/opt_types.py "types,C4,C5" "" "mg_overlap_build=0" "cg_single_reduction=0" "cg_single_reduction=0,mg_overlap_build=0" > $OUT/types.txt 2>&1
cat $OUT/types.txt
