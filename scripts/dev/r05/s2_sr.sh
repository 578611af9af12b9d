#!/bin/bash
# round 5, call 2: single-reduction PCG + pause rule — new tests, anchors, A/B of the option sets on C3 (and C4)
export TMPDIR=/tmp
OUT=gpurun_out/r05_s2
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
timeout 900 python -m pytest tests/test_gpu_single_reduction.py tests/test_gpu_parity.py tests/test_gpu_coarse.py -x -q -m gpu -p no:cacheprovider > $OUT/tests_new.txt 2>&1
tail -5 $OUT/tests_new.txt
timeout 600 python scripts/dev/r05/ab_options.py C3 20 3 "" "cg_single_reduction=0" "cg_pause_always=1" "cg_single_reduction=0,cg_pause_always=1" > $OUT/ab_c3.txt 2>&1
cat $OUT/ab_c3.txt
timeout 600 python scripts/dev/r05/ab_options.py C4 20 2 "" "cg_single_reduction=0,cg_pause_always=1" > $OUT/ab_c4.txt 2>&1
cat $OUT/ab_c4.txt
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/tests_all.txt 2>&1
tail -8 $OUT/tests_all.txt
