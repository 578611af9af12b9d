#!/bin/bash
# round 5, call 3: explicit transfer operator of the smoothed transition — new tests, A/B on C3 / C4, hierarchy option scan, then the whole suite
export TMPDIR=/tmp
OUT=gpurun_out/r05_s3
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
timeout 900 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_coarse.py::test_a_graph_whose_hierarchy_does_not_coarsen_falls_back_to_the_two_level_method -x -q -m gpu -p no:cacheprovider > $OUT/tests_new.txt 2>&1
tail -5 $OUT/tests_new.txt
timeout 600 python scripts/dev/r05/ab_options.py C3 20 3 "" "mg_explicit_transfer=0" > $OUT/ab_c3.txt 2>&1
cat $OUT/ab_c3.txt
timeout 600 python scripts/dev/r05/ab_options.py C4 20 2 "" "mg_explicit_transfer=0" > $OUT/ab_c4.txt 2>&1
cat $OUT/ab_c4.txt
timeout 600 python scripts/dev/r05/ab_options.py C3 20 1 "mg_passes=3" "mg_dense_max_nodes=384" "mg_dense_max_nodes=256" "mg_passes=3,mg_dense_max_nodes=384" "mg_smoothed_levels=2" "mg_first_passes=2" > $OUT/scan_c3.txt 2>&1
cat $OUT/scan_c3.txt
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/tests_all.txt 2>&1
tail -8 $OUT/tests_all.txt
python bench.py --steps 20 --warmup 5 --no-k1-out-of-cache > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['lm_iters_per_s_including_transfers'], d['chi2_rel_diff'], d['chi2_converged_rel_diff']); print(json.dumps(d['timed_region_breakdown'], indent=1)); print(d['roofline_pcg']['avg_iteration_ms'], d['roofline_mg']['avg_iteration_ms'], d['roofline_mg']['frac'])"
