#!/bin/bash
# round 5, call 4: where the explicit cycle's microseconds go (kernel stats of the multigrid iteration), and level-1 aggregates of 4 vs 8 on every graph type
export TMPDIR=/tmp
OUT=gpurun_out/r05_s4
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace_mg -o t -- python scripts/gpu_mg_iteration_only.py > $OUT/trace_mg.log 2>&1
python scripts/rocpd_summary.py stats $(find $OUT/trace_mg -name "*.db" | head -1) > $OUT/mg_iteration_kernel_stats.txt; rm -rf $OUT/trace_mg
head -14 $OUT/mg_iteration_kernel_stats.txt | cut -c1-60,82-140
timeout 1500 python scripts/dev/r05/opt_types.py "types,C4,C5" "" "mg_first_passes=2" > $OUT/first_passes.txt 2>&1
cat $OUT/first_passes.txt
