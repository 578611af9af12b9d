#!/bin/bash
# the idle-gap part of profile_r05_final.sh alone (a rocprofv3 segmentation fault took the C3 half in the final call)
export TMPDIR=/tmp
OUT=gpurun_out/r05_final
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
SHA=$(sha256sum solve_keyframe_pose_graph_amd/libpgo.so | cut -d' ' -f1)
{ echo "# libpgo.so sha256 $SHA"
  for t in "C3 20 0 cg_use_graph=0" "S400 10"; do set -- $t
    for attempt in 1 2 3; do
      rm -rf gpurun_out/r05_final/trace_gaps_$1
      rocprofv3 --kernel-trace -d gpurun_out/r05_final/trace_gaps_$1 -o t -- python scripts/dev/timed_region.py $1 $2 $3 $4 > $OUT/timed_$1.log 2>&1 && break
      echo "(rocprofv3 attempt $attempt failed: $(tail -1 $OUT/timed_$1.log | cut -c1-120))"
    done
    echo "## python scripts/dev/timed_region.py $1 $2 $3 $4   (C3: PCG chunks launched eagerly, not as hipGraphs — rocprofv3 7.2 --kernel-trace segfaults in hipGraphLaunch of this run once the end game interleaves eager chunks and graph replays; plain runs and the other traces are unaffected; segment 1 = warm-up leg, the LAST segment = the timed leg; under rocprofv3 --kernel-trace every kernel boundary costs more than in a plain run)"
    python scripts/rocpd_summary.py segments $(find gpurun_out/r05_final/trace_gaps_$1 -name "*.db" | head -1) 50 200
    grep "steps" $OUT/timed_$1.log | tail -1; rm -rf gpurun_out/r05_final/trace_gaps_$1; done; } > $OUT/r05_idle_gaps.txt 2>&1
cat $OUT/r05_idle_gaps.txt | cut -c1-180
