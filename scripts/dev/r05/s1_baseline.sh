#!/bin/bash
# round 5, call 1: baseline of the round-4 build — idle-gap segments of the timed region (C3 x 20 steps; a 400-keyframe and a 3000-keyframe trigger), verbose host phases
export TMPDIR=/tmp
OUT=gpurun_out/r05_s1
mkdir -p $OUT
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > $OUT/sha.txt
for t in "C3 20" "S400 10" "S3000 10"; do
  set -- $t
  rocprofv3 --kernel-trace -d $OUT/trace_$1 -o t -- python scripts/dev/timed_region.py $1 $2 > $OUT/timed_$1.log 2>&1
  python scripts/rocpd_summary.py segments $(find $OUT/trace_$1 -name "*.db" | head -1) 50 200 > $OUT/segments_$1.txt 2>&1
  rm -rf $OUT/trace_$1
done
python scripts/dev/timed_region.py C3 20 2 > $OUT/verbose_C3.log 2>&1
python scripts/dev/timed_region.py C3 20 0 > $OUT/plain_C3.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/plain_C3.log
