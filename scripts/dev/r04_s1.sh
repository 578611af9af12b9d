#!/bin/bash
# round 4, GPU call 1: the whole GPU suite (no -x), the same under PGO_DEBUG_POISON=1, the C2 reproduction loop in fresh processes
mkdir -p gpurun_out/r04_s1
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > gpurun_out/r04_s1/sha.txt
( timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/r04_s1/gputests.log
( PGO_DEBUG_POISON=1 timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -80 ) > gpurun_out/r04_s1/gputests_poison.log
for i in $(seq 1 ${1:-8}); do
  timeout 300 python scripts/dev/repro_c2.py C2 0 > gpurun_out/r04_s1/repro_$i.log 2>&1
done
grep -h "^default:\|^plain" gpurun_out/r04_s1/repro_*.log | sort | uniq -c
tail -15 gpurun_out/r04_s1/gputests.log
tail -15 gpurun_out/r04_s1/gputests_poison.log
