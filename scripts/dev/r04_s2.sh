#!/bin/bash
# round 4, GPU call 2: the driver's own command (with -x), the suite under PGO_DEBUG_POISON=1, and the small_configs test 20x in separate processes
mkdir -p gpurun_out/r04_s2
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > gpurun_out/r04_s2/sha.txt
( timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r04_s2/gputests.log
( PGO_DEBUG_POISON=1 timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r04_s2/gputests_poison.log
for i in $(seq 1 ${1:-20}); do
  ( timeout 300 python -m pytest tests/test_gpu_coarse.py -q -m gpu -p no:cacheprovider -k small_configs 2>&1 | tail -3 ) > gpurun_out/r04_s2/loop_$i.log
done
grep -h "passed\|failed" gpurun_out/r04_s2/loop_*.log | sort | uniq -c
tail -8 gpurun_out/r04_s2/gputests.log
tail -12 gpurun_out/r04_s2/gputests_poison.log
