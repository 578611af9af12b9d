#!/bin/bash
# The C5 golden with the reference's 10-iteration budget (tests/golden/make_c3_trajectory.py 10 C5 ... converge) resumed from its checkpoint on a GPU box's HOST (about 4x this
# container's speed) — CPU only.  After every finished iteration the first-N-iterations golden is written from the checkpoint (make_trajectory_from_checkpoint.py), so that a
# cut-off call still returns the longest trajectory reached; the 120-MB checkpoint itself stays on the box (gpurun_out is capped at 64 MiB).
OUT=gpurun_out/r05_c5budget
mkdir -p $OUT
( last=0
  while true; do sleep 45
    cp tests/golden/c5_ten_iterations.json.partial $OUT/ 2>/dev/null
    n=$(python -c "
import numpy as np
try:
    z = np.load('tests/golden/c5_ten_iterations.json.state.npz', allow_pickle=False); print(int(z['next_it']) - 1)
except Exception:
    print(0)")
    if [ "$n" -gt "$last" ]; then
      python tests/golden/make_trajectory_from_checkpoint.py tests/golden/c5_ten_iterations.json.state.npz C5 $n c5_first_${n}_iterations.json > /dev/null 2>&1 && cp tests/golden/c5_first_${n}_iterations.json $OUT/ && last=$n
    fi
  done ) &
SAVER=$!
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 python tests/golden/make_c3_trajectory.py 10 C5 c5_ten_iterations.json mg converge > $OUT/log.txt 2>&1
sleep 50; kill $SAVER
cp tests/golden/c5_ten_iterations.json $OUT/ 2>/dev/null
tail -8 $OUT/log.txt
