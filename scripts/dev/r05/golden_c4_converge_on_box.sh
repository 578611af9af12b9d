#!/bin/bash
# The C4 run to convergence (tests/golden/make_c3_trajectory.py ... converge) resumed from its checkpoint on a GPU box's HOST (EPYC 9575F: ~1.6x this container's single-thread
# speed, nothing else running) — CPU only; the checkpoint and the partial log are copied out every minute so that a cut-off call loses nothing.
OUT=gpurun_out/r05_c4conv
mkdir -p $OUT
( while true; do sleep 60; cp tests/golden/c4_converged.json.state.npz tests/golden/c4_converged.json.partial $OUT/ 2>/dev/null; done ) &
SAVER=$!
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 python tests/golden/make_c3_trajectory.py 60 C4 c4_converged.json mg converge > $OUT/log.txt 2>&1
kill $SAVER
cp tests/golden/c4_converged.json tests/golden/c4_converged.json.state.npz $OUT/ 2>/dev/null
tail -5 $OUT/log.txt
