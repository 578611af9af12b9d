#!/bin/bash
O=gpurun_out/r04_s18; mkdir -p $O
for ga in 192 96 48 24 2; do
  echo "== graph_after $ga"
  PGO_DEBUG_GRAPH_AFTER=$ga python scripts/research/session_step_times.py 400,1000,3000 2>&1 | grep -v "^\[pgo\]" | grep keyframes
  PGO_DEBUG_GRAPH_AFTER=$ga python scripts/gpu_session_replay.py 3000 600 100 2 2>/dev/null | tail -1
done > $O/graph_after_scan.txt 2>&1
cat $O/graph_after_scan.txt
