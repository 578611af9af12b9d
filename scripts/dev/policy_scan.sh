#!/bin/bash
# hybrid policy constants (dev): start factor, wait factor, equivalence, switch iterations
for cfg in "2.25 2.0 8 400" "1.5 2.0 8 400" "1.5 1.5 8 300" "1.5 1.5 8 250" "1.2 1.5 8 250" "1.5 1.5 10 200" "1.0 1.5 8 300"; do
  set -- $cfg
  echo "start $1 wait $2 equiv $3 switch $4"
  PGO_MG_START_FACTOR=$1 PGO_MG_WAIT_FACTOR=$2 PGO_MG_EQUIV=$3 python scripts/gpu_opt_scan2.py C3 "mg_switch_iterations=$4" 2>&1 | tail -1
  PGO_MG_START_FACTOR=$1 PGO_MG_WAIT_FACTOR=$2 PGO_MG_EQUIV=$3 python scripts/gpu_opt_scan2.py C4 "mg_switch_iterations=$4" 2>&1 | tail -1
done
