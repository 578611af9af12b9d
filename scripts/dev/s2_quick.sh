#!/bin/bash
O=gpurun_out/s2; mkdir -p $O
for SF in 1.0 1.25 1.5 1.75 2.0 2.25; do for WF in 1.5 2.0; do
echo "start $SF wait $WF: C3 $(PGO_DEV_START_FACTOR=$SF PGO_DEV_WAIT_FACTOR=$WF python scripts/gpu_step_times.py C3 20 2>&1 | tail -n 1)  C4 $(PGO_DEV_START_FACTOR=$SF PGO_DEV_WAIT_FACTOR=$WF python scripts/gpu_step_times.py C4 20 2>&1 | tail -n 1)"
done; done
