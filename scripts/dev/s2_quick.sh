#!/bin/bash
for S in 10 7 5 3.5 2.5; do echo "blocks per lane group <= $S"; PGO_DEV_SEGT=$S python scripts/gpu_opt_scan3.py C3,C4 "" 2>&1 | grep -v "^\[pgo\]"; done
