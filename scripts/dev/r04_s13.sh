#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_s13
rocprofv3 --list-avail > gpurun_out/r04_s13/avail.txt 2>&1
grep -c . gpurun_out/r04_s13/avail.txt
grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" gpurun_out/r04_s13/avail.txt | sed 's/Name:[[:space:]]*//' | sort -u | tr '\n' ' ' | head -c 6000
