#!/bin/bash
O=gpurun_out/r04_s20; mkdir -p $O
python scripts/dev/ab_variant.py pollcopy "-DPGO_POLL_BY_COPY" 3 -- scripts/dev/setup_time.py > $O/ab_poll.txt 2>&1
cat $O/ab_poll.txt
python -m pytest tests/test_gpu_determinism.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
