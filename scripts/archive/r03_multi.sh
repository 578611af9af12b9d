#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r03_multi
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_two_ranks_one_gpu.py tests/test_gpu_c5.py -x -q --durations=10 > $OUT/multi_tests.log 2>&1
tail -25 $OUT/multi_tests.log
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --collective gloo > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err
tail -c 1500 $OUT/bench_gloo2.json; tail -5 $OUT/bench_gloo2.err
