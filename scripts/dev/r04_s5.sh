#!/bin/bash
mkdir -p gpurun_out/r04_s5
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > gpurun_out/r04_s5/sha.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "converged or c2_ten" 2>&1 | tail -25 ) > gpurun_out/r04_s5/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > gpurun_out/r04_s5/bench.json 2> gpurun_out/r04_s5/bench.err
python scripts/gpu_all_configs.py > gpurun_out/r04_s5/all_configs.txt 2>&1
python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "build_graph" | head -20 > gpurun_out/r04_s5/build_phases.txt
head -30 gpurun_out/r04_s5/tests.log
cat gpurun_out/r04_s5/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers','chi2_rel_diff','chi2_converged_rel_diff') if k in d}); print(d.get('converged'))"
cat gpurun_out/r04_s5/all_configs.txt
cat gpurun_out/r04_s5/build_phases.txt
tail -3 gpurun_out/r04_s5/bench.err
