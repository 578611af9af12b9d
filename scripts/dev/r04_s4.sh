#!/bin/bash
# round 4, GPU call 4: merged PCG kernel heads — timeline, bench line, a parity subset
mkdir -p gpurun_out/r04_s4
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > gpurun_out/r04_s4/sha.txt
python scripts/dev/mf_timeline.py C3 > gpurun_out/r04_s4/mf_timeline.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > gpurun_out/r04_s4/bench.json 2> gpurun_out/r04_s4/bench.err
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_coarse.py tests/test_gpu_determinism.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r04_s4/tests.log
cat gpurun_out/r04_s4/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers','chi2_rel_diff','cg_iterations_total') if k in d}); print({k: d['roofline_pcg'][k] for k in ('frac','avg_iteration_ms','matvec','update')}); print({k: d['roofline_mg'][k] for k in ('frac','avg_iteration_ms','cycle_kernels')})"
cat gpurun_out/r04_s4/mf_timeline.txt
tail -5 gpurun_out/r04_s4/tests.log
