#!/bin/bash
# round 4, GPU call 3: where the new build stands (bench line, build_graph phases, all configs) and where the matvec's time goes (timeline variant)
mkdir -p gpurun_out/r04_s3
sha256sum solve_keyframe_pose_graph_amd/libpgo.so > gpurun_out/r04_s3/sha.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_s3/bench.json 2> gpurun_out/r04_s3/bench.err
python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "build_graph\|multigrid:.*keyframes\|host" | head -40 > gpurun_out/r04_s3/build_phases.txt
python scripts/dev/mf_timeline.py C3 --built > gpurun_out/r04_s3/mf_timeline.txt 2>&1
python scripts/gpu_all_configs.py > gpurun_out/r04_s3/all_configs.txt 2>&1
cat gpurun_out/r04_s3/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','lm_iters_per_s_including_transfers','chi2_rel_diff') if k in d}); print(d.get('roofline')); print(d.get('roofline_pcg')); print(d.get('roofline_mg'))"
cat gpurun_out/r04_s3/build_phases.txt
cat gpurun_out/r04_s3/mf_timeline.txt
cat gpurun_out/r04_s3/all_configs.txt
