#!/bin/bash
# Round-2 final evidence (run on the GPU box through gpurun): bench line, the same under rocprofv3 --kernel-trace --stats, PMC passes of K1 and the
# PCG kernels (scripts/profile_r02_pmc.sh), all configs, session replay, other graph types, kernel stats of a multigrid solve and of a session-sized solve.
# Output: gpurun_out/r02_final/ -> copied into profiles/ by hand.
export TMPDIR=/tmp
OUT=gpurun_out/r02_final
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/r02_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/r02_bench_under_rocprof.json 2> $OUT/bench_rocprof.err
python scripts/rocpd_summary.py stats $(find $OUT/trace -name "*.db" | head -1) > $OUT/r02_bench_kernel_stats.txt
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats -d $OUT/trace_mg -o mg -- python scripts/gpu_mg_profile.py > $OUT/mg_profile.log 2>&1
python scripts/rocpd_summary.py stats $(find $OUT/trace_mg -name "*.db" | head -1) > $OUT/r02_mg_kernel_stats.txt
rm -rf $OUT/trace_mg
rocprofv3 --kernel-trace --stats -d $OUT/trace_s -o s -- python scripts/research/session_one_solve.py 3000 > $OUT/session_one.log 2>&1
python scripts/rocpd_summary.py stats $(find $OUT/trace_s -name "*.db" | head -1) > $OUT/r02_session_kernel_stats.txt
rm -rf $OUT/trace_s
bash scripts/profile_r02_pmc.sh > $OUT/pmc.log 2>&1
python scripts/gpu_all_configs.py > $OUT/r02_all_configs.txt 2>&1
python scripts/gpu_session_replay.py 3000 600 100 2 > $OUT/r02_session_replay_2deg.jsonl 2> $OUT/replay.err
python scripts/gpu_mg_graph_types.py 20 > $OUT/r02_mg_graph_types.txt 2>&1
python scripts/gpu_session_aggregates.py 300,500,1000,3000,6000,12000,23000 512,768 > $OUT/r02_session_aggregates.txt 2>&1
ls -la $OUT gpurun_out/pmc_r02
