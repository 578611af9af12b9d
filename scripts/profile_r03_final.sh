#!/bin/bash
# Round-3 evidence, run on the GPU box through gpurun on the round's FINAL build.  Everything lands in gpurun_out/r03_final/ and every file carries the sha256 of the
# libpgo.so that produced it (first line / "libpgo_sha256" key); scripts/pmc_r03_summary.py then writes the files committed under profiles/.
export TMPDIR=/tmp
OUT=gpurun_out/r03_final
PM=gpurun_out/r03_final/pmc
mkdir -p gpurun_out/r03_final/pmc
SHA=$(sha256sum solve_keyframe_pose_graph_amd/libpgo.so | cut -d' ' -f1)
echo $SHA > $OUT/libpgo_sha256.txt
stamp() { sed -i "1i # libpgo.so sha256 $SHA" "$1"; }
trace() {   # trace <name> <stats file> <command...>: rocprofv3 --kernel-trace --stats of a command, summarised, the raw database dropped
  local name=$1 stats=$2; shift 2
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03_final/trace_$name -o t -- "$@" > $OUT/trace_$name.log 2>&1
  python scripts/rocpd_summary.py stats $(find gpurun_out/r03_final/trace_$name -name "*.db" | head -1) > $OUT/$stats; stamp $OUT/$stats
  rm -rf gpurun_out/r03_final/trace_$name
}
pmc() {     # pmc <name> <counters...> -- <command...>: one --pmc pass; leaves the database for pmc_sum, which removes it
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d gpurun_out/r03_final/pmc/db_$name -o pmc -- "$@" > $PM/$name.log 2>&1
}
pmc_sum() { python scripts/rocpd_summary.py pmc $(find gpurun_out/r03_final/pmc/db_$1 -name "*.db" | head -1) $2 $3 > $PM/$4; }
pmc_drop() { rm -rf gpurun_out/r03_final/pmc/db_$1; }
# 2. one hard LM system with the multigrid from the first iteration; K1 on the 400k-keyframe graph (its output cannot sit in the Infinity Cache); a 3000-keyframe session solve
trace mg r03_mg_kernel_stats.txt python scripts/gpu_mg_profile.py
trace k1big r03_k1_400k_kernel_stats.txt python scripts/k1_only.py 400000
trace session r03_session_kernel_stats.txt python scripts/research/session_one_solve.py 3000
# 3. PMC passes (separate runs, --kernel-trace only): HBM traffic of K1, of the block-Jacobi PCG kernels and of every kernel of a multigrid PCG iteration; wait / L2 counters
for c in FETCH_SIZE WRITE_SIZE; do
  pmc k1_$c $c -- python scripts/k1_only.py; pmc_sum k1_$c $c k1_edges_kernel k1_$c.json; pmc_drop k1_$c
  pmc pcg_$c $c -- python scripts/gpu_pcg_kernel_times.py C3; pmc_sum pcg_$c $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_$c $c cg_update pcg_update_$c.json; pmc_drop pcg_$c
  pmc mg_$c $c -- python scripts/gpu_mg_iteration_only.py; pmc_sum mg_$c $c pgo mg_all_$c.json; pmc_drop mg_$c
done
pmc pcg_l2 TCC_HIT_sum TCC_MISS_sum -- python scripts/gpu_pcg_kernel_times.py C3
for c in TCC_HIT_sum TCC_MISS_sum; do pmc_sum pcg_l2 $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_l2 $c cg_update pcg_update_$c.json; done; pmc_drop pcg_l2
pmc pcg_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- python scripts/gpu_pcg_kernel_times.py C3
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU; do pmc_sum pcg_sq $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_sq $c cg_update pcg_update_$c.json; done; pmc_drop pcg_sq
# 3b. condense the PMC passes NOW (on this box) so that the bench below reports the static traffic figures of exactly this build, then the bench line (driver's command)
python scripts/pmc_r03_summary.py $OUT > $OUT/pmc_summary.log 2>&1
cp profiles/k1_pmc_latest.json profiles/k1_pmc_r03.json profiles/pcg_pmc_latest.json profiles/mg_pmc_latest.json profiles/r03_pcg_pmc.txt $OUT/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/r03_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r03_final/trace_bench -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/r03_bench_under_rocprof.json 2> $OUT/bench_rocprof.err
python scripts/rocpd_summary.py stats $(find gpurun_out/r03_final/trace_bench -name "*.db" | head -1) > $OUT/r03_bench_kernel_stats.txt; stamp $OUT/r03_bench_kernel_stats.txt
rm -rf gpurun_out/r03_final/trace_bench
# 4. all configs, graph types, smoothed-prolongator A/B, session replay, multi-rank overhead, the 2-rank bench through gloo on this one GPU
python scripts/gpu_all_configs.py > $OUT/r03_all_configs.txt 2>&1; stamp $OUT/r03_all_configs.txt
python scripts/gpu_mg_graph_types.py 20 > $OUT/r03_mg_graph_types.txt 2>&1; stamp $OUT/r03_mg_graph_types.txt
python scripts/gpu_sa_ab.py c3,c4,types > $OUT/r03_smoothed_ab.txt 2>&1; stamp $OUT/r03_smoothed_ab.txt
python scripts/gpu_session_replay.py 3000 600 100 2 > $OUT/r03_session_replay_2deg.jsonl 2> $OUT/replay.err
python scripts/gpu_c5_tolerance_check.py 2>&1 | grep -v "^\[pgo\]" > $OUT/r03_c5_tolerance.txt; stamp $OUT/r03_c5_tolerance.txt
python scripts/gpu_c3_tolerance_scan.py 2>&1 | grep -v "^\[pgo\]" > $OUT/r03_c3_tolerance_scan.txt; stamp $OUT/r03_c3_tolerance_scan.txt
python scripts/gpu_mg_crossover.py 4000,6000,8000,12000,18000 2>&1 | grep -v "^\[pgo\]" > $OUT/r03_mg_crossover.txt; stamp $OUT/r03_mg_crossover.txt
python scripts/research/session_step_times.py 400,1000,3000 2>&1 | grep -v "^\[pgo\]" > $OUT/r03_session_step_times.txt; stamp $OUT/r03_session_step_times.txt
python scripts/gpu_multi_overhead.py > $OUT/multi_overhead.log 2>&1
python scripts/gpu_multi_overhead.py mg > $OUT/multi_overhead_mg.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --collective gloo > $OUT/r03_bench_gloo2.json 2> $OUT/bench_gloo2.err
cp gpurun_out/multi_overhead.json $OUT/r03_multi_overhead.json 2>/dev/null; cp gpurun_out/multi_overhead_mg.json $OUT/r03_multi_overhead_mg.json 2>/dev/null
ls -la $OUT $PM
