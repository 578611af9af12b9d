#!/bin/bash
# Round-6 evidence, run on the GPU box through gpurun on the round's FINAL build.  Everything lands in gpurun_out/r06_final/ and every file carries the sha256 of the
# libpgo.so that produced it (first line / "libpgo_sha256" key); scripts/pmc_r06_summary.py then writes the files committed under profiles/.
#   part "gate": the driver's own test command, the same under PGO_DEBUG_POISON=1, the collection order
#   part "prof": rocprofv3 kernel stats + PMC passes + bench line + per-config rows + session replay + multi-rank checks
export TMPDIR=/tmp
export PGO_ENABLE_DEBUG_HOOKS=1      # (the poisoned gate run needs the master switch of the debug hooks)
OUT=gpurun_out/r06_final
PM=gpurun_out/r06_final/pmc
mkdir -p gpurun_out/r06_final/pmc
python -c "from solve_keyframe_pose_graph_amd import _build; _build.build_libpgo(); _build.build_host(); _build.build_graphgen()"   # (the snapshot's file times can make the shipped library look stale: whatever the loader would rebuild is rebuilt NOW, before the sha is taken)
SHA=$(sha256sum solve_keyframe_pose_graph_amd/libpgo.so | cut -d' ' -f1)
echo $SHA > $OUT/libpgo_sha256.txt
stamp() { sed -i "1i # libpgo.so sha256 $SHA" "$1"; }
PARTS=${1:-gate,prof}
if [[ $PARTS == *gate* ]]; then
  {
    echo "## python -m pytest tests/ -x -q -m gpu      (the driver's command)"
    timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
    echo "## PGO_DEBUG_POISON=1 python -m pytest tests/ -q -m gpu      (every new device allocation NaN-filled)"
    PGO_DEBUG_POISON=1 timeout 1500 python -X faulthandler -m pytest tests/ -v -m gpu -p no:cacheprovider > $OUT/gate_poison_full.log 2>&1 < /dev/null; prc=$?
    grep -v "PASSED\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|^Extension modules\|^tests/.*::\|^$" $OUT/gate_poison_full.log | tail -$([ $prc -eq 0 ] && echo 6 || echo 80)      # (a crash keeps its whole traceback)
    echo "## python -m pytest tests -m gpu --collect-only -q | head -12      (oracle anchors of C1..C5 first)"
    python -m pytest tests -m gpu --collect-only -q -p no:cacheprovider 2>/dev/null | head -12
  } > $OUT/r06_gate.txt 2>&1
  stamp $OUT/r06_gate.txt
fi
[[ $PARTS == *prof* ]] || exit 0
trace() {   # trace <name> <stats file> <command...>: rocprofv3 --kernel-trace --stats of a command, summarised, the raw database dropped
  local name=$1 stats=$2; shift 2
  rocprofv3 --kernel-trace --stats -d gpurun_out/r06_final/trace_$name -o t -- "$@" > $OUT/trace_$name.log 2>&1
  python scripts/rocpd_summary.py stats $(find gpurun_out/r06_final/trace_$name -name "*.db" | head -1) > $OUT/$stats; stamp $OUT/$stats
  rm -rf gpurun_out/r06_final/trace_$name
}
pmc() {     # pmc <name> <counters...> -- <command...>: one --pmc pass; leaves the database for pmc_sum, which removes it
  local name=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d gpurun_out/r06_final/pmc/db_$name -o pmc -- "$@" > $PM/$name.log 2>&1
}
pmc_sum() { python scripts/rocpd_summary.py pmc $(find gpurun_out/r06_final/pmc/db_$1 -name "*.db" | head -1) $2 $3 > $PM/$4; }
pmc_drop() { rm -rf gpurun_out/r06_final/pmc/db_$1; }
trace mg r06_mg_kernel_stats.txt python scripts/gpu_mg_profile.py
trace k1big r06_k1_400k_kernel_stats.txt python scripts/k1_only.py 400000
trace session r06_session_kernel_stats.txt python scripts/research/session_one_solve.py 3000
for c in FETCH_SIZE WRITE_SIZE; do
  pmc k1_$c $c -- python scripts/k1_only.py; pmc_sum k1_$c $c k1_edges_kernel k1_$c.json; pmc_drop k1_$c
  pmc pcg_$c $c -- python scripts/gpu_pcg_kernel_times.py C3; pmc_sum pcg_$c $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_$c $c cg_update_kernel pcg_update_$c.json; pmc_drop pcg_$c
  pmc mg_$c $c -- python scripts/gpu_mg_iteration_only.py; pmc_sum mg_$c $c pgo mg_all_$c.json; pmc_drop mg_$c
done
pmc pcg_l2 TCC_HIT_sum TCC_MISS_sum -- python scripts/gpu_pcg_kernel_times.py C3
for c in TCC_HIT_sum TCC_MISS_sum; do pmc_sum pcg_l2 $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_l2 $c cg_update pcg_update_$c.json; done; pmc_drop pcg_l2
pmc pcg_sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -- python scripts/gpu_pcg_kernel_times.py C3
for c in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU; do pmc_sum pcg_sq $c mf_spmv pcg_spmv_$c.json; pmc_sum pcg_sq $c cg_update pcg_update_$c.json; done; pmc_drop pcg_sq
# the PMC passes condensed NOW (on this box) so that the bench below reports the static traffic figures of exactly this build, then the bench line (driver's command)

python scripts/pmc_r06_summary.py $OUT > $OUT/pmc_summary.log 2>&1
cp profiles/k1_pmc_latest.json profiles/k1_pmc_r06.json profiles/pcg_pmc_latest.json profiles/mg_pmc_latest.json profiles/r06_pcg_pmc.txt $OUT/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_final/trace_bench -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-k1-out-of-cache > $OUT/r06_bench_under_rocprof.json 2> $OUT/bench_rocprof.err
python scripts/rocpd_summary.py stats $(find gpurun_out/r06_final/trace_bench -name "*.db" | head -1) > $OUT/r06_bench_kernel_stats.txt; stamp $OUT/r06_bench_kernel_stats.txt
rm -rf gpurun_out/r06_final/trace_bench
# idle gaps of the timed region (bench.py's K LM steps, the device idle for 0.3 s on both sides) and of a 400-keyframe trigger
{ for t in "C3 20 0 cg_use_graph=0" "S400 10"; do set -- $t
    rocprofv3 --kernel-trace -d gpurun_out/r06_final/trace_gaps_$1 -o t -- python scripts/dev/timed_region.py $1 $2 $3 $4 > $OUT/timed_$1.log 2>&1
    echo "## python scripts/dev/timed_region.py $1 $2 $3 $4   (C3: PCG chunks launched eagerly, not as hipGraphs — rocprofv3 7.2 --kernel-trace segfaults in hipGraphLaunch of this run once the end game interleaves eager chunks and graph replays; plain runs and the other traces are unaffected; segment 1 = warm-up leg, the LAST segment = the timed leg; under rocprofv3 --kernel-trace every kernel boundary costs more than in a plain run)"
    python scripts/rocpd_summary.py segments $(find gpurun_out/r06_final/trace_gaps_$1 -name "*.db" | head -1) 50 200
    tail -1 $OUT/timed_$1.log; rm -rf gpurun_out/r06_final/trace_gaps_$1; done; } > $OUT/r06_idle_gaps.txt 2>&1; stamp $OUT/r06_idle_gaps.txt
# round 6: the by-density rule of the smoothed keyframe transition, the new two-level / multigrid crossover, the ranks' counters (in-process ranks on this one GPU)
python scripts/dev/r05/opt_types.py "types,C3,C2,C2S,mid" "" "mg_smoothed_fine=0" "mg_smoothed_fine=1" > $OUT/r06_smoothed_fine_rule.txt 2>&1; stamp $OUT/r06_smoothed_fine_rule.txt
python scripts/dev/r05/opt_types.py "scan" "mg_min_keyframes=0" "" > $OUT/r06_mg_crossover.txt 2>&1; stamp $OUT/r06_mg_crossover.txt
python scripts/gpu_ranks_counters.py C3 4 10 > $OUT/r06_ranks_c3x4.json 2> $OUT/ranks_c3x4.err
python scripts/gpu_ranks_counters.py C5 8 2 > $OUT/r06_ranks_c5x8.json 2> $OUT/ranks_c5x8.err
python scripts/gpu_ranks_counters.py C3 8 10 > $OUT/r06_ranks_c3x8.json 2> $OUT/ranks_c3x8.err
python scripts/gpu_ranks_counters.py C5 8 6 > $OUT/r06_ranks_c5x8_after_regroup.json 2> $OUT/ranks_c5x8_after_regroup.err      # (6 LM iterations: past the regroup inside the solve — another hierarchy: 4 distributed levels)
timeout 900 python scripts/gpu_dist_setup_check.py > $OUT/r06_dist_setup_check.txt 2>&1 < /dev/null; stamp $OUT/r06_dist_setup_check.txt
trace setup_ranks r06_dist_setup_kernel_stats.txt python scripts/dev/setup_kernels_ranks.py C5 8 1      # the set-up's kernels per rank (config 5 on 8 in-process ranks)
python scripts/dev/verbose_solve.py C3 2 2>&1 | grep "build_graph\|hierarchy (host)" > $OUT/r06_build_phases.txt; stamp $OUT/r06_build_phases.txt
python scripts/gpu_all_configs.py > $OUT/r06_all_configs.txt 2>&1; stamp $OUT/r06_all_configs.txt
python scripts/gpu_mg_graph_types.py 20 > $OUT/r06_mg_graph_types.txt 2>&1; stamp $OUT/r06_mg_graph_types.txt
python scripts/gpu_session_replay.py 3000 600 100 2 > $OUT/r06_session_replay_2deg.jsonl 2> $OUT/replay.err
python scripts/research/session_step_times.py 400,1000,3000 2>&1 | grep -v "^\[pgo\]" > $OUT/r06_session_step_times.txt; stamp $OUT/r06_session_step_times.txt
python scripts/gpu_multi_overhead.py > $OUT/multi_overhead.log 2>&1
python scripts/gpu_multi_overhead.py mg > $OUT/multi_overhead_mg.log 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --collective gloo > $OUT/r06_bench_gloo2.json 2> $OUT/bench_gloo2.err
python bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/r06_bench_c5_strong_1gpu.json 2> $OUT/bench_c5.err
cp gpurun_out/multi_overhead.json $OUT/r06_multi_overhead.json 2>/dev/null; cp gpurun_out/multi_overhead_mg.json $OUT/r06_multi_overhead_mg.json 2>/dev/null
python scripts/pmc_r06_summary.py $OUT > $OUT/pmc_summary.log 2>&1
ls -la $OUT $PM
