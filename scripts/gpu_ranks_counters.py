#!/usr/bin/env python3
"""Round 6: what the DISTRIBUTED multi-rank solver does, measured with `world` in-process ranks on ONE MI355X (pgo_comm_init_local) — everything multi-GPU stays
"unmeasured on hardware"; what can be measured on one GPU is what each rank's GPU would compute and send:

  * decisions and PCG iteration counts of the N-rank solve against the single-handle solve of the same graph;
  * per rank: level-kernel time of one multigrid cycle (pgo_time_kernel(7): the rank's share of every level, no exchanges, the ranks taking turns on the GPU) against the
    single handle's full cycle = what round 5's replicated levels cost on EVERY rank;
  * per rank: bytes sent per PCG iteration by the exchange plans against what round 5's union all-reduce carried on the same graph; exchanges per iteration;
  * per rank: the kernels of ONE multigrid set-up (pgo_time_kernel(8): level operators + dense inverse, no exchanges, ranks taking turns) with the distributed set-up
    (mg_dist_setup = 1) and with the replicated one (= 0: every rank forms every level), the blocks it forms and the bytes its block exchanges send per set-up.

  python scripts/gpu_ranks_counters.py C3 4 [lm_iterations] [policy] [option=value,...]      ->  one JSON line on stdout (and on stderr the library's hierarchy log of rank 0)"""
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding  # noqa: E402
from tests import util  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    policy = sys.argv[4] if len(sys.argv) > 4 else "spatial"
    extra = dict((k, int(v)) for k, v in (kv.split("=") for kv in sys.argv[5].split(","))) if len(sys.argv) > 5 and sys.argv[5] else {}      # e.g. mg_dist_min_rows=16384 (integer options)
    g = graphgen.config(name)
    q, t, s = util.initial_state(g, True)
    opts = dict(max_num_iterations=n_it, **extra)
    # ---- the single handle
    P = util.pgo_problem(g, True, **opts)
    t0 = time.time()
    q1, t1, s1, sum1 = P.solve(q, t, s)
    single_s = time.time() - t0
    P.solve_begin(q1, t1, s1)                      # a late linearisation: the hard systems are the ones the multigrid runs on
    cyc_ms, cyc_bytes = P.time_kernel(7, 50)
    it_ms, _ = P.time_kernel(6, 50)
    setup_ms, _ = P.time_kernel(8, 10)
    P.solve_end()
    P.close()
    # ---- the ranks
    parts = sharding.partition(g, world, policy)
    pst = sharding.partition_stats(g, parts)
    group = capi.local_group_create(world)
    out, stats, lvl, err = [None] * world, [None] * world, [None] * world, []
    wall = [0.0] * world
    su, su_rep = [None] * world, [None] * world

    def run_replicated(rank):      # the same ranks with rounds 3-5's replicated set-up: only its kernel time is wanted
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], mg_dist_setup=0, **opts)
            Pr.comm_init_local(rank, world, group)
            Pr.solve_begin(out[rank][0], out[rank][1], out[rank][2])
            su_rep[rank] = Pr.time_kernel(8, 10)
            Pr.solve_end()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))
            capi.local_group_abort(group)

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], verbosity=1 if rank == 0 else 0, **opts)
            Pr.comm_init_local(rank, world, group)
            t0 = time.time()
            out[rank] = Pr.solve(q, t, s)
            wall[rank] = time.time() - t0
            stats[rank] = Pr.sharding_stats().as_dict()
            Pr.set_options(verbosity=0)
            Pr.solve_begin(out[rank][0], out[rank][1], out[rank][2])
            lvl[rank] = Pr.time_kernel(7, 50)     # the ranks take turns inside the library
            su[rank] = Pr.time_kernel(8, 10)
            Pr.solve_end()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:   # noqa: BLE001
            err.append(repr(e))
            capi.local_group_abort(group)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    capi.local_group_destroy(group)
    if not err:
        group = capi.local_group_create(world)
        th = [threading.Thread(target=run_replicated, args=(r,)) for r in range(world)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        capi.local_group_destroy(group)
    if err:
        print(json.dumps({"config": name, "world": world, "error": err}))
        sys.exit(1)
    sumr = out[0][3]
    seq1 = [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
    seqr = [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)]
    rec = {
        "what": "in-process ranks on ONE MI355X (pgo_comm_init_local); nothing here is a multi-GPU timing", "config": name, "world": world, "policy": policy, "lm_iterations": n_it, "options": extra,
        "partition": pst,
        "decisions_equal": seq1 == seqr, "cost_rel_diff_max": max(abs(sumr.iterations[k].cost - sum1.iterations[k].cost) / sum1.iterations[k].cost for k in range(sum1.num_logged)),
        "pcg_iterations_single": int(sum1.cg_iterations), "pcg_iterations_ranks": int(sumr.cg_iterations), "pcg_ratio": sumr.cg_iterations / max(1, sum1.cg_iterations),
        "pcg_multigrid_single": int(sum1.cg_iterations_multigrid), "pcg_multigrid_ranks": int(sumr.cg_iterations_multigrid),
        "all_ranks_identical": all(np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][2], out[r][2]) for r in range(world)),
        "t_max_abs_diff": float(np.abs(out[0][1] - t1).max()),
        "single_handle": {"level_kernels_per_cycle_ms": cyc_ms, "multigrid_iteration_ms": it_ms, "setup_kernels_ms": setup_ms, "solve_s": single_s},
        "ranks": [{"rank": r, "level_kernels_per_cycle_ms": lvl[r][0], "level_kernel_time_vs_replicated": lvl[r][0] / cyc_ms,
                   "bytes_sent_per_mg_iteration": stats[r]["bytes_sent_per_mg_iteration"], "bytes_round5_per_mg_iteration": stats[r]["bytes_round5_per_mg_iteration"],
                   "bytes_vs_round5": stats[r]["bytes_sent_per_mg_iteration"] / max(1.0, stats[r]["bytes_round5_per_mg_iteration"]),
                   "bytes_sent_per_bj_iteration": stats[r]["bytes_sent_per_bj_iteration"], "bytes_round5_per_bj_iteration": stats[r]["bytes_round5_per_bj_iteration"],
                   "exchanges_per_mg_iteration": stats[r]["exchanges_per_mg_iteration"], "mg_levels": stats[r]["mg_levels"], "mg_levels_distributed": stats[r]["mg_levels_distributed"],
                   "mg_rows_own": stats[r]["mg_rows_own"], "mg_rows_total": stats[r]["mg_rows_total"], "mg_blocks_own": stats[r]["mg_blocks_own"], "mg_blocks_total": stats[r]["mg_blocks_total"],
                   "setup_kernels_ms": su[r][0], "setup_kernels_replicated_ms": su_rep[r][0], "setup_kernel_time_vs_replicated": su[r][0] / su_rep[r][0],
                   "mg_setup_levels_own_rows": stats[r]["mg_setup_levels_own_rows"], "mg_setup_exchanges": stats[r]["mg_setup_exchanges"], "mg_setup_blocks_own": stats[r]["mg_setup_blocks_own"],
                   "mg_setup_blocks_total": stats[r]["mg_setup_blocks_total"], "bytes_sent_per_mg_setup": stats[r]["bytes_sent_per_mg_setup"],
                   "bytes_allreduce_replicated_setup": stats[r]["bytes_allreduce_replicated_setup"],
                   "keyframes_local": stats[r]["keyframes_local"], "keyframes_shared": stats[r]["keyframes_shared"], "solve_wall_s_sharing_one_gpu": wall[r]} for r in range(world)],
    }
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
