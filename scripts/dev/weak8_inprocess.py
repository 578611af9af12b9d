#!/usr/bin/env python3
"""The graph of `bench.py --gpus N` (N x C3 as one graph: N x 100 000 poses, N x 100 003 switchable loops) solved by N in-process ranks on ONE GPU and by a single handle: a functional check
of the weak-scaling leg at its real size (decisions, costs, PCG counts, the hierarchy's shape, the counters) — the timings mean nothing (the ranks share one GPU).
  python scripts/dev/weak8_inprocess.py [N = 8] [lm_iterations = 10] [policy = spatial] [ranks-only]"""
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding  # noqa: E402
from tests import util  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
policy = sys.argv[3] if len(sys.argv) > 3 else "spatial"
skip_single = len(sys.argv) > 4 and sys.argv[4] == "ranks-only"
g = graphgen.generate(100000 * N, 100003 * N, odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=n_it)
t0 = time.time(); _, t1, s1, sum1 = P.solve(q, t, s); single_s = time.time() - t0
P.close()
parts = sharding.partition(g, N, policy)
group = capi.local_group_create(N)
out, err = [None] * N, []


def run(rank):
    try:
        Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], verbosity=1 if rank == 0 else 0, max_num_iterations=n_it)
        Pr.comm_init_local(rank, N, group)
        t0 = time.time(); res = Pr.solve(q, t, s); dt = time.time() - t0
        out[rank] = (res, Pr.sharding_stats().as_dict(), dt)
        Pr.comm_destroy(); Pr.close()
    except Exception as e:   # noqa: BLE001
        err.append(repr(e)); capi.local_group_abort(group)


th = [threading.Thread(target=run, args=(r,)) for r in range(N)]
[x.start() for x in th]; [x.join() for x in th]
capi.local_group_destroy(group)
if err:
    print(json.dumps({"error": err})); sys.exit(1)
sr = out[0][0][3]
d1 = [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
dr = [sr.iterations[k].step_is_successful for k in range(sr.num_logged)]
st = out[0][1]
print(json.dumps({"what": "bench.py's weak-scaling graph at N = %d (%d poses / %d edges) on %d in-process ranks of ONE GPU against the single handle: functional, not a timing" % (N, g.n_poses, g.n_odom + g.n_loops, N),
                  "policy": policy, "decisions_equal": d1 == dr, "decisions": dr, "cost_rel_diff_max": max(abs(sr.iterations[k].cost - sum1.iterations[k].cost) / sum1.iterations[k].cost for k in range(sum1.num_logged)),
                  "pcg_single": int(sum1.cg_iterations), "pcg_ranks": int(sr.cg_iterations), "pcg_retries": int(sr.pcg_retries),
                  "all_ranks_identical": all(np.array_equal(out[0][0][1], o[0][1]) and np.array_equal(out[0][0][2], o[0][2]) for o in out),
                  "t_max_abs_diff": float(np.abs(out[0][0][1] - t1).max()), "single_handle_s": single_s, "ranks_wall_s_sharing_one_gpu": max(o[2] for o in out),
                  "rank0_counters": {k: st[k] for k in ("keyframes_local", "keyframes_shared", "mg_levels", "mg_levels_distributed", "mg_rows_own", "mg_rows_total", "mg_blocks_own", "mg_blocks_total", "exchanges_per_mg_iteration",
                                                         "bytes_sent_per_mg_iteration", "bytes_round5_per_mg_iteration", "mg_setup_levels_own_rows", "mg_setup_exchanges", "mg_setup_blocks_own", "mg_setup_blocks_total",
                                                         "bytes_sent_per_mg_setup", "bytes_allreduce_replicated_setup")}}))
