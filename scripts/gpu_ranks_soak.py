#!/usr/bin/env python3
"""Round 6 soak of the distributed multi-rank solver: random graphs (1 500 - 9 000 keyframes, loop density, odometry policy, outliers), 2 - 7 in-process ranks on ONE GPU, the three
partition policies (+ an idle rank now and then), 0 - 2 smoothed transitions, every level distributed or the library's threshold, the set-up distributed (3 of 4) or replicated — each solved by the ranks and by a single handle with
the same options: same accept/reject sequence, costs within 1e-7, no PCG retry, identical results on every rank.  Prints one line per case and a summary; exit code 1 on any mismatch.
  python scripts/gpu_ranks_soak.py [cases] [seed] [only this case] [its set-up forced: 0 replicated / 1 distributed]"""
import sys
import threading

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding  # noqa: E402
from tests import util  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
force_setup = int(sys.argv[4]) if len(sys.argv) > 4 else -1
bad = 0
for case in range(n_cases):
    n = int(rng.integers(1500, 9000))
    loops = int(n * rng.choice([0.1, 0.3, 0.5, 1.0]))
    f = int(rng.choice([1, 2, 3, 5]))
    world = int(rng.integers(2, 8))
    policy = str(rng.choice(["spatial", "spatial", "chain", "contiguous"]))
    idle = bool(rng.random() < 0.15) and world >= 3
    opts = dict(mg_min_keyframes=1000, mg_min_keyframes_switchable=1000, mg_switch_iterations=int(rng.choice([0, 0, 60])), max_num_iterations=int(rng.integers(4, 9)), mg_smoothed_fine=0,
                mg_smoothed_levels=int(rng.choice([0, 1, 1, 2])), mg_dense_max_nodes=int(rng.choice([16, 48, 128])), cg_rel_tolerance=1e-11)
    dist_min = int(rng.choice([1, 1, 200, 8192]))
    dist_setup = int(rng.choice([1, 1, 1, 0]))      # (the multigrid's set-up distributed like its cycle — the default — or rounds 3-5's replicated one)
    gseed, gout = int(rng.integers(1, 10 ** 6)), float(rng.choice([0.0, 0.1, 0.3]))
    if only >= 0 and case != only:      # (python scripts/gpu_ranks_soak.py <cases> <seed> <case> [0|1]: that one case again, the set-up's mode forced)
        continue
    if only >= 0 and force_setup >= 0:
        dist_setup = force_setup
    g = graphgen.generate(n, loops, odom_f_max=f, apply_yaw_weight=bool(f == 5), seed=gseed, outlier_frac=gout)
    q, t, s = util.initial_state(g, True)
    P = util.pgo_problem(g, True, **opts)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    parts = (sharding.partition(g, world - 1, policy) + [lambda kind, m: np.arange(0)]) if idle else sharding.partition(g, world, policy)
    group = capi.local_group_create(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], mg_dist_min_rows=dist_min, mg_dist_setup=dist_setup, **opts)
            Pr.comm_init_local(rank, world, group)
            out[rank] = Pr.solve(q, t, s) + (Pr.sharding_stats().as_dict(),)
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
    why = []
    if err:
        why.append("error %s" % err[:1])
    else:
        sumr = out[0][3]
        if [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] != [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]:
            why.append("decisions differ")
        else:
            dev = max(abs(sumr.iterations[k].cost - sum1.iterations[k].cost) / max(sum1.iterations[k].cost, 1e-300) for k in range(sum1.num_logged))
            if dev > 1e-7:
                why.append("cost deviates by %.1e" % dev)
        if any(o[3].pcg_retries for o in out):
            why.append("PCG retried")
        if not all(np.array_equal(out[0][1], o[1]) and np.array_equal(out[0][2], o[2]) for o in out):
            why.append("ranks differ")
        if np.abs(out[0][1] - t1).max() > 1e-5:
            why.append("positions deviate by %.1e" % np.abs(out[0][1] - t1).max())
    bad += bool(why)
    st = out[0][4] if out[0] else {}
    print("case %2d: %5d keyframes %5d loops f=%d outliers | %d ranks %-10s%s smoothed %d dense<=%3d dist_min %4d set-up %s | single cg %6d (mg %6d)  ranks cg %6d  levels %s/%s  exchanges/it %s | %s" % (
        case, n, loops, f, world, policy, " +idle" if idle else "", opts["mg_smoothed_levels"], opts["mg_dense_max_nodes"], dist_min, "distributed" if dist_setup else "replicated ", sum1.cg_iterations, sum1.cg_iterations_multigrid,
        out[0][3].cg_iterations if out[0] else -1, st.get("mg_levels_distributed"), st.get("mg_levels"), st.get("exchanges_per_mg_iteration"), "ok" if not why else "MISMATCH: " + "; ".join(why)), flush=True)
print("%d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
