#!/usr/bin/env python3
"""Round 6: the multigrid's DISTRIBUTED SET-UP (pgo_options.mg_dist_setup = 1) against the replicated one (= 0) with in-process ranks on one GPU: after ONE LM iteration from the same
state (multigrid from the first PCG iteration: one set-up on identical inputs) every rank's level operators — its rows' blocks, their fp32 copy, block-Jacobi inverses, R^T, R, the dense
inverse (pgo_mg_level_norms) — must agree to summation order; then full solves: same decisions, costs to 1e-8, PCG counts within 3 %.
  python scripts/gpu_dist_setup_check.py [n_keyframes] [ranks] [dist_min_rows] [smoothed_levels]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding  # noqa: E402
from tests import util  # noqa: E402


def run_ranks(g, world, policy, q, t, s, opts, norms=False):
    parts = sharding.partition(g, world, policy)
    group = capi.local_group_create(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **opts)
            Pr.comm_init_local(rank, world, group)
            t0 = time.time()
            res = Pr.solve(q, t, s)
            dt = time.time() - t0
            st = Pr.sharding_stats().as_dict()
            nr = [Pr.mg_level_norms(l + 1) for l in range(st["mg_levels"])] if norms and st["mg_levels"] > 0 else []
            out[rank] = res + (st, nr, dt)
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
    if err:
        raise RuntimeError(err[0])
    return out


def check(n, world, dist_min, smoothed, policy="spatial", loops=None, f=2, verbose=True):
    g = graphgen.generate(n, loops if loops is not None else n, odom_f_max=f, apply_yaw_weight=bool(f == 5), seed=3)
    q, t, s = util.initial_state(g, True)
    base = dict(mg_min_keyframes=1000, mg_min_keyframes_switchable=1000, mg_switch_iterations=0, mg_smoothed_fine=0, mg_smoothed_levels=smoothed, mg_dist_min_rows=dist_min, cg_rel_tolerance=1e-11)
    worst = 0.0
    A = run_ranks(g, world, policy, q, t, s, dict(base, max_num_iterations=1, mg_dist_setup=1), norms=True)
    B = run_ranks(g, world, policy, q, t, s, dict(base, max_num_iterations=1, mg_dist_setup=0), norms=True)
    names = ["blocks", "fp32 blocks", "Dinv", "R^T", "R", "dense inverse"]
    for r in range(world):
        for l, (na, nb) in enumerate(zip(A[r][5], B[r][5])):
            for k in range(6):
                if nb[k] == 0.0 and na[k] == 0.0:
                    continue
                d = abs(na[k] - nb[k]) / max(abs(nb[k]), 1e-300)
                worst = max(worst, d)
                if verbose and (d > 1e-9 or r == 0):
                    print("  rank %d level %d %-13s distributed %.15e replicated %.15e  rel diff %.1e" % (r, l + 1, names[k], na[k], nb[k], d))
    st = A[0][4]
    print("%d keyframes, %d ranks (%s), dist_min_rows %d, smoothed levels %d: levels %d (%d distributed); one LM iteration: PCG %d (distributed set-up) : %d (replicated); worst relative difference of the level norms %.1e" % (
        n, world, policy, dist_min, smoothed, st["mg_levels"], st["mg_levels_distributed"], A[0][3].cg_iterations, B[0][3].cg_iterations, worst))
    A = run_ranks(g, world, policy, q, t, s, dict(base, max_num_iterations=8, mg_dist_setup=1))
    B = run_ranks(g, world, policy, q, t, s, dict(base, max_num_iterations=8, mg_dist_setup=0))
    sa, sb = A[0][3], B[0][3]
    seq_a = [sa.iterations[k].step_is_successful for k in range(sa.num_logged)]
    seq_b = [sb.iterations[k].step_is_successful for k in range(sb.num_logged)]
    dev = max(abs(sa.iterations[k].cost - sb.iterations[k].cost) / max(sb.iterations[k].cost, 1e-300) for k in range(min(sa.num_logged, sb.num_logged)))
    same_bits = all(np.array_equal(A[0][1], o[1]) and np.array_equal(A[0][2], o[2]) for o in A)
    ok = seq_a == seq_b and dev <= 1e-8 and worst <= 1e-9 and same_bits and sa.pcg_retries == 0 and abs(sa.cg_iterations - sb.cg_iterations) <= 0.03 * sb.cg_iterations
    print("  8 LM iterations: decisions %s, cost deviation %.1e, PCG %d : %d, ranks identical %s, retries %d, time %.3f : %.3f s -> %s" % (
        "equal" if seq_a == seq_b else "DIFFERENT", dev, sa.cg_iterations, sb.cg_iterations, same_bits, sa.pcg_retries, max(o[6] for o in A), max(o[6] for o in B), "ok" if ok else "MISMATCH"), flush=True)
    return ok


if __name__ == "__main__":
    if len(sys.argv) > 1:
        ok = check(int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 300, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
        sys.exit(0 if ok else 1)
    bad = 0
    for (n, world, dm, sm, pol, f) in [(6000, 3, 300, 1, "spatial", 2), (6000, 4, 1, 0, "spatial", 2), (8000, 4, 200, 2, "spatial", 2), (12000, 5, 64, 1, "spatial", 5), (6000, 3, 300, 1, "chain", 2),
                                       (20000, 8, 1000, 1, "spatial", 2), (9000, 2, 8192, 1, "spatial", 2)]:
        bad += 0 if check(n, world, dm, sm, pol, f=f, verbose=False) else 1
    print("mismatches: %d" % bad)
    sys.exit(1 if bad else 0)
