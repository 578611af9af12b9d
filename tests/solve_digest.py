"""Helper of tests/test_gpu_determinism.py, run as `python -m tests.solve_digest <graph> [opt=value ...]` in a process of its own: solves one graph through the
C-ABI on cuda:0 and prints ONE JSON line with a sha256 over every output array and the whole pgo_iteration log (costs as hex floats), so that solves made in
different processes (and under PGO_DEBUG_POISON=1) can be compared bit for bit."""
import hashlib
import json
import sys

import numpy as np


def graph(name):
    from solve_keyframe_pose_graph_amd import graphgen
    if name in ("C1", "C1F5", "C2", "C3", "C4"):
        return graphgen.config(name), name != "C2"
    if name == "G4000":          # below mg_min_keyframes_switchable (5 000 since round 6): the two-level method, switchable loop closures with outliers
        return graphgen.generate(4000, 2000, odom_f_max=2, seed=21, outlier_frac=0.1), True
    if name == "G6000":          # (rounds 3-5: two-level method; round 6: multigrid with the smoothed keyframe transition decided by the density of its levels)
        return graphgen.generate(6000, 3000, odom_f_max=2, seed=21, outlier_frac=0.1), True
    if name == "G12000":         # above mg_min_keyframes_switchable: the hybrid block-Jacobi / multigrid schedule with its regroup worker
        return graphgen.generate(12000, 12000, odom_f_max=2, seed=3), True
    if name == "P9000":          # plain loops, chain-like (C2's regime): the multigrid since round 6 (two-level method before)
        return graphgen.generate(9000, 900, odom_f_max=1, seed=2, outlier_frac=0.0), False
    raise SystemExit("unknown graph %r" % name)


def digest(name, **opt):
    from solve_keyframe_pose_graph_amd import capi
    from tests import util
    g, switchable = graph(name)
    q, t, s = util.initial_state(g, switchable)
    P = util.pgo_problem(g, switchable, **opt)
    qo, to, so, sm = P.solve(q, t, s)
    P.close()
    h = hashlib.sha256(np.ascontiguousarray(qo).tobytes() + np.ascontiguousarray(to).tobytes() + np.ascontiguousarray(so).tobytes()).hexdigest()
    log = [[it.iteration, it.step_is_valid, it.step_is_successful, capi.STEP_REASONS[it.reason], it.preconditioner, it.cg_iterations, float(it.cost).hex(), float(it.relative_decrease).hex()]
           for it in (sm.iterations[k] for k in range(sm.num_logged))]
    return {"graph": name, "sha256": h, "final_cost": float(sm.final_cost).hex(), "cg_iterations": int(sm.cg_iterations), "pcg_retries": int(sm.pcg_retries), "log": log}


if __name__ == "__main__":
    kw = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    print("DIGEST " + json.dumps(digest(sys.argv[1], **kw)))
