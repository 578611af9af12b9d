"""Fresh-process reproduction of the C2 two-level trajectory: prints the whole pgo_iteration log of a solve with the
library defaults and of one with plain block-Jacobi, plus a sha256 over every output array, so that runs in separate processes / on separate
boxes can be compared bit for bit.  Usage: python scripts/dev/repro_c2.py [config] [verbosity]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solve_keyframe_pose_graph_amd import capi, graphgen  # noqa: E402
from tests import util  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
verb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
switchable = name != "C2"
g = graphgen.config(name)


def run(**kw):
    q, t, s = util.initial_state(g, switchable)
    P = util.pgo_problem(g, switchable, verbosity=verb, **kw)
    out = P.solve(q, t, s)
    P.close()
    return out


def show(tag, out):
    q, t, s, sm = out
    h = hashlib.sha256(q.tobytes() + t.tobytes() + s.tobytes()).hexdigest()[:16]
    print("%s: iterations %d ok %d bad %d cg %d final %.17g sha %s" % (tag, sm.num_iterations, sm.num_successful_steps, sm.num_unsuccessful_steps, sm.cg_iterations, sm.final_cost, h))
    for k in range(sm.num_logged):
        it = sm.iterations[k]
        print("   %s it %2d valid %d ok %d cost %.17g dcost %.6e model %.6e rho %.6e radius %.6e cg %d res %.3e reason %s" % (
            tag, it.iteration, it.step_is_valid, it.step_is_successful, it.cost, it.cost_change, it.model_cost_change, it.relative_decrease, it.trust_region_radius,
            it.cg_iterations, it.cg_residual, getattr(it, "reason", "-")))


show("default", run())
show("plain  ", run(coarse_aggregates=0, cg_max_iterations=200000))
