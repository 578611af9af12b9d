#!/usr/bin/env python3
"""One handle on the N x C3 graph (bench.py's weak-scaling graph) with mg_smoothed_levels left to the library (-1: off beyond 500 000 keyframes) and forced to 1 / 0: seconds, PCG iterations, decisions.
  python scripts/dev/big_graph_smoothed.py [N = 8] [lm_iterations = 10]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402
from tests import util  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 10
g = graphgen.generate(100000 * N, 100003 * N, odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
for sm in (-1, 1, 0):
    P = util.pgo_problem(g, True, max_num_iterations=n_it, mg_smoothed_levels=sm)
    P.solve(q, t, s)
    t0 = time.time(); _, _, _, summ = P.solve(q, t, s); dt = time.time() - t0
    P.close()
    print("%d x C3, mg_smoothed_levels %2d: %.3f s, PCG %6d (multigrid %6d), decisions %s, final cost %.9e" % (N, sm, dt, summ.cg_iterations, summ.cg_iterations_multigrid,
          "".join(str(summ.iterations[k].step_is_successful) for k in range(summ.num_logged)), summ.final_cost), flush=True)
