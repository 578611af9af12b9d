"""Set-up cost of the multigrid operators and of the two-level coarse operator: C3 20 LM steps (seconds, with 7 operator builds) and a 3 000-keyframe session-structured solve (two-level method),
plus the per-kernel averages of the set-up kernels from the library's own verbose timing where available."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config('C3'); q, t, s = util.initial_state(g, True)
best = 1e9
for _ in range(3):
    P = util.pgo_problem(g, True, max_num_iterations=20, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    P.solve(q, t, s)
    _, _, _, sm = P.solve(q, t, s); P.close()
    best = min(best, sm.seconds_device)
g2 = graphgen.generate(3000, 600, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0)); q2, t2, s2 = util.initial_state(g2, True)
b2 = 1e9
for _ in range(3):
    P = util.pgo_problem(g2, True)
    P.solve(q2, t2, s2)
    _, _, _, sm2 = P.solve(q2, t2, s2); P.close()
    b2 = min(b2, sm2.seconds_device)
print('C3 20 LM steps %.4f s (best of 3) | 3000-keyframe session graph 10 LM steps %.4f s, %d PCG iterations' % (best, b2, sm2.cg_iterations))
