"""Session-sized graph: per-LM-step seconds vs PCG iterations (defaults), to split a solve into per-step fixed cost and per-iteration cost."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for n in [int(x) for x in sys.argv[1].split(',')]:
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    for rep in range(2):
        P = util.pgo_problem(g, True, max_num_iterations=10, verbosity=1 if rep else 0)
        _, _, _, sm = P.solve(q, t, s)
        P.close()
    its = np.array([sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], float)
    sec = np.array([sm.iterations[k].seconds for k in range(1, sm.num_logged)])
    A = np.stack([np.ones_like(its), its], 1)
    coef, *_ = np.linalg.lstsq(A, sec, rcond=None)
    print('%d keyframes: step seconds ~ %.3f ms + %.2f us x iterations; device total %.2f ms, iteration 0 %.2f ms' % (n, coef[0] * 1e3, coef[1] * 1e6, sm.seconds_device * 1e3, sm.iterations[0].seconds * 1e3))
    print('   ', ' '.join('%d:%.1fms' % (a, b * 1e3) for a, b in zip(its, sec)))
