import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
g = graphgen.generate(n, n, odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
for frac in (0.0, 0.005):
    P = util.pgo_problem(g, True, max_num_iterations=16, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, mg_switch_iterations=0, mg_regroup_fraction=frac, verbosity=1)
    _, t1, s1, sm = P.solve(q, t, s)
    print('fraction', frac, [sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], ['%.2e' % sm.iterations[k].cg_residual for k in range(12, sm.num_logged)], '%.12e' % sm.final_cost)
print('moved vs 0.9801 by > 0.5:', int((np.abs(s1 ** 2 - 0.9801) > 0.5).sum()), 'of', len(s1), 'edges', g.n_odom + g.n_loops)
print([sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)])
