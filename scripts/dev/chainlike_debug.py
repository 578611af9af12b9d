import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.generate(60000, 6000, odom_f_max=2, seed=7)
q, t, s = util.initial_state(g, True)
kw = {}
for item in (sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []):
    k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
P = util.pgo_problem(g, True, max_num_iterations=20, cg_max_iterations=200000, **kw)
_, _, _, sm = P.solve(q, t, s)
for k in range(sm.num_logged):
    it = sm.iterations[k]
    print(k, '%.9e' % it.cost, it.step_is_valid, it.step_is_successful, 'rho %.3f' % it.relative_decrease, 'cg', it.cg_iterations, 'res %.1e' % it.cg_residual, 'radius %.1e' % it.trust_region_radius)
print('mg iterations', sm.cg_iterations_multigrid, 'final %.9e' % sm.final_cost)
