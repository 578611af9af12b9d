import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config("C3"); q, t, s = util.initial_state(g, True)
kw = {}
for item in (sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []):
    k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
P = util.pgo_problem(g, True, verbosity=1, **kw)
for rep in range(2):
    _, _, _, sm = P.solve(q, t, s)
    print('solve', rep, '%.12e' % sm.final_cost, [sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], 'mg', sm.cg_iterations_multigrid)
