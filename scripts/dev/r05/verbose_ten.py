import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1]
kw = {}
for item in (sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] else []):
    k, x = item.split('='); kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
kw.setdefault('verbosity', 1)
P = util.pgo_problem(g, True, max_num_iterations=10, **kw)
_, _, _, sm = P.solve(q, t, s); P.close()
print(name, kw, 'device s', sm.seconds_device, 'cg', sm.cg_iterations, 'mg', sm.cg_iterations_multigrid)
