import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config('C3')
q, t, s = util.initial_state(g, True)
import os
P = util.pgo_problem(g, True, max_num_iterations=20, verbosity=int(os.environ.get("VERB", "0")), cg_use_graph=int(os.environ.get("USE_GRAPH", "1")))
_, _, _, sm = P.solve(q, t, s); P.close()
print(sm.seconds_device, sm.cg_iterations)
