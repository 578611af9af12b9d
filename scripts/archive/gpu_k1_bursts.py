import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config("C3")
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True)
P.solve_begin(q, t, s)
print([round(P.time_kernel(0, 50)[0] * 1e3, 2) for _ in range(6)])
for _ in range(3): P.lm_step()
print([round(P.time_kernel(0, 50)[0] * 1e3, 2) for _ in range(6)])
print([round(P.time_kernel(0, 500)[0] * 1e3, 2) for _ in range(3)])
P.solve_end(); P.close()
