"""C3, 20 LM steps with the library defaults, one solve (after one warm-up solve when 'warm' is given): the target of rocprofv3 --kernel-trace for the idle-gap analysis."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=20)
_, _, _, sm = P.solve(q, t, s)
print(name, 'device s', sm.seconds_device, 'cg', sm.cg_iterations, 'mg', sm.cg_iterations_multigrid)
P.close()
