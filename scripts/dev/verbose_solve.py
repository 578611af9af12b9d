import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = graphgen.config(name); q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=iters, verbosity=2, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
_, _, _, sm = P.solve(q, t, s); P.close()
print(sm.seconds_device, sm.cg_iterations)
