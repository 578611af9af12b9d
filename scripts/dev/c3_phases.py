"""Host-side phase times of every LM step of C3's 20-step solve (second solve on the handle, verbosity 2)."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=20, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
P.solve(q, t, s)
P.set_options(verbosity=2) if hasattr(P, 'set_options') else None
_, _, _, sm = P.solve(q, t, s)
print(name, 'device s', sm.seconds_device, 'cg', sm.cg_iterations, 'mg', sm.cg_iterations_multigrid)
P.close()
