"""C4 to Ceres' own convergence with the library defaults: iteration count, termination, per-iteration costs."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config(sys.argv[1] if len(sys.argv) > 1 else 'C4')
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=400)
_, _, sp, sm = P.solve(q, t, s)
for k in range(sm.num_logged):
    it = sm.iterations[k]
    print('it %2d cost %.12e rho %.3e ok %d cg %d' % (k, it.cost, it.relative_decrease, it.step_is_successful, it.cg_iterations))
print(sm.num_iterations, sm.message, sm.seconds_device, 'min |s - 0.5|', abs(sp - 0.5).min())
