"""One multigrid-preconditioned PCG iteration (pgo_time_kernel 6) and its level kernels alone (7) on a config, after `steps` LM iterations; also the 20-step solve time."""
import sys, time; sys.path.insert(0, '.')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
g = graphgen.config(name); q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=10 ** 6)
P.solve_begin(q, t, s)
for _ in range(12):
    P.lm_step(ignore_termination=True)
a = [P.time_kernel(6, 100)[0] for _ in range(3)]
b = [P.time_kernel(7, 100)[0] for _ in range(3)]
c = [P.time_kernel(2, 100)[0] for _ in range(3)]
P.solve_end(); P.close()
P = util.pgo_problem(g, True, max_num_iterations=20, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
P.solve(q, t, s)
_, _, _, sm = P.solve(q, t, s)
P.close()
print('%s: multigrid PCG iteration %.2f us (min of 3), level kernels %.2f us, block-Jacobi iteration %.2f us | 20 LM steps %.4f s device, %d PCG iterations (%d multigrid)' % (
    name, min(a) * 1e3, min(b) * 1e3, min(c) * 1e3, sm.seconds_device, sm.cg_iterations, sm.cg_iterations_multigrid))
