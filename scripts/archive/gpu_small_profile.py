"""Per-LM-iteration time of one small graph (CFG=C1F5 / C1 / ..., or N=<keyframes> for a session-like graph); run under
`rocprofv3 --kernel-trace --stats` to see how much of the device time is kernels and how much is gaps between them."""
import os, sys, time
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
if os.environ.get("N"):
    n = int(os.environ["N"])
    g = graphgen.generate(n, max(2, n // 6), odom_f_max=5, apply_yaw_weight=1, seed=n, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
else:
    g = graphgen.config(os.environ.get("CFG", "C1F5"))
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, cg_use_graph=int(os.environ.get('PGO_GRAPH', '1')))
P.solve(q, t, s)
t0 = time.perf_counter()
_, _, _, sm = P.solve(q, t, s)
print('wall %.2f ms device %.2f ms, LM %d, PCG %d' % ((time.perf_counter() - t0) * 1e3, sm.seconds_device * 1e3, sm.num_iterations, sm.cg_iterations))
for k in range(sm.num_logged): print('  it', k, 'ms %.2f' % (sm.iterations[k].seconds * 1e3), 'cg', sm.iterations[k].cg_iterations)
P.close()
