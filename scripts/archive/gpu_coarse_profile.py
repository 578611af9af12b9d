"""One mid-size solve with the two-level preconditioner on — run under `rocprofv3 --kernel-trace --stats` to see where K6 spends its time."""
import os, sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

g = graphgen.generate(6918, 1382, odom_f_max=3, seed=12, outlier_frac=0.3)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, cg_use_graph=int(os.environ.get('PGO_GRAPH', '1')), cg_check_every=int(os.environ.get('PGO_EVERY', '24')))
P.solve(q, t, s)
_, _, _, sm = P.solve(q, t, s)
print('device seconds %.4f, PCG iterations %d' % (sm.seconds_device, sm.cg_iterations))
P.close()
