"""C3, state after two LM steps, then 4 x 60 multigrid-preconditioned PCG iterations through pgo_time_kernel(6) with the given options (rocprofv3 target: per-kernel times of one iteration)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
kw = {}
for item in (sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []):
    k, x = item.split('='); kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
g = graphgen.config('C3')
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, **kw)
P.solve_begin(q, t, s)
for _ in range(2): P.lm_step(ignore_termination=True)
best = min(P.time_kernel(6, 60)[0] for _ in range(3))
ms, by = P.time_kernel(6, 60)
P.solve_end(); P.close()
print('multigrid PCG iteration %.2f us, %.1f MB by the design count' % (best * 1e3, by / 1e6), flush=True)
