"""C5 (1M keyframes), three LM steps with the multigrid from the first iteration: for rocprofv3 --kernel-trace --stats (per-kernel times at the bandwidth-bound size)."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config(sys.argv[1] if len(sys.argv) > 1 else 'C5')
q, t, s = util.initial_state(g, True)
kw = {}
for item in (sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] else []):
    k, x = item.split('='); kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
P = util.pgo_problem(g, True, max_num_iterations=3, mg_switch_iterations=0, cg_use_graph=0, **kw)
_, _, _, sm = P.solve(q, t, s); P.close()
print(sm.seconds_device, sm.cg_iterations)
