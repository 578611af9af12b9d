"""One hard LM system of C3 (first step, radius 1e6) solved with the multigrid from the first iteration: the target of rocprofv3 --kernel-trace."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config('C3')
q, t, s = util.initial_state(g, True)
kw = {}
for item in (sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] else []):
    k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
P = util.pgo_problem(g, True, max_num_iterations=1, initial_trust_region_radius=1e6, cg_early_tolerance=0.0, cg_mid_tolerance=0.0, mg_min_keyframes=1, mg_switch_iterations=0, **kw)
_, _, _, sm = P.solve(q, t, s)
print('cg', sm.iterations[1].cg_iterations, 'ms', sm.iterations[1].seconds * 1e3)
