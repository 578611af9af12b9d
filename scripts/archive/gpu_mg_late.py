"""Late-state linear systems: K LM steps with block-Jacobi, then ONE step from that state at several radii, block-Jacobi vs multigrid."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'; K = int(sys.argv[2]) if len(sys.argv) > 2 else 17
g = graphgen.config(name) if name.startswith('C') else graphgen.generate(int(name), int(name), odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=K, mg_min_keyframes=0, coarse_aggregates=0)
q, t, s, sm = P.solve(q, t, s); P.close()
q = q.reshape(-1, 4); t = t.reshape(-1, 3)
print('state after %d LM steps: cost %.6e, switches < 0.5: %d of %d, in (0.05, 0.9): %d' % (K, sm.final_cost, (s < 0.5).sum(), len(s), ((s > 0.05) & (s < 0.9)).sum()), flush=True)
for radius in [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '1e4,1e5,1e6').split(',')]:
    row = []
    for kw in (dict(mg_min_keyframes=0, coarse_aggregates=0), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_correction_scale=1.0), dict(mg_min_keyframes=1, mg_switch_iterations=0)):
        P = util.pgo_problem(g, True, max_num_iterations=1, initial_trust_region_radius=radius, cg_early_tolerance=0.0, cg_mid_tolerance=0.0, cg_max_iterations=200000, **kw)
        _, _, _, sm1 = P.solve(q, t, s)
        P.close()
        row.append((sm1.iterations[1].cg_iterations, sm1.iterations[1].seconds))
    print('%s late, radius %g: block-Jacobi %d its (%.1f ms)   multigrid scale 1.0: %d its (%.1f ms)   scale 1.6: %d its (%.1f ms)' % (name, radius, row[0][0], row[0][1] * 1e3, row[1][0], row[1][1] * 1e3, row[2][0], row[2][1] * 1e3), flush=True)
