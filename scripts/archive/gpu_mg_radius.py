"""First LM step of C3 (or a generated graph) at several trust-region radii, block-Jacobi vs multigrid from the first iteration: PCG iterations.
Comparable one to one with scripts/research/amg_probe.py (same linearisation: the initial state)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
g = graphgen.config(name) if name.startswith('C') else graphgen.generate(int(name), int(name), odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
for radius in [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1e4,9e4,1e6,1e8').split(',')]:
    row = []
    for kw in (dict(mg_min_keyframes=0, coarse_aggregates=0), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_correction_scale=1.0), dict(mg_min_keyframes=1, mg_switch_iterations=0)):
        P = util.pgo_problem(g, True, max_num_iterations=1, initial_trust_region_radius=radius, cg_early_tolerance=0.0, cg_mid_tolerance=0.0, cg_max_iterations=200000, **kw)
        _, _, _, sm = P.solve(q, t, s)
        P.close()
        row.append((sm.iterations[1].cg_iterations, sm.iterations[1].seconds))
    print('%s radius %g: block-Jacobi %d its (%.1f ms)   multigrid scale 1.0: %d its (%.1f ms)   scale 1.6: %d its (%.1f ms)' % (name, radius, row[0][0], row[0][1] * 1e3, row[1][0], row[1][1] * 1e3, row[2][0], row[2][1] * 1e3), flush=True)
