"""Session-sized graphs: the PCG as separate kernels (block-Jacobi; two-level) vs the resident kernel (pgo_resident_kernels.hpp).  Per size:
device seconds, LM iterations, PCG iterations, final cost."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '400,1000,2000,3000,6000').split(',')]
turn = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
variants = [('separate kernels, block-Jacobi', dict(resident_max_keyframes=0, coarse_aggregates=0)),
            ('separate kernels, two-level policy', dict(resident_max_keyframes=0)),
            ('resident, block-Jacobi', dict(resident_max_keyframes=1 << 30, coarse_aggregates=0)),
            ('defaults', dict())]
for n in sizes:
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=turn))
    q, t, s = util.initial_state(g, True)
    for name, kw in variants:
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, True, max_num_iterations=10, **kw)
            _, _, _, sm = P.solve(q, t, s)
            P.close()
            if best is None or sm.seconds_device < best.seconds_device: best = sm
        its = [best.iterations[k].cg_iterations for k in range(1, best.num_logged)]
        print('%6d keyframes  %-36s dev %8.2f ms  lm %2d  cg %6d  (%.1f us/cg)  cost %.12e' % (n, name, best.seconds_device * 1e3, best.num_iterations, sum(its), 1e6 * best.seconds_device / max(sum(its), 1), best.final_cost), flush=True)
