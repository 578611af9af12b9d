"""Session-sized graphs: how many coarse aggregates?  Per size and coarse_aggregates: device ms, LM iterations, PCG iterations."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '300,500,800,1500,3000,6000,12000').split(',')]
aggs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '64,128,256,512').split(',')]
extra = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
for n in sizes:
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    row = []
    for a in aggs:
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, True, max_num_iterations=10, coarse_aggregates=a, **extra)
            _, _, _, sm = P.solve(q, t, s)
            P.close()
            if best is None or sm.seconds_device < best.seconds_device: best = sm
        row.append('%d: %.1f ms (cg %d)' % (a, best.seconds_device * 1e3, best.cg_iterations))
    print('%6d keyframes   ' % n + '   '.join(row), flush=True)
