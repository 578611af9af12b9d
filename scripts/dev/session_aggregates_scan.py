"""One 10-iteration solve of a session-sized graph per cap on the number of aggregates of the two-level method: device seconds, PCG iterations."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
sizes = [int(x) for x in sys.argv[1].split(',')]
caps = [int(x) for x in sys.argv[2].split(',')]
for n in sizes:
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    row = []
    for c in caps:
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, True, max_num_iterations=10, coarse_aggregates=c)
            _, _, _, sm = P.solve(q, t, s); P.close()
            best = sm.seconds_device if best is None else min(best, sm.seconds_device)
        row.append('%d: %.1f ms (cg %d, cost %.9e)' % (c, best * 1e3, sm.cg_iterations, sm.final_cost))
    print('%6d keyframes   ' % n + '   '.join(row), flush=True)
