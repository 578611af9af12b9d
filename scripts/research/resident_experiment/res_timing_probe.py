import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
n = int(sys.argv[1])
g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True, max_num_iterations=1, resident_max_keyframes=1 << 30, coarse_aggregates=0)
P.solve(q, t, s); P.close()
