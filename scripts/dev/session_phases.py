"""Host-side phase times of every LM step of one session-sized solve (verbosity 2 prints them): where the fixed cost per LM iteration goes."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
n = int(sys.argv[1])
g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
q, t, s = util.initial_state(g, True)
for rep in range(2):
    P = util.pgo_problem(g, True, max_num_iterations=10, verbosity=2 if rep else 0)
    _, _, _, sm = P.solve(q, t, s); P.close()
print(n, sm.seconds_device, sm.cg_iterations)
