"""relative_decrease of every LM iteration for the benchmark configs (to tune the early-rejection thresholds)"""
import sys; sys.path.insert(0, '.')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for name in sys.argv[1:]:
    g = graphgen.config(name); q, t, s = util.initial_state(g, name != "C2")
    P = util.pgo_problem(g, name != "C2", cg_early_tolerance=0.0, cg_mid_tolerance=0.0)
    _, _, _, sm = P.solve(q, t, s)
    print(name, ' '.join('%d:%+.3f(%d)' % (i.step_is_successful, i.relative_decrease, i.cg_iterations) for i in [sm.iterations[k] for k in range(1, sm.num_logged)]), flush=True)
    P.close()
