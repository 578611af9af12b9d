"""C5 (1M keyframes / 3M edges), 10 LM iterations: plain aggregation above level 1 (the default beyond 500 000 keyframes) against one smoothed transition."""
import sys, time
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config('C5')
q, t, s = util.initial_state(g, True)
for sm_levels in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,1').split(',')]:
    P = util.pgo_problem(g, True, max_num_iterations=10, mg_smoothed_levels=sm_levels)
    _, _, _, sm = P.solve(q, t, s)
    print('mg_smoothed_levels %d: device %.3f s, PCG %d (multigrid %d), final cost %.9e' % (sm_levels, sm.seconds_device, sm.cg_iterations, sm.cg_iterations_multigrid, sm.final_cost), flush=True)
    P.close()
