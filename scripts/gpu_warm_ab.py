import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
g = graphgen.config('C3'); q,t,s = util.initial_state(g, True)
ref=None
for warm in (0, 1):
    P = util.pgo_problem(g, True, cg_warm_start=warm)
    qq,tt,ss,summ = P.solve(q,t,s)
    its=[summ.iterations[k] for k in range(summ.num_logged)]
    if ref is None: ref=summ.final_cost
    print('warm', warm, 'cg', summ.cg_iterations, [i.cg_iterations for i in its[1:]], 'dev %.3fs' % summ.seconds_device, 'final %.12e rel diff %.2e' % (summ.final_cost, abs(summ.final_cost-ref)/ref), ''.join(str(i.step_is_successful) for i in its))
    P.close()
