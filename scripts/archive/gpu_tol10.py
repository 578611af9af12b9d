import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
g = graphgen.config('C3')
q,t,s = util.initial_state(g, True)
ref=None
for tol in [1e-13, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5]:
    P = util.pgo_problem(g, True, cg_rel_tolerance=tol, cg_max_iterations=20000, cg_check_every=50)
    t0=time.time(); qq,tt,ss,summ = P.solve(q,t,s); dt=time.time()-t0
    its=[summ.iterations[k] for k in range(summ.num_logged)]
    if ref is None: ref=(summ.final_cost, tt, ss, [i.cost for i in its])
    print('tol %.0e cg %6d time %.3f final %.12e rel-diff-vs-1e-13 %.2e  dt %.2e ds %.2e  accept %s' % (tol, summ.cg_iterations, summ.seconds_device, summ.final_cost, abs(summ.final_cost-ref[0])/ref[0], np.abs(tt-ref[1]).max(), np.abs(ss-ref[2]).max(), ''.join(str(i.step_is_successful) for i in its)))
    P.close()
