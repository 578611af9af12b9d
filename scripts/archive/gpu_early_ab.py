"""A/B of the early-rejection pauses of the PCG (10 LM iterations): iterates must be identical, CG iterations fewer."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
g = graphgen.config(sys.argv[1] if len(sys.argv) > 1 else 'C3'); q,t,s = util.initial_state(g, True)
ref=None
for early, mid, mid_rho in ((0.0, 0.0, -0.05), (0.0, 1e-4, -0.5), (1e-2, 0.0, -0.05), (1e-2, 1e-4, -0.05), (1e-2, 1e-4, -0.02), (1e-1, 1e-3, -0.05)):
    P = util.pgo_problem(g, True, cg_early_tolerance=early, cg_mid_tolerance=mid, cg_mid_reject_rho=mid_rho)
    P.solve(q,t,s)
    qq,tt,ss,summ = P.solve(q,t,s)
    its=[summ.iterations[k] for k in range(summ.num_logged)]
    if ref is None: ref=(summ.final_cost, tt)
    print('early %g mid %g (rho %g)' % (early, mid, mid_rho), 'cg', summ.cg_iterations, [i.cg_iterations for i in its[1:]], 'dev %.3fs' % summ.seconds_device, 'final %.12e rel diff %.2e dt %.2e' % (summ.final_cost, abs(summ.final_cost-ref[0])/ref[0], np.abs(tt-ref[1]).max()), ''.join(str(i.step_is_successful) for i in its))
    P.close()
