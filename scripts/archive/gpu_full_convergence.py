"""Soak: C3 run to full convergence with library defaults (function tolerance 1e-6 as Ceres) and to 1e-8."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
g = graphgen.config('C3'); q,t,s = util.initial_state(g, True)
for ftol in (1e-6, 1e-8):
    P = util.pgo_problem(g, True, max_num_iterations=200, function_tolerance=ftol)
    t0=time.time(); qq,tt,ss,summ = P.solve(q,t,s); dt=time.time()-t0
    its=[summ.iterations[k] for k in range(summ.num_logged)]
    print('ftol %g: LM %d (succ %d) cg %d time %.2fs final cost %.10e  %s' % (ftol, summ.num_iterations, summ.num_successful_steps, summ.cg_iterations, dt, summ.final_cost, summ.message.decode()))
    print('   cg/it', [i.cg_iterations for i in its[1:]])
    print('   inlier s mean %.4f outlier s mean %.5f  max pos err vs truth %.2f (init %.2f)' % (ss[g.loop_is_outlier==0].mean(), ss[g.loop_is_outlier==1].mean(), np.linalg.norm(tt.reshape(-1,3)-g.truth_t,axis=1).max(), np.linalg.norm(g.init_t-g.truth_t,axis=1).max()))
    P.close()
