import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
sw = name != 'C2'
t0=time.time(); g = graphgen.config(name); print('gen', time.time()-t0, g.n_poses, g.n_odom, g.n_loops)
P = util.pgo_problem(g, sw, verbosity=1, cg_rel_tolerance=float(sys.argv[2]) if len(sys.argv)>2 else 1e-6, cg_max_iterations=int(sys.argv[3]) if len(sys.argv)>3 else 3000, cg_check_every=50)
q,t,s = util.initial_state(g, sw)
t0=time.time(); P.solve_begin(q,t,s); print('begin', time.time()-t0)
for w in (0,1,2,3):
    ms, by = P.time_kernel(w, 20); print('kernel', w, 'ms', ms, 'GB/s', by/ms/1e6, 'bytes', by)
t0=time.time()
for i in range(10):
    if P.lm_step(): break
qq,tt,ss,summ = P.solve_end()
print('total', time.time()-t0, 'iters', summ.num_iterations, 'cost', summ.initial_cost, '->', summ.final_cost, summ.message, 'cg', summ.cg_iterations)
if sw: print('s inliers', ss[g.loop_is_outlier==0].mean(), 'outliers', ss[g.loop_is_outlier==1].mean())
print('pos err init', np.linalg.norm(g.init_t-g.truth_t,axis=1).max(), 'final', np.linalg.norm(tt.reshape(-1,3)-g.truth_t,axis=1).max())
