import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
t0=time.time(); g = graphgen.config('C5'); print('gen %.1fs' % (time.time()-t0), g.n_poses, g.n_odom, g.n_loops, flush=True)
P = capi.problem_from_graph(g, switchable=True, verbosity=1)
s0 = np.full(g.n_loops, 0.99)
t0=time.time(); P.solve_begin(g.init_q, g.init_t, s0); print('begin (H2D + graph build + iteration 0) %.2fs' % (time.time()-t0), flush=True)
for w,name in [(0,'K1'),(1,'K2'),(2,'PCG iteration'),(3,'K1 cost-only')]:
    ms,by = P.time_kernel(w, 20); print('%-14s %.1f us  %.0f GB/s (algorithmic)' % (name, ms*1e3, by/ms/1e6), flush=True)
t0=time.time()
for i in range(10): P.lm_step(ignore_termination=True)
dt=time.time()-t0
q,t,s,summ = P.solve_end()
print('10 LM iterations %.2fs -> %.2f it/s, cg total %d, chi2 %.6e -> %.6e' % (dt, 10/dt, summ.cg_iterations, 2*summ.initial_cost, 2*summ.final_cost))
