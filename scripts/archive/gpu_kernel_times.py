import sys; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
g = graphgen.config(sys.argv[1] if len(sys.argv)>1 else "C3")
P = capi.problem_from_graph(g, switchable=True, linear_solver=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
P.solve_begin(g.init_q, g.init_t, np.full(g.n_loops, 0.99))
for w,name in [(0,'K1'),(1,'K2'),(2,'PCG iteration'),(3,'K1 cost-only')]:
    ms,by = P.time_kernel(w, 50); print('%-14s %.2f us  %.0f GB/s (algorithmic)' % (name, ms*1e3, by/ms/1e6))
P.solve_end()
