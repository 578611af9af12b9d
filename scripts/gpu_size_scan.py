import sys; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
for n in [12500, 25000, 50000, 75000, 100000, 150000, 200000, 400000]:
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    P = capi.problem_from_graph(g, switchable=True)
    P.solve_begin(g.init_q, g.init_t, np.full(g.n_loops, 0.99))
    r=[]
    for w in (0,1,2,3):
        ms,by = P.time_kernel(w, 50); r.append((ms*1e3, by/ms/1e6))
    nnzb = n + 2*(g.n_odom+g.n_loops)
    print('N %7d  K1 %6.1f us %5.0f GB/s | K2 %6.1f us | PCG it %6.1f us  (BSR %.0f MB -> %.0f GB/s on BSR bytes) | K1c %5.1f us' % (n, r[0][0], r[0][1], r[1][0], r[2][0], nnzb*288/1e6, nnzb*288/r[2][0]/1e3, r[3][0]))
    P.solve_end(); P.close()
