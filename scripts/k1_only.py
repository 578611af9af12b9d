"""Runs only the K1 kernel (plus setup) on C3 a fixed number of times — the target of the rocprofv3 --pmc passes."""
import sys
sys.path.insert(0, ".")
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
import sys
n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = graphgen.generate(n_big, n_big, odom_f_max=2, seed=3) if n_big else graphgen.config("C3")
P = capi.problem_from_graph(g, switchable=True)
P.solve_begin(g.init_q, g.init_t, np.full(g.n_loops, 0.99))
ms, by = P.time_kernel(0, 20)
print("k1 ms", ms, "GB/s", by / ms / 1e6)
P.solve_end()
