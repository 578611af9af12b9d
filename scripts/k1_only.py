"""Runs only the K1 kernel (plus setup) on C3 a fixed number of times — the target of the rocprofv3 --pmc passes."""
import sys
sys.path.insert(0, ".")
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
g = graphgen.config("C3")
P = capi.problem_from_graph(g, switchable=True)
P.solve_begin(g.init_q, g.init_t, np.full(g.n_loops, 0.99))
ms, by = P.time_kernel(0, 20)
print("k1 ms", ms, "GB/s", by / ms / 1e6)
P.solve_end()
