import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import capi
P = capi.Problem()
n = 3072
rng = np.random.default_rng(n)
B = rng.standard_normal((n, n // 2))
A = B @ B.T + np.diag(rng.uniform(1e-3, 1.0, n))
inv, ms = P.dense_spd_inverse(A, launches=2)
print('n', n, 'ms', ms)
P.close()
