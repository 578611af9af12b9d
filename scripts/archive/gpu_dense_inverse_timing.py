"""HIP-event time of K6's dense inverse (blocked Gauss-Jordan on the fp64 MFMA) on coarse-operator sizes."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import capi

P = capi.Problem()
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "384,768,1216,1536,2048,2304,3072,4608".split(","))]:
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n // 2))
    A = B @ B.T + np.diag(rng.uniform(1e-3, 1.0, n))
    P.dense_spd_inverse(A)
    inv, ms = P.dense_spd_inverse(A, launches=5)
    err = np.abs(inv @ A - np.eye(n)).max()
    print('n %5d  %.3f ms  (%.1f GFLOP at n^3 -> %.2f TFLOP/s)  max |A^-1 A - I| %.1e' % (n, ms, n ** 3 / 1e9, n ** 3 / ms / 1e9, err), flush=True)
P.close()
