"""Research probe (CPU, scipy): the session graph whose yaw-weighted odometry chain falls apart (15 degrees per keyframe in the turns).
PCG iteration counts on the damped reduced system at a late-stage linearisation for block-Jacobi, the uniform chain aggregates the
library uses, and aggregates whose boundaries sit on the weak links.  Not part of the product or the tests."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts/research')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
from precond_probe import build_system, block_diag_inv, prolongation, pcg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
turn = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=turn))
q, t, s = util.initial_state(g, True)
O = util.oracle_problem(g, True)
q, t, s, sm = O.solve(q, t, s)          # a late-stage point: what the triggers of a running session see
print('oracle: %d LM iterations, cost %.4g -> %.4g' % (sm.num_iterations, sm.initial_cost, sm.final_cost))
N = g.n_poses
# strength of the link between keyframes i and i+1: the odometry weights that span it
link = np.zeros(N - 1)
for c1, c2, w in zip(g.odom_c1, g.odom_c2, g.odom_w):
    lo, hi = min(c1, c2), max(c1, c2)
    link[lo:hi] += w * w
print('links: %d of %d below 1e-6 of the strongest' % ((link < 1e-6 * link.max()).sum(), N - 1))

def pieces(theta):
    cut = link < theta * link.max()
    return np.concatenate([[0], np.cumsum(cut)])

def merged(agg, target):
    """merge consecutive pieces until at most `target` aggregates are left (uniform over pieces)"""
    na = agg.max() + 1
    if na <= target: return agg
    return (agg * target) // na

for radius in (1e4, 1e6, 1e8):
    A, b = build_system(g, q, t, s, radius)
    Dinv = block_diag_inv(A, N)
    x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); print('radius %.0e: block-Jacobi %d' % (radius, k), flush=True)
    def two_level(agg, label):
        P, _ = prolongation(np.asarray(t).reshape(-1, 3), agg, True)
        Ac = (P.T @ A @ P).toarray()
        Aci = np.linalg.inv(Ac)
        x2, k2 = pcg(A, b, lambda r: Dinv @ r + P @ (Aci @ (P.T @ r)), 1e-9, maxit=60000)
        print('   %-46s aggregates %5d  its %6d' % (label, agg.max() + 1, k2), flush=True)
    for m in (3, 6):
        two_level(np.arange(N) // m, 'uniform chain aggregates of %d' % m)
    ag = pieces(1e-6)
    two_level(ag, 'pieces (cut at links < 1e-6 max)')
    two_level(merged(ag, 512), 'pieces merged to <= 512')
    two_level(merged(ag, 256), 'pieces merged to <= 256')
