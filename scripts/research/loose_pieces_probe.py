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

# ---- piece-aligned block-Jacobi: every strongly connected run of keyframes is one dense diagonal block ----
def piece_block_jacobi(A, agg):
    Ad = A.tocsr()
    blocks, idx = [], []
    for a in range(agg.max() + 1):
        ks = np.flatnonzero(agg == a)
        ii = (ks[:, None] * 6 + np.arange(6)[None, :]).ravel()
        blocks.append(np.linalg.inv(Ad[ii][:, ii].toarray())); idx.append(ii)
    def M(r):
        z = np.empty_like(r)
        for B, ii in zip(blocks, idx): z[ii] = B @ r[ii]
        return z
    return M

print('piece-aligned block-Jacobi')
for radius in (1e4, 1e6, 1e8):
    A, b = build_system(g, q, t, s, radius)
    for theta in (1e-6, 1e-2):
        ag = pieces(theta)
        x, k = pcg(A, b, piece_block_jacobi(A, ag), 1e-9, maxit=60000)
        print('   radius %.0e  cut at %.0e: %d blocks (largest %d keyframes)  its %d' % (radius, theta, ag.max() + 1, np.bincount(ag).max(), k), flush=True)

def capped(agg, cap):
    """split pieces longer than `cap` keyframes into consecutive runs of at most `cap`"""
    out = np.zeros_like(agg); a = -1; run = 0
    for i in range(len(agg)):
        if i == 0 or agg[i] != agg[i - 1] or run == cap: a += 1; run = 0
        out[i] = a; run += 1
    return out

print('capped piece blocks, alone and with the rigid modes of the (uncapped) pieces as coarse space')
for radius in (1e4, 1e6, 1e8):
    A, b = build_system(g, q, t, s, radius)
    ag = pieces(1e-6)
    P, _ = prolongation(np.asarray(t).reshape(-1, 3), ag, True)
    Aci = np.linalg.inv((P.T @ A @ P).toarray())
    for cap in (2, 4, 8, 1000):
        Mb = piece_block_jacobi(A, capped(ag, cap))
        x, k = pcg(A, b, Mb, 1e-9, maxit=60000)
        x, k2 = pcg(A, b, lambda r: Mb(r) + P @ (Aci @ (P.T @ r)), 1e-9, maxit=60000)
        print('   radius %.0e  cap %4d: %4d blocks  its %5d   + piece rigid modes %5d' % (radius, cap, capped(ag, cap).max() + 1, k, k2), flush=True)

print('6x6 block-Jacobi + rigid modes of runs of <= m keyframes INSIDE pieces (boundaries forced at weak links); singletons with / without coarse unknowns')
for radius in (1e4, 1e6, 1e8):
    A, b = build_system(g, q, t, s, radius)
    Dinv = block_diag_inv(A, N)
    ag0 = pieces(1e-6)
    for m in (4, 8, 16, 32):
        ag = capped(ag0, m)
        P, _ = prolongation(np.asarray(t).reshape(-1, 3), ag, True)
        sizes = np.bincount(ag)
        for drop_single in (False, True):
            Pm = P
            if drop_single:
                keep = np.repeat(sizes > 1, 6)
                Pm = P[:, np.flatnonzero(keep)]
            Aci = np.linalg.inv((Pm.T @ A @ Pm).toarray())
            x, k = pcg(A, b, lambda r: Dinv @ r + Pm @ (Aci @ (Pm.T @ r)), 1e-9, maxit=60000)
            print('   radius %.0e  m %2d  singletons %s: coarse unknowns %5d  its %5d' % (radius, m, 'dropped' if drop_single else 'kept   ', Pm.shape[1], k), flush=True)
