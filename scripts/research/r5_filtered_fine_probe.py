"""CPU probe (round 5, after the device measurement of mg_smoothed_fine): a smoothed keyframe -> level-1 prolongator formed with a FILTERED matrix — loop closures dropped from
(I - w D_f^-1 A_f) P_0 so that level 1 does not take every loop closure of a neighbouring keyframe along — with the dropped blocks LUMPED into the diagonal so that A_f keeps A's
action on the rigid-body modes (A_f B = A B, B_i = the rigid motion seen at keyframe i): A_ii^f = A_ii + sum_dropped A_ij B_j B_i^-1.  Prints PCG iterations and level sizes.
  python scripts/research/r5_filtered_fine_probe.py <system dump of scripts/research/r3_cycle_probe.py>"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from scripts.research.precond_probe import block_diag_inv, pcg, prolongation
from scripts.research.r3_cycle_probe import load, agg_product
from scripts.research.r5_smoother_probe import CycleNu
path = sys.argv[1]
g, t, A, b, s = load(path, 1e6)
N = len(t)


def block_mask(N, pairs):
    r = np.concatenate([np.arange(N)] + [p[0] for p in pairs] + [p[1] for p in pairs]); c = np.concatenate([np.arange(N)] + [p[1] for p in pairs] + [p[0] for p in pairs])
    Mb = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(N, N)); Mb.data[:] = 1
    return sp.kron(Mb, np.ones((6, 6))).tocsr()


def lumped_filter(A, t, keep_mask):
    """A_f = kept blocks of A, diagonal blocks corrected so that A_f B = A B for the six global rigid modes B"""
    N = len(t)
    B, _ = prolongation(t, np.zeros(N, int), True)            # (6N x 6): B_i
    kept = A.multiply(keep_mask).tocsr()
    dropped = (A - kept).tocsr()
    DB = (dropped @ B).toarray().reshape(N, 6, 6) if sp.issparse(dropped @ B) else np.asarray(dropped @ B).reshape(N, 6, 6)
    Bi = B.toarray().reshape(N, 6, 6)
    corr = np.einsum('nij,njk->nik', DB, np.linalg.inv(Bi))
    corr = 0.5 * (corr + corr.transpose(0, 2, 1))                 # (keep the diagonal blocks symmetric)
    C = sp.bsr_matrix((corr, np.arange(N), np.arange(N + 1)), shape=(6 * N, 6 * N)).tocsr()
    return (kept + C).tocsr()


class HierF:
    def __init__(self, A, t, agg_fns, filt0=None, lump=True, min_coarse=512, smooth_levels=(0, 1), omega_p=0.6):
        self.levels = []
        N = A.shape[0] // 6; lvl = 0
        while True:
            if N <= min_coarse:
                self.levels.append(dict(A=A, lu=spla.splu(A.tocsc()), N=N, nnzb=A.nnz // 36)); break
            Dinv = block_diag_inv(A, N)
            agg = agg_fns(A, N, lvl)
            P, cen = prolongation(t, agg, True)
            if lvl in smooth_levels:
                if lvl == 0 and filt0 is not None:
                    Af = lumped_filter(A, t, filt0) if lump else A.multiply(filt0).tocsr()
                    P = (P - omega_p * (block_diag_inv(Af, N) @ (Af @ P))).tocsr()
                else:
                    P = (P - omega_p * (Dinv @ (A @ P))).tocsr()
                P.eliminate_zeros()
            Ac = (P.T @ A @ P).tocsr()
            if lvl >= 1:      # the product's smoother safety (mg_limit_smoother): w lambda_max(D^-1 A) above 1.75 is scaled down to 1.5 — without it a level whose lambda_max exceeds 2 / w makes the cycle indefinite
                v = 1.0 + 0.5 * np.sin(0.7 * np.arange(A.shape[0]))
                for _ in range(8): w = v; v = Dinv @ (A @ v)
                lam = np.linalg.norm(v) / np.linalg.norm(w)
                if 0.9 * lam > 1.75: Dinv = Dinv * (1.5 / (0.9 * lam)); print('   level %d: w lambda_max %.2f -> smoother scaled to 1.5' % (lvl, 0.9 * lam))
            self.levels.append(dict(A=A, Dinv=Dinv, P=P, N=N, nnzb=A.nnz // 36))
            A, t, N = Ac, cen, agg.max() + 1; lvl += 1
        print('   levels:', [(int(l['N']), int(l['nnzb'])) for l in self.levels], ' Ps_0 blocks per keyframe %.2f' % (self.levels[0]['P'].nnz / 36 / self.levels[0]['N']), flush=True)


odo = block_mask(N, [(g.odom_c1, g.odom_c2)])
for name, kw in (("default shape: smoothed (1,)", dict(smooth_levels=(1,))), ("smoothed (0, 1), full A", dict()), ("smoothed (0, 1), level 0 along odometry, NOT lumped", dict(filt0=odo, lump=False)),
                 ("smoothed (0, 1), level 0 along odometry, lumped", dict(filt0=odo, lump=True))):
    H = HierF(A, t, agg_product(g, 3, 2), **kw)
    x2, k2 = pcg(A, b, CycleNu(H), 1e-9, maxit=3000)
    print('%-58s its %4d' % (name, k2), flush=True)

# (a PCG whose r.z turns negative "converges" at once: the filtered variant is checked against the solution of the unfiltered one, and the lumped diagonal blocks for definiteness)
H = HierF(A, t, agg_product(g, 3, 2))
xref, kref = pcg(A, b, CycleNu(H), 1e-10, maxit=3000)
H = HierF(A, t, agg_product(g, 3, 2), filt0=odo, lump=True)
x2, k2 = pcg(A, b, CycleNu(H), 1e-9, maxit=3000)
print('lumped filter: its %d, error against the unfiltered hierarchy\'s solution (%d its to 1e-10) %.1e' % (k2, kref, np.abs(x2 - xref).max() / np.abs(xref).max()), flush=True)
Bf = sp.bsr_matrix(lumped_filter(A, t, odo), blocksize=(6, 6)); rows = np.repeat(np.arange(N), np.diff(Bf.indptr))
print("smallest eigenvalue over the lumped diagonal blocks: %.3e" % np.linalg.eigvalsh(Bf.data[rows == Bf.indices]).min())
