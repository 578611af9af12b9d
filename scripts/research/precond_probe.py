"""Research probe (CPU, scipy): PCG iteration counts on the Schur-reduced damped LM system of a C3-structured graph with
  (a) 6x6 block-Jacobi, (b) two-level: block-Jacobi smoother + chain-aggregate coarse space with rigid-body-mode prolongation,
  (c) multilevel version of (b).  Uses the oracle only to linearise.  Not part of the product or the tests."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

def build_system(g, q, t, s, radius):
    O = util.oracle_problem(g, True)
    N, S = g.n_poses, g.n_loops
    J1r, J2r, _ = O.jacobian_blocks(q, t, s, 0)
    J1s, J2s, dss = O.jacobian_blocks(q, t, s, 1)
    J1p, _, _ = O.jacobian_blocks(q, t, s, 2)
    cost, res, grad = O.evaluate(q, t, s)
    rows, cols, blks = [], [], []
    Hd = np.zeros((N, 6, 6))
    def add(c1, c2, J1, J2):
        np.add.at(Hd, c1, np.einsum('eia,eib->eab', J1, J1)); np.add.at(Hd, c2, np.einsum('eia,eib->eab', J2, J2))
        return np.einsum('eia,eib->eab', J1, J2)
    Hoff_r = add(g.odom_c1, g.odom_c2, J1r, J2r)
    Hoff_s = add(g.loop_c1, g.loop_c2, J1s, J2s)
    np.add.at(Hd, g.reg_node, np.einsum('eia,eib->eab', J1p, J1p))
    c1v = np.einsum('eia,ei->ea', J1s, dss[:, :6]); c2v = np.einsum('eia,ei->ea', J2s, dss[:, :6])
    hss = (dss ** 2).sum(1)
    diag = np.einsum('naa->na', Hd).copy()
    sc_p = 1 / (1 + np.sqrt(diag)); sc_s = 1 / (1 + np.sqrt(hss))
    lam_p = np.clip(sc_p ** 2 * diag, 1e-6, 1e32) / (radius * sc_p ** 2)
    lam_s = np.clip(sc_s ** 2 * hss, 1e-6, 1e32) / (radius * sc_s ** 2)
    a = hss + lam_s
    Hoff_s = Hoff_s - np.einsum('ea,eb->eab', c1v, c2v) / a[:, None, None]
    np.add.at(Hd, g.loop_c1, -np.einsum('ea,eb->eab', c1v, c1v) / a[:, None, None])
    np.add.at(Hd, g.loop_c2, -np.einsum('ea,eb->eab', c2v, c2v) / a[:, None, None])
    Hd[np.arange(N)[:, None], np.arange(6), np.arange(6)] += lam_p
    r_ = np.concatenate([np.arange(N), g.odom_c1, g.odom_c2, g.loop_c1, g.loop_c2])
    c_ = np.concatenate([np.arange(N), g.odom_c2, g.odom_c1, g.loop_c2, g.loop_c1])
    b_ = np.concatenate([Hd, Hoff_r, Hoff_r.transpose(0, 2, 1), Hoff_s, Hoff_s.transpose(0, 2, 1)])
    # COO of blocks -> BSR (duplicates summed)
    ii = (r_[:, None, None] * 6 + np.arange(6)[None, :, None]) + 0 * np.arange(6)[None, None, :]
    jj = (c_[:, None, None] * 6 + np.arange(6)[None, None, :]) + 0 * np.arange(6)[None, :, None]
    A = sp.coo_matrix((b_.ravel(), (ii.ravel(), jj.ravel())), shape=(6 * N, 6 * N)).tocsr()
    gs = np.einsum('ei,ei->e', dss, res[6 * g.n_odom:6 * g.n_odom + 7 * S].reshape(S, 7))
    b = -grad[:6 * N].copy()
    np.add.at(b.reshape(N, 6), g.loop_c1, c1v * (gs / a)[:, None]); np.add.at(b.reshape(N, 6), g.loop_c2, c2v * (gs / a)[:, None])
    return A, b

def block_diag_inv(A, N):
    D = np.zeros((N, 6, 6))
    Ab = A.tobsr(blocksize=(6, 6))
    for n in range(N):
        for k in range(Ab.indptr[n], Ab.indptr[n + 1]):
            if Ab.indices[k] == n: D[n] = Ab.data[k]
    Di = np.linalg.inv(D)
    return sp.block_diag(list(Di), format='csr') if N < 30000 else sp.bsr_matrix((Di, np.arange(N), np.arange(N + 1)), shape=(6 * N, 6 * N)).tocsr()

def prolongation(t, agg, rigid=True):
    """P: fine (dtheta_i, dt_i) <- coarse (dtheta_a, dt_a):  dtheta_i = dtheta_a ; dt_i = dt_a + 2 dtheta_a x (t_i - c_a)"""
    N = len(agg); na = agg.max() + 1
    cen = np.zeros((na, 3)); cnt = np.bincount(agg, minlength=na)
    np.add.at(cen, agg, t); cen /= cnt[:, None]
    blocks = np.zeros((N, 6, 6)); blocks[:, np.arange(6), np.arange(6)] = 1
    if rigid:
        d = t - cen[agg]
        # dt_i = 2 * (dtheta x d) = -2 [d]x dtheta
        X = np.zeros((N, 3, 3)); X[:, 0, 1] = -d[:, 2]; X[:, 0, 2] = d[:, 1]; X[:, 1, 0] = d[:, 2]; X[:, 1, 2] = -d[:, 0]; X[:, 2, 0] = -d[:, 1]; X[:, 2, 1] = d[:, 0]
        blocks[:, 3:, :3] = -2 * X
    P = sp.bsr_matrix((blocks, agg, np.arange(N + 1)), shape=(6 * N, 6 * na)).tocsr()
    return P, cen

def pcg(A, b, M, tol, maxit=20000):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; rz0 = rz; k = 0
    while k < maxit:
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q; z = M(r); rzn = r @ z; k += 1
        if rzn <= tol * tol * rz0: break
        p = z + (rzn / rz) * p; rz = rzn
    return x, k

class MG:
    def __init__(self, A, t, m, rigid=True, min_coarse=400, nu=1, omega=1.0):
        self.levels = []
        N = A.shape[0] // 6
        while True:
            Dinv = block_diag_inv(A, N)
            if N <= min_coarse:
                self.levels.append(dict(A=A, lu=spla.splu(A.tocsc())))
                break
            agg = np.arange(N) // m
            P, cen = prolongation(t, agg, rigid)
            Ac = (P.T @ A @ P).tocsr()
            self.levels.append(dict(A=A, Dinv=Dinv, P=P))
            A, t, N = Ac, cen, agg.max() + 1
        self.nu = nu; self.omega = omega
        print('   MG levels:', [(l['A'].shape[0] // 6, l['A'].nnz // 36) for l in self.levels])
    def vcycle(self, lvl, r):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(r)
        x = self.omega * (L['Dinv'] @ r)
        for _ in range(self.nu - 1): x += self.omega * (L['Dinv'] @ (r - L['A'] @ x))
        rc = L['P'].T @ (r - L['A'] @ x)
        x += L['P'] @ self.vcycle(lvl + 1, rc)
        for _ in range(self.nu): x += self.omega * (L['Dinv'] @ (r - L['A'] @ x))
        return x
    def __call__(self, r): return self.vcycle(0, r)

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    for radius in [1e4, 1e6, 1e9]:
        t0 = time.time(); A, b = build_system(g, q, t, s, radius); print('radius', radius, 'build', time.time() - t0)
        N = g.n_poses
        Dinv = block_diag_inv(A, N)
        x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8); print('  block-Jacobi            its', k)
        for m in (4, 8, 16):
            for rigid in (False, True):
                M = MG(A, t, m, rigid=rigid)
                x2, k2 = pcg(A, b, M, 1e-8); print('  MG m=%d rigid=%d nu=1       its' % (m, rigid), k2, 'err', np.abs(x2 - x).max() / np.abs(x).max())


class MGAdd(MG):
    """additive multilevel (no residual SpMVs): z = D0^-1 r + P (recursive)(P^T r)"""
    def vcycle(self, lvl, r):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(r)
        return L['Dinv'] @ r + L['P'] @ self.vcycle(lvl + 1, L['P'].T @ r)


class MGW(MG):
    gamma = 2
    def vcycle(self, lvl, r):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(r)
        x = self.omega * (L['Dinv'] @ r)
        for _ in range(self.gamma if lvl >= 1 else 1):
            rc = L['P'].T @ (r - L['A'] @ x)
            x += L['P'] @ self.vcycle(lvl + 1, rc)
        x += self.omega * (L['Dinv'] @ (r - L['A'] @ x))
        return x


def probe2(n, radii, m_list, min_coarse):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        N = g.n_poses
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8, maxit=30000); print('radius %g  block-Jacobi its %d (%.1fs)' % (radius, k, time.time() - t0), flush=True)
        for m in m_list:
            for cls in (MG, MGAdd, MGW):
                M = cls(A, t, m, rigid=True, min_coarse=min_coarse)
                t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-8, maxit=5000)
                print('  %s m=%d its %d (%.1fs) err %.1e' % (cls.__name__, m, k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)


class MGList(MG):
    """V-cycle with a per-level aggregate size list; last level solved exactly."""
    def __init__(self, A, t, ms, nu=1, omega=1.0, post_only=False):
        self.levels = []
        N = A.shape[0] // 6
        for m in ms:
            Dinv = block_diag_inv(A, N)
            agg = np.arange(N) // m
            P, cen = prolongation(t, agg, True)
            Ac = (P.T @ A @ P).tocsr()
            self.levels.append(dict(A=A, Dinv=Dinv, P=P))
            A, t, N = Ac, cen, agg.max() + 1
        self.levels.append(dict(A=A, lu=spla.splu(A.tocsc())))
        self.nu = nu; self.omega = omega
        print('   levels:', [(l['A'].shape[0] // 6, l['A'].nnz // 36) for l in self.levels], flush=True)


def probe3(n, radii, configs):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        xref = None
        for ms in configs:
            M = MGList(A, t, ms)
            t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-8, maxit=5000)
            if xref is None: xref = x2
            print('radius %g  ms=%s its %d (%.1fs) diff %.1e' % (radius, ms, k2, time.time() - t0, np.abs(x2 - xref).max() / np.abs(xref).max()), flush=True)


class SegTwoLevel:
    """smoother = exact solve of the diagonal super-blocks of `mseg` consecutive keyframes (non-overlapping additive Schwarz
    along the odometry chain); coarse space = rigid-body modes of aggregates of `magg` consecutive keyframes, solved exactly."""
    def __init__(self, A, t, mseg, magg, sym=True):
        N = A.shape[0] // 6
        seg = (np.arange(6 * N) // 6) // mseg
        Ac = A.tocoo()
        keep = seg[Ac.row] == seg[Ac.col]
        B = sp.coo_matrix((Ac.data[keep], (Ac.row[keep], Ac.col[keep])), shape=A.shape).tocsc()
        self.S = spla.splu(B)
        self.A = A
        if magg:
            self.P, _ = prolongation(t, np.arange(N) // magg, True)
            self.C = spla.splu((self.P.T @ A @ self.P).tocsc())
        else:
            self.P = None
        self.sym = sym
    def __call__(self, r):
        x = self.S.solve(r)
        if self.P is not None:
            x = x + self.P @ self.C.solve(self.P.T @ (r - self.A @ x))
            if self.sym:
                x = x + self.S.solve(r - self.A @ x)
        return x


def probe4(n, radii, configs):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        xref = None
        for (mseg, magg) in configs:
            M = SegTwoLevel(A, t, mseg, magg)
            t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-8, maxit=5000)
            if xref is None: xref = x2
            print('radius %g  seg=%d agg=%s its %d (%.1fs) diff %.1e' % (radius, mseg, magg, k2, time.time() - t0, np.abs(x2 - xref).max() / np.abs(xref).max()), flush=True)


class MGK(MGList):
    """K-cycle (Notay): the coarse problem of every level is solved by `kit` steps of flexible CG preconditioned by the next level."""
    kit = 2
    def vcycle(self, lvl, r):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(r)
        x = self.omega * (L['Dinv'] @ r)
        rc = L['P'].T @ (r - L['A'] @ x)
        x += L['P'] @ self.coarse_solve(lvl + 1, rc)
        x += self.omega * (L['Dinv'] @ (r - L['A'] @ x))
        return x
    def coarse_solve(self, lvl, b):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(b)
        A = L['A']
        # flexible CG, kit iterations, preconditioner = vcycle(lvl)
        x = np.zeros_like(b); r = b.copy(); ps = []; qs = []
        for it in range(self.kit):
            z = self.vcycle(lvl, r)
            p = z.copy()
            for (pp_, qq_) in zip(ps, qs): p -= (z @ qq_) / (pp_ @ qq_) * pp_
            q = A @ p
            al = (p @ r) / (p @ q)
            x += al * p; r -= al * q
            ps.append(p); qs.append(q)
        return x


def fpcg(A, b, M, tol, maxit=5000, trunc=1):
    """flexible PCG (needed when M is a K-cycle = nonlinear)"""
    x = np.zeros_like(b); r = b.copy(); k = 0; ps = []; qs = []; rz0 = None
    while k < maxit:
        z = M(r); rz = r @ z
        if rz0 is None: rz0 = rz
        if rz <= tol * tol * rz0: break
        p = z.copy()
        for (pp_, qq_) in zip(ps[-trunc:], qs[-trunc:]): p -= (z @ qq_) / (pp_ @ qq_) * pp_
        q = A @ p; al = (p @ r) / (p @ q); x += al * p; r -= al * q; ps.append(p); qs.append(q); k += 1
        ps = ps[-trunc:]; qs = qs[-trunc:]
    return x, k


def probe5(n, radii, configs):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        xref = None
        for ms in configs:
            M = MGK(A, t, ms)
            t0 = time.time(); x2, k2 = fpcg(A, b, M, 1e-8)
            if xref is None: xref = x2
            print('radius %g  K-cycle ms=%s its %d (%.1fs) diff %.1e' % (radius, ms, k2, time.time() - t0, np.abs(x2 - xref).max() / np.abs(xref).max()), flush=True)


class TwoLevel:
    """block-Jacobi + ONE coarse space of rigid-body modes of large aggregates, coarse problem solved exactly (on the GPU: a dense
    explicit inverse).  mode 'add': z = D^-1 r + P Ac^-1 P^T r ; 'mult': Jacobi, coarse correction, Jacobi (2 extra matvecs)."""
    def __init__(self, A, t, agg, Dinv, mode):
        self.A, self.Dinv, self.mode = A, Dinv, mode
        self.P, _ = prolongation(t, agg, True)
        Ac = (self.P.T @ A @ self.P).tocsc()
        self.C = spla.splu(Ac)
        self.nc = Ac.shape[0]
    def __call__(self, r):
        if self.mode == 'add':
            return self.Dinv @ r + self.P @ self.C.solve(self.P.T @ r)
        if self.mode == 'coarse_first':       # z = Pc r + D^-1 (r - A Pc r) ... unsymmetric, for flexible CG only
            x = self.P @ self.C.solve(self.P.T @ r)
            return x + self.Dinv @ (r - self.A @ x)
        x = self.Dinv @ r
        x = x + self.P @ self.C.solve(self.P.T @ (r - self.A @ x))
        return x + self.Dinv @ (r - self.A @ x)


def spatial_aggregates(t, n_agg, seed=0):
    """k-means-like clustering of keyframe positions (a few Lloyd iterations)"""
    from scipy.cluster.vq import kmeans2
    rng = np.random.default_rng(seed)
    cen = t[rng.choice(len(t), n_agg, replace=False)]
    _, lab = kmeans2(t, cen, iter=8, minit='matrix')
    _, lab = np.unique(lab, return_inverse=True)
    return lab


def probe6(n, radii, n_aggs, loops=None):
    g = graphgen.generate(n, loops if loops is not None else n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8, maxit=30000); print('radius %g  block-Jacobi its %d (%.1fs)' % (radius, k, time.time() - t0), flush=True)
        for na in n_aggs:
            for kind in ('chain', 'spatial'):
                agg = (np.arange(N) // int(np.ceil(N / na))) if kind == 'chain' else spatial_aggregates(t, na)
                for mode in ('add', 'mult'):
                    M = TwoLevel(A, t, agg, Dinv, mode)
                    t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-8, maxit=4000)
                    print('  n_agg %5d %-7s %-4s coarse dim %5d: its %4d (%.1fs) err %.1e' % (na, kind, mode, M.nc, k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)


class ChainTridiag:
    """M = block-tridiagonal part of A along the keyframe chain (diagonal blocks of A + the (i,i+1) blocks), solved exactly.
    SPD because every edge's contribution is PSD and dropping a PSD edge's off-diagonal blocks keeps its diagonal blocks."""
    def __init__(self, A, N, band=1):
        Ab = A.tobsr(blocksize=(6, 6))
        rows = np.repeat(np.arange(N), np.diff(Ab.indptr))
        keep = np.abs(rows - Ab.indices) <= band
        indptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=N))])
        M = sp.bsr_matrix((Ab.data[keep], Ab.indices[keep], indptr), shape=A.shape)
        self.lu = spla.splu(M.tocsc(), permc_spec='NATURAL', diag_pivot_thresh=0.0)
    def __call__(self, r): return self.lu.solve(r)


def probe7(n, loops, f, switchable, radii, bands=(1, 2)):
    g = graphgen.generate(n, loops, odom_f_max=f, seed=3)
    q, t, s = util.initial_state(g, switchable)
    N = g.n_poses
    for radius in radii:
        if switchable:
            A, b = build_system(g, q, t, s, radius)
        else:
            A, b = build_system_plain(g, q, t, radius)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8, maxit=60000); print('n %d loops %d f %d radius %g  block-Jacobi its %d (%.1fs)' % (n, loops, f, radius, k, time.time() - t0), flush=True)
        for band in bands:
            M = ChainTridiag(A, N, band)
            t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-8, maxit=20000)
            print('   chain band %d: its %d (%.1fs) err %.1e' % (band, k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)


def build_system_plain(g, q, t, radius):
    """all edges as plain relative-pose edges (C2 style)"""
    O = util.oracle_problem(g, False)
    N = g.n_poses
    s = np.zeros(0)
    J1r, J2r, _ = O.jacobian_blocks(q, t, s, 0)
    J1p, _, _ = O.jacobian_blocks(q, t, s, 2)
    cost, res, grad = O.evaluate(q, t, s)
    c1 = np.concatenate([g.odom_c1, g.loop_c1]); c2 = np.concatenate([g.odom_c2, g.loop_c2])
    Hd = np.zeros((N, 6, 6))
    np.add.at(Hd, c1, np.einsum('eia,eib->eab', J1r, J1r)); np.add.at(Hd, c2, np.einsum('eia,eib->eab', J2r, J2r))
    Hoff = np.einsum('eia,eib->eab', J1r, J2r)
    np.add.at(Hd, g.reg_node, np.einsum('eia,eib->eab', J1p, J1p))
    diag = np.einsum('naa->na', Hd).copy()
    sc = 1 / (1 + np.sqrt(diag))
    lam = np.clip(sc ** 2 * diag, 1e-6, 1e32) / (radius * sc ** 2)
    Hd[np.arange(N)[:, None], np.arange(6), np.arange(6)] += lam
    r_ = np.concatenate([np.arange(N), c1, c2]); c_ = np.concatenate([np.arange(N), c2, c1])
    b_ = np.concatenate([Hd, Hoff, Hoff.transpose(0, 2, 1)])
    ii = (r_[:, None, None] * 6 + np.arange(6)[None, :, None]) + 0 * np.arange(6)[None, None, :]
    jj = (c_[:, None, None] * 6 + np.arange(6)[None, None, :]) + 0 * np.arange(6)[None, :, None]
    A = sp.coo_matrix((b_.ravel(), (ii.ravel(), jj.ravel())), shape=(6 * N, 6 * N)).tocsr()
    return A, -grad[:6 * N].copy()


class ChainBlocks:
    """block-Jacobi with blocks of `m` consecutive keyframes (6m x 6m), solved exactly"""
    def __init__(self, A, N, m):
        seg = (np.arange(6 * N) // 6) // m
        Ac = A.tocoo()
        keep = seg[Ac.row] == seg[Ac.col]
        B = sp.coo_matrix((Ac.data[keep], (Ac.row[keep], Ac.col[keep])), shape=A.shape).tocsc()
        self.lu = spla.splu(B, permc_spec='NATURAL', diag_pivot_thresh=0.0)
    def __call__(self, r): return self.lu.solve(r)


def probe8(n, radii, ms=(2, 4, 8, 16)):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8, maxit=60000); print('n %d radius %g  block-Jacobi its %d' % (n, radius, k), flush=True)
        for m in ms:
            M = ChainBlocks(A, N, m)
            x2, k2 = pcg(A, b, M, 1e-8, maxit=20000)
            print('   %d-keyframe blocks: its %d' % (m, k2), flush=True)


class ClusterBlocks:
    """block-Jacobi whose blocks are arbitrary keyframe clusters (labels), each solved exactly"""
    def __init__(self, A, N, labels):
        lab = np.repeat(labels, 6)
        Ac = A.tocoo()
        keep = lab[Ac.row] == lab[Ac.col]
        B = sp.coo_matrix((Ac.data[keep], (Ac.row[keep], Ac.col[keep])), shape=A.shape).tocsc()
        self.lu = spla.splu(B)
    def __call__(self, r): return self.lu.solve(r)


def graph_clusters(g, m, seed=0):
    """greedy clusters of <= m keyframes grown along the heaviest connections (odometry + loop edges): BFS from unassigned seeds"""
    N = g.n_poses
    adj = [[] for _ in range(N)]
    for a, b in zip(np.concatenate([g.odom_c1, g.loop_c1]), np.concatenate([g.odom_c2, g.loop_c2])):
        adj[a].append(b); adj[b].append(a)
    lab = -np.ones(N, int); c = 0
    for s0 in range(N):
        if lab[s0] >= 0: continue
        grp = [s0]; lab[s0] = c; head = 0
        while head < len(grp) and len(grp) < m:
            u = grp[head]; head += 1
            for v in adj[u]:
                if lab[v] < 0 and len(grp) < m:
                    lab[v] = c; grp.append(v)
        c += 1
    return lab


def probe9(n, radii, ms=(4, 8, 16, 32)):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        x, k = pcg(A, b, lambda r: Dinv @ r, 1e-8, maxit=60000); print('n %d radius %g  block-Jacobi its %d' % (n, radius, k), flush=True)
        for m in ms:
            lab = graph_clusters(g, m)
            M = ClusterBlocks(A, N, lab)
            x2, k2 = pcg(A, b, M, 1e-8, maxit=20000)
            print('   graph clusters <= %d keyframes (%d clusters): its %d' % (m, lab.max() + 1, k2), flush=True)


def pcg_cg(A, b, M, tol, maxit=20000):
    """Chronopoulos-Gear PCG: ONE synchronisation point per iteration (both dot products right after the matvec)"""
    x = np.zeros_like(b); r = b.copy(); u = M(r); w = A @ u
    gam = r @ u; dlt = w @ u; gam0 = gam
    p = np.zeros_like(b); s = np.zeros_like(b); al = gam / dlt; be = 0.0; k = 0
    while k < maxit:
        p = u + be * p; s = w + be * s
        x += al * p; r -= al * s
        u = M(r); w = A @ u
        gam_new = r @ u; dlt = w @ u; k += 1
        if gam_new <= tol * tol * gam0: break
        be = gam_new / gam; al = gam_new / (dlt - be * gam_new / al); gam = gam_new
    return x, k, np.linalg.norm(b - A @ x) / np.linalg.norm(b)


def probe10(n, radii):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        M = lambda r: Dinv @ r
        for tol in (1e-9, 1e-12):
            x, k = pcg(A, b, M, tol, maxit=60000)
            x2, k2, res2 = pcg_cg(A, b, M, tol, maxit=60000)
            print('radius %g tol %g: PCG its %d true res %.1e | CG-CG its %d true res %.1e  dx %.1e' % (radius, tol, k, np.linalg.norm(b - A @ x) / np.linalg.norm(b), k2, res2,
                  np.abs(x2 - x).max() / np.abs(x).max()), flush=True)


def probe11(n, radii, ms=(4, 4, 4, 4, 4)):
    """V / W / K cycles at full size (the three heavy steps of the C3 trajectory run at radius 1e4, 3e4, 9e4)"""
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); print('n %d radius %g  block-Jacobi its %d (%.0fs)' % (n, radius, k, time.time() - t0), flush=True)
        M = MGList(A, t, list(ms))
        t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-9, maxit=5000); print('   V(1,1) ms=%s: its %d (%.0fs) err %.1e' % (list(ms), k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)
        M = MGK(A, t, list(ms))
        t0 = time.time(); x3, k3 = fpcg(A, b, M, 1e-9); print('   K-cycle: its %d (%.0fs) err %.1e' % (k3, time.time() - t0, np.abs(x3 - x).max() / np.abs(x).max()), flush=True)


def probe12(n, radii, n_aggs=(128, 512, 2048)):
    """two-level (block-Jacobi + one rigid-body coarse space solved exactly) at LARGE trust regions, where the slow modes are the long
    wavelengths: does a coarse space small enough for a dense inverse pay there?"""
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=200000); print('n %d radius %g  block-Jacobi its %d (%.0fs)' % (n, radius, k, time.time() - t0), flush=True)
        for na in n_aggs:
            agg = np.arange(N) // int(np.ceil(N / na))
            for mode in ('add', 'mult'):
                M = TwoLevel(A, t, agg, Dinv, mode)
                t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-9, maxit=50000)
                print('   chain aggregates %5d (coarse dim %5d) %-4s: its %5d (%.0fs) err %.1e' % (na, M.nc, mode, k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)


class MGAddList(MGList):
    """additive multilevel (BPX-like): z = D0^-1 r + P1 (D1^-1 + P2 (... exact)) P1^T r — no residual matvecs, symmetric"""
    def vcycle(self, lvl, r):
        L = self.levels[lvl]
        if 'lu' in L: return L['lu'].solve(r)
        return L['Dinv'] @ r + L['P'] @ self.vcycle(lvl + 1, L['P'].T @ r)


def probe13(n, radii, configs):
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        A, b = build_system(g, q, t, s, radius)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); print('n %d radius %g  block-Jacobi its %d (%.0fs)' % (n, radius, k, time.time() - t0), flush=True)
        for ms in configs:
            M = MGAddList(A, t, list(ms))
            t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-9, maxit=20000)
            print('   additive multilevel ms=%s: its %d (%.0fs) err %.1e' % (list(ms), k2, time.time() - t0, np.abs(x2 - x).max() / np.abs(x).max()), flush=True)
