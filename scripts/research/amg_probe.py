"""Research probe (CPU, scipy): aggregation multigrid with GRAPH-BASED aggregates (heavy-edge pairwise matching over odometry AND
loop-closure couplings) and rigid-body-mode prolongation, as preconditioner of the PCG on the Schur-reduced damped LM system.
Reports per configuration: levels (keyframes, blocks), operator complexity, PCG iterations, and a cost in fine-matvec units.
Not part of the product or the tests."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from scripts.research.precond_probe import build_system, block_diag_inv, prolongation, pcg, fpcg
from solve_keyframe_pose_graph_amd import graphgen
from tests import util


def block_graph(A, N):
    """(rows, cols, strength) of the off-diagonal 6x6 blocks: strength = ||A_ij||_F / sqrt(||A_ii||_F ||A_jj||_F)"""
    Ab = A.tobsr(blocksize=(6, 6)); Ab.sort_indices()
    rows = np.repeat(np.arange(N), np.diff(Ab.indptr))
    fro = np.sqrt((Ab.data ** 2).sum((1, 2)))
    dmask = rows == Ab.indices
    d = np.zeros(N); d[rows[dmask]] = fro[dmask]
    off = ~dmask
    return rows[off], Ab.indices[off], fro[off] / np.sqrt(d[rows[off]] * d[Ab.indices[off]])


def pairwise_match(n, r, c, w, sizes=None, max_size=None):
    """greedy heavy-edge matching: edges by decreasing weight; returns aggregate id per node"""
    order = np.argsort(-w, kind='stable')
    mate = -np.ones(n, dtype=np.int64)
    for k in order:
        i, j = r[k], c[k]
        if i == j or mate[i] >= 0 or mate[j] >= 0: continue
        if max_size is not None and sizes[i] + sizes[j] > max_size: continue
        mate[i] = j; mate[j] = i
    agg = -np.ones(n, dtype=np.int64); na = 0
    for i in range(n):
        if agg[i] >= 0: continue
        agg[i] = na
        if mate[i] >= 0: agg[mate[i]] = na
        na += 1
    return agg


def graph_aggregates(A, N, passes, max_size=None):
    """`passes` rounds of pairwise matching on the (re-aggregated) strength graph -> aggregates of up to 2^passes keyframes"""
    r, c, w = block_graph(A, N)
    agg = np.arange(N); sizes = np.ones(N, dtype=np.int64)
    n = N
    for _ in range(passes):
        a2 = pairwise_match(n, r, c, w, sizes, max_size)
        # collapse graph
        agg = a2[agg]
        n2 = a2.max() + 1
        sizes = np.bincount(a2, weights=sizes, minlength=n2).astype(np.int64)
        rr, cc = a2[r], a2[c]
        keep = rr != cc
        M = sp.coo_matrix((w[keep], (rr[keep], cc[keep])), shape=(n2, n2)).tocsr().tocoo()   # sums duplicates
        r, c, w = M.row, M.col, M.data
        n = n2
    return agg


class Hier:
    def __init__(self, A, t, agg_fn, min_coarse=300, max_levels=12, verbose=True):
        self.levels = []
        N = A.shape[0] // 6
        while True:
            if N <= min_coarse or len(self.levels) >= max_levels:
                self.levels.append(dict(A=A, lu=spla.splu(A.tocsc()), N=N, nnzb=A.nnz // 36))
                break
            Dinv = block_diag_inv(A, N)
            agg = agg_fn(A, N, len(self.levels))
            P, cen = prolongation(t, agg, True)
            Ac = (P.T @ A @ P).tocsr()
            self.levels.append(dict(A=A, Dinv=Dinv, P=P, N=N, nnzb=A.nnz // 36))
            A, t, N = Ac, cen, agg.max() + 1
        self.nnz0 = self.levels[0]['nnzb']
        if verbose:
            print('   levels:', [(l['N'], l['nnzb']) for l in self.levels], 'op-complexity %.2f' % (sum(l['nnzb'] for l in self.levels) / self.nnz0), flush=True)

    def lam_max(self, lvl, its=20):
        L = self.levels[lvl]
        v = np.random.default_rng(0).normal(size=L['A'].shape[0])
        for _ in range(its):
            v = L['Dinv'] @ (L['A'] @ v); lam = np.linalg.norm(v); v /= lam
        return lam


class Cycle:
    """cycle: 'V', 'W', 'K' ; smoother: ('jac', omega) or ('cheb', degree)"""
    def __init__(self, H, cycle='V', smoother=('jac', 0.8), nu=1, kit=2, fine_additive=False, alpha=1.0, alpha0=1.0):
        self.alpha, self.alpha0 = alpha, alpha0
        self.H, self.cycle, self.smoother, self.nu, self.kit = H, cycle, smoother, nu, kit
        self.fine_additive = fine_additive
        self.work = 0.0    # in fine-matvec units (blocks touched / fine blocks)
        self.syncs = 0
        if smoother[0] == 'cheb':
            self.lmax = [H.lam_max(l) * 1.1 for l in range(len(H.levels) - 1)]
    def mv(self, lvl, x):
        self.work += self.H.levels[lvl]['nnzb'] / self.H.nnz0; self.syncs += 1
        return self.H.levels[lvl]['A'] @ x
    def smooth(self, lvl, x, r, first):
        L = self.H.levels[lvl]
        if self.smoother[0] == 'jac':
            om = self.smoother[1]
            for k in range(self.nu):
                if first and k == 0 and x is None: x = om * (L['Dinv'] @ r)
                else: x = x + om * (L['Dinv'] @ (r - self.mv(lvl, x)))
            return x
        # Chebyshev of degree d on D^-1 A, eigenvalue interval [lmax/ratio, lmax]
        d = self.smoother[1]; ratio = self.smoother[2] if len(self.smoother) > 2 else 4.0
        lmax = self.lmax[lvl]; lmin = lmax / ratio
        theta = 0.5 * (lmax + lmin); delta = 0.5 * (lmax - lmin)
        if x is None: x = np.zeros_like(r); res = r.copy()
        else: res = r - self.mv(lvl, x)
        sigma = theta / delta; rho = 1.0 / sigma
        dk = (L['Dinv'] @ res) / theta
        x = x + dk
        for k in range(1, d):
            res = res - self.mv(lvl, dk)
            rho_n = 1.0 / (2 * sigma - rho)
            dk = rho_n * rho * dk + (2 * rho_n / delta) * (L['Dinv'] @ res)
            rho = rho_n
            x = x + dk
        return x
    def cyc(self, lvl, r):
        L = self.H.levels[lvl]
        if 'lu' in L: self.syncs += 1; return L['lu'].solve(r)
        if lvl == 0 and self.fine_additive:
            return L['Dinv'] @ r + self.alpha0 * (L['P'] @ self.coarse(lvl + 1, L['P'].T @ r))
        x = self.smooth(lvl, None, r, True)
        rc = L['P'].T @ (r - self.mv(lvl, x)); self.syncs += 1
        x = x + self.alpha * (L['P'] @ self.coarse(lvl + 1, rc)); self.syncs += 1
        x = self.smooth(lvl, x, r, False)
        return x
    def coarse(self, lvl, b):
        L = self.H.levels[lvl]
        if 'lu' in L or self.cycle == 'V': return self.cyc(lvl, b)
        if self.cycle == 'W':
            x = self.cyc(lvl, b)
            x = x + self.cyc(lvl, b - self.mv(lvl, x))
            return x
        # K-cycle: kit steps of flexible CG preconditioned by the cycle at this level
        x = np.zeros_like(b); r = b.copy(); ps = []; qs = []
        for it in range(self.kit):
            z = self.cyc(lvl, r); p = z.copy()
            for (pp_, qq_) in zip(ps, qs): p -= (z @ qq_) / (pp_ @ qq_) * pp_
            q = self.mv(lvl, p); al = (p @ r) / (p @ q); x += al * p; r -= al * q; ps.append(p); qs.append(q); self.syncs += 2
        return x
    def __call__(self, r): return self.cyc(0, r)


def run(n, radii, passes_list=(2, 3), loops=None, seed=3):
    g = graphgen.generate(n, loops if loops is not None else n, odom_f_max=2, seed=seed)
    q, t, s = util.initial_state(g, True)
    N = g.n_poses
    for radius in radii:
        t0 = time.time(); A, b = build_system(g, q, t, s, radius); print('radius %g build %.1fs' % (radius, time.time() - t0), flush=True)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=40000); print('  block-Jacobi its %d (%.1fs)' % (k, time.time() - t0), flush=True)
        for passes in passes_list:
            for kind in ('graph', 'chain'):
                if kind == 'graph': fn = lambda A_, N_, lvl, p_=passes: graph_aggregates(A_, N_, p_)
                else: fn = lambda A_, N_, lvl, p_=passes: np.arange(N_) // (2 ** p_)
                t0 = time.time(); H = Hier(A, t, fn); print('   (%s, 2^%d) setup %.1fs' % (kind, passes, time.time() - t0), flush=True)
                for (cyc, sm, fa) in (('V', ('jac', 0.7), False), ('V', ('jac', 1.0), False), ('V', ('cheb', 2), False), ('V', ('cheb', 3), False), ('W', ('jac', 0.7), False), ('K', ('jac', 0.7), False),
                                      ('V', ('jac', 0.7), True), ('K', ('jac', 0.7), True)):
                    M = Cycle(H, cyc, sm, fine_additive=fa)
                    t0 = time.time()
                    try:
                        x2, k2 = (fpcg if cyc == 'K' else pcg)(A, b, M, 1e-9, maxit=1500)
                    except Exception as ex:
                        print('      %s %s fail %s' % (cyc, sm, ex)); continue
                    err = np.abs(x2 - x).max() / np.abs(x).max()
                    w = M.work / max(k2, 1)
                    print('      %s %-14s fineadd=%d its %4d  work/it %.2f (+1 outer)  syncs/it %3d  total work %.0f vs BJ %d  err %.1e (%.1fs)' %
                          (cyc, sm, fa, k2, w, M.syncs / max(k2, 1), (w + 1) * k2, k, err, time.time() - t0), flush=True)




def topo_aggregates(g, N, passes, loop_w=1.0):
    """aggregation from the graph topology and nominal edge weights only (what the host knows at graph build): strength = sum of w^2 of
    parallel edges; loops weigh loop_w"""
    r = np.concatenate([g.odom_c1, g.loop_c1]); c = np.concatenate([g.odom_c2, g.loop_c2])
    w = np.concatenate([g.odom_w ** 2, np.full(g.n_loops, loop_w)])
    M = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([r, c]), np.concatenate([c, r]))), shape=(N, N)).tocsr().tocoo()
    r, c, w = M.row, M.col, M.data
    agg = np.arange(N); n = N
    for _ in range(passes):
        a2 = pairwise_match(n, r, c, w)
        agg = a2[agg]; n2 = a2.max() + 1
        rr, cc = a2[r], a2[c]; keep = rr != cc
        M = sp.coo_matrix((w[keep], (rr[keep], cc[keep])), shape=(n2, n2)).tocsr().tocoo()
        r, c, w = M.row, M.col, M.data; n = n2
    return agg


def run2(n, radii, schedules, loops=None, seed=3, state=None, topo=False):
    g = graphgen.generate(n, loops if loops is not None else n, odom_f_max=2, seed=seed)
    q, t, s = util.initial_state(g, True)
    if state == 'late':   # a late-stage linearisation: 8 LM iterations of the oracle
        O = util.oracle_problem(g, True)
        import ctypes
        q, t, s, summ = O.solve(q, t, s, max_num_iterations=8)
        print('late state: cost', summ.final_cost, 'switches < 0.5:', int((s < 0.5).sum()), flush=True)
    N = g.n_poses
    for radius in radii:
        t0 = time.time(); A, b = build_system(g, q, t, s, radius); print('radius %g build %.1fs' % (radius, time.time() - t0), flush=True)
        Dinv = block_diag_inv(A, N)
        t0 = time.time(); x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); print('  block-Jacobi its %d (%.1fs)' % (k, time.time() - t0), flush=True)
        for sched in schedules:
            def fn(A_, N_, lvl, sched=sched):
                p_ = sched[min(lvl, len(sched) - 1)]
                if lvl == 0 and topo: return topo_aggregates(g, N_, p_)
                return graph_aggregates(A_, N_, p_)
            t0 = time.time(); H = Hier(A, t, fn, min_coarse=500); print('   sched %s topo=%d setup %.1fs' % (sched, topo, time.time() - t0), flush=True)
            for (cyc, sm, nu) in (('V', ('jac', 0.7), 1), ('V', ('jac', 1.0), 1), ('V', ('jac', 0.7), 2), ('W', ('jac', 0.7), 1)):
                M = Cycle(H, cyc, sm, nu=nu, fine_additive=True)
                t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-9, maxit=3000)
                err = np.abs(x2 - x).max() / np.abs(x).max(); w = M.work / max(k2, 1)
                print('      %s %-12s nu=%d its %4d  work/it %.2f  syncs/it %3d  BJ/its %.1f  err %.1e (%.1fs)' % (cyc, sm, nu, k2, w, M.syncs / max(k2, 1), k / k2, err, time.time() - t0), flush=True)

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    radii = [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1e4, 1e6]
    mode = sys.argv[3] if len(sys.argv) > 3 else 'run'
    if mode == 'run': run(n, radii)
    else:
        scheds = [(2,), (3,), (2, 3), (1, 2), (3, 2)]
        run2(n, radii, scheds, state='late' if 'late' in mode else None, topo='topo' in mode)
