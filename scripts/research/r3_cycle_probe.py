"""Research probe (CPU, scipy), round 3: which change to the multigrid cycle buys the most PCG iterations on a late C3-structured linearisation?
Variants: exact two-level (how much the recursion loses), K-cycle on level 1 only / on all levels, level-1 aggregate size, smoothed prolongation,
multiplicative fine level, over-correction.  Input: a system cached by r3_cache_system.py.  Not part of the product or the tests."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from scripts.research.precond_probe import block_diag_inv, prolongation, pcg
from scripts.research import amg_probe as ap

class G: pass

def load(path, radius):
    z = np.load(path)
    g = G(); g.odom_c1 = z['odom_c1']; g.odom_c2 = z['odom_c2']; g.odom_w = z['odom_w']; g.loop_c1 = z['loop_c1']; g.loop_c2 = z['loop_c2']; g.n_loops = len(g.loop_c1)
    t = z['t']; N = len(t)
    A = sp.csr_matrix((z['A_%g_data' % radius], z['A_%g_indices' % radius], z['A_%g_indptr' % radius]), shape=(6 * N, 6 * N))
    return g, t, A, z['b_%g' % radius], z['s']

def fpcg(A, b, M, tol, maxit=3000, trunc=1):
    """flexible PCG (Notay): direction orthogonalised against the last `trunc` directions"""
    x = np.zeros_like(b); r = b.copy(); k = 0; ps = []; qs = []; rz0 = None
    while k < maxit:
        z = M(r); rz = r @ z
        if rz0 is None: rz0 = rz
        if rz <= tol * tol * rz0: break
        p = z.copy()
        for (pp, qq, pq) in zip(ps, qs, [a @ c for a, c in zip(ps, qs)]): p -= (z @ qq) / pq * pp
        q = A @ p; al = (p @ r) / (p @ q); x += al * p; r -= al * q; k += 1
        ps.append(p); qs.append(q); ps = ps[-trunc:]; qs = qs[-trunc:]
    return x, k

class Hier:
    def __init__(self, A, t, agg_fns, min_coarse=512, smooth_levels=(), omega_p=0.66, verbose=True):
        self.levels = []
        N = A.shape[0] // 6
        lvl = 0
        while True:
            if N <= min_coarse:
                self.levels.append(dict(A=A, lu=spla.splu(A.tocsc()), N=N, nnzb=A.nnz // 36)); break
            Dinv = block_diag_inv(A, N)
            agg = agg_fns(A, N, lvl)
            P, cen = prolongation(t, agg, True)
            if lvl in smooth_levels:
                P = (P - omega_p * (Dinv @ (A @ P))).tocsr()
            Ac = (P.T @ A @ P).tocsr()
            self.levels.append(dict(A=A, Dinv=Dinv, P=P, N=N, nnzb=A.nnz // 36))
            A, t, N = Ac, cen, agg.max() + 1; lvl += 1
        self.nnz0 = self.levels[0]['nnzb']
        if verbose: print('   levels:', [(l['N'], l['nnzb']) for l in self.levels], flush=True)

class Cycle:
    """fine: 'add' (z = D^-1 r + P C(P^T r)) or 'mult' (V(1,1) on the fine level too)
    kcyc: set of levels whose system is solved by `kit` FCG steps preconditioned by the cycle from that level (K-cycle there); others: one cycle visit
    exact_from: level index from which the system is solved by LU (two-level exact = 1)"""
    def __init__(self, H, fine='add', omega=0.9, kcyc=(), kit=2, exact_from=None, wcyc=(), alpha=1.0, alpha0=None):
        self.H, self.fine, self.omega, self.kcyc, self.kit, self.wcyc, self.alpha = H, fine, omega, set(kcyc), kit, set(wcyc), alpha
        self.alpha0 = alpha if alpha0 is None else alpha0
        self.visits = [0] * len(H.levels); self.mvs = [0] * len(H.levels)
        self.lu = {}
        if exact_from is not None:
            for l in range(exact_from, len(H.levels)):
                if 'lu' not in H.levels[l]: self.lu[l] = spla.splu(H.levels[l]['A'].tocsc())
                break
    def mv(self, lvl, x): self.mvs[lvl] += 1; return self.H.levels[lvl]['A'] @ x
    def cyc(self, lvl, r):
        L = self.H.levels[lvl]; self.visits[lvl] += 1
        if 'lu' in L: return L['lu'].solve(r)
        if lvl in self.lu: return self.lu[lvl].solve(r)
        om = self.omega
        if lvl == 0 and self.fine == 'add':
            return L['Dinv'] @ r + self.alpha0 * (L['P'] @ self.coarse(1, L['P'].T @ r))
        x = om * (L['Dinv'] @ r)
        rc = L['P'].T @ (r - self.mv(lvl, x))
        x = x + self.alpha * (L['P'] @ self.coarse(lvl + 1, rc))
        x = x + om * (L['Dinv'] @ (r - self.mv(lvl, x)))
        return x
    def coarse(self, lvl, b):
        L = self.H.levels[lvl]
        if 'lu' in L or lvl in self.lu: return self.cyc(lvl, b)
        if lvl in self.wcyc:
            x = self.cyc(lvl, b); return x + self.cyc(lvl, b - self.mv(lvl, x))
        if lvl not in self.kcyc: return self.cyc(lvl, b)
        x = np.zeros_like(b); r = b.copy(); ps = []; qs = []
        for it in range(self.kit):
            z = self.cyc(lvl, r); p = z.copy()
            for (pp_, qq_) in zip(ps, qs): p -= (z @ qq_) / (pp_ @ qq_) * pp_
            q = self.mv(lvl, p); al = (p @ r) / (p @ q); x += al * p; r -= al * q; ps.append(p); qs.append(q)
        return x
    def __call__(self, r): return self.cyc(0, r)

def agg_product(g, p0, p, loop_w0=0.0):
    return lambda A_, N_, lvl: ap.topo_aggregates(g, N_, p0, loop_w=loop_w0) if lvl == 0 else ap.graph_aggregates(A_, N_, p)

if __name__ == '__main__':
    path = sys.argv[1]; radius = float(sys.argv[2]); which = sys.argv[3].split(',') if len(sys.argv) > 3 else ['base']
    g, t, A, b, s = load(path, radius)
    N = len(t)
    Dinv = block_diag_inv(A, N)
    import os
    cache = path + ".xbj_%g.npy" % radius
    if os.path.exists(cache): xbj = np.load(cache); kbj = -1
    else:
        xbj, kbj = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); np.save(cache, xbj)
    print("radius %g: block-Jacobi %d its" % (radius, kbj), flush=True)
    def report(name, H, M, flexible=False):
        t0 = time.time()
        x2, k2 = (fpcg if flexible else pcg)(A, b, M, 1e-9, maxit=3000)
        err = np.abs(x2 - xbj).max() / np.abs(xbj).max()
        print('   %-46s its %4d  visits/it %s  level-matvecs/it %s  err %.0e (%.0fs)' % (name, k2, ['%.1f' % (v / max(k2, 1)) for v in M.visits], ['%.1f' % (v / max(k2, 1)) for v in M.mvs], err, time.time() - t0), flush=True)
    for w in which:
        if w == 'base':
            for p0 in (3, 2):
                H = Hier(A, t, agg_product(g, p0, 3))
                report('p0=%d V add (product)' % p0, H, Cycle(H))
                report('p0=%d exact two-level add' % p0, H, Cycle(H, exact_from=1))
                report('p0=%d exact from level 2, add' % p0, H, Cycle(H, exact_from=2))
                report('p0=%d K on level 1 only (kit 2), add' % p0, H, Cycle(H, kcyc=(1,)), True)
                report('p0=%d K on level 1,2 (kit 2), add' % p0, H, Cycle(H, kcyc=(1, 2)), True)
                report('p0=%d K on level 2 only, add' % p0, H, Cycle(H, kcyc=(2,)), True)
                report('p0=%d W on level 2 only, add' % p0, H, Cycle(H, wcyc=(2,)))
                report('p0=%d V mult' % p0, H, Cycle(H, fine='mult'))
                report('p0=%d exact two-level mult' % p0, H, Cycle(H, fine='mult', exact_from=1))
        if w == 'smooth':
            for sl in ((0,), (0, 1), (1, 2)):
                H = Hier(A, t, agg_product(g, 3, 3), smooth_levels=sl)
                report('smoothed P on %s, V add' % (sl,), H, Cycle(H))
                report('smoothed P on %s, exact two-level add' % (sl,), H, Cycle(H, exact_from=1))
    for w in which:
        if w == 'kit':
            H = Hier(A, t, agg_product(g, 3, 3))
            for kit in (2, 3, 4, 6, 8):
                report('K on level 1, kit %d, add' % kit, H, Cycle(H, kcyc=(1,), kit=kit), True)
            H = Hier(A, t, agg_product(g, 3, 3), smooth_levels=(1, 2))
            for kit in (1, 2, 3, 4):
                report('smoothed(1,2) K on level 1, kit %d, add' % kit, H, Cycle(H, kcyc=(1,) if kit > 1 else (), kit=kit), True)
        if w == 'deep':
            for sl in ((), (1, 2, 3), (0, 1, 2, 3)):
                H = Hier(A, t, agg_product(g, 3, 3), min_coarse=64, smooth_levels=sl)
                report('deep smooth %s V add' % (sl,), H, Cycle(H))
                report('deep smooth %s exact from 2' % (sl,), H, Cycle(H, exact_from=2))
                report('deep smooth %s K on 1' % (sl,), H, Cycle(H, kcyc=(1,)), True)
                report('deep smooth %s K on 2,3' % (sl,), H, Cycle(H, kcyc=(2, 3)), True)
                report('deep smooth %s K on 1,2,3' % (sl,), H, Cycle(H, kcyc=(1, 2, 3)), True)
    for w in which:
        if w == 'shape':
            for (p0, p, mc) in ((3, 3, 100), (3, 5, 100), (3, 4, 200), (4, 3, 100), (4, 4, 100), (2, 4, 100), (3, 2, 100)):
                H = Hier(A, t, agg_product(g, p0, p), min_coarse=mc)
                report('p0=%d p=%d dense<=%d V add' % (p0, p, mc), H, Cycle(H))
    for w in which:
        if w == 'blocksmooth':
            # fine-level smoother = exact inverse of the diagonal block of every level-1 AGGREGATE (chains of <= 8 keyframes) instead of the 6x6 blocks
            agg0 = ap.topo_aggregates(g, N, 3, loop_w=0.0)
            order = np.argsort(agg0, kind='stable'); cnt = np.bincount(agg0)
            rows = []; cols = []; vals = []
            Acsr = A.tocsr(); start = 0
            for a in range(len(cnt)):
                mem = order[start:start + cnt[a]]; start += cnt[a]
                idx = (mem[:, None] * 6 + np.arange(6)[None, :]).ravel()
                Bi = np.linalg.inv(Acsr[idx][:, idx].toarray())
                rr, cc = np.meshgrid(idx, idx, indexing='ij'); rows.append(rr.ravel()); cols.append(cc.ravel()); vals.append(Bi.ravel())
            Binv = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=A.shape).tocsr()
            x2, k2 = pcg(A, b, lambda r: Binv @ r, 1e-9, maxit=60000); print('   aggregate-block Jacobi alone: %d its' % k2, flush=True)
            H = Hier(A, t, lambda A_, N_, lvl: agg0 if lvl == 0 else ap.graph_aggregates(A_, N_, 3))
            H.levels[0]['Dinv'] = Binv
            report('aggregate-block smoother, V add', H, Cycle(H))
            report('aggregate-block smoother, exact two-level add', H, Cycle(H, exact_from=1))
            report('aggregate-block smoother, K level 1 kit 2', H, Cycle(H, kcyc=(1,)), True)

    for w in which:
        if w == 'alpha':
            H = Hier(A, t, agg_product(g, 3, 3))
            for (a0, a) in ((1.0, 1.0), (1.0, 1.3), (1.0, 1.6), (1.0, 2.0), (1.5, 1.0), (1.5, 1.5), (2.0, 1.0), (0.7, 1.0)):
                report('alpha0 %.1f alpha %.1f V add' % (a0, a), H, Cycle(H, alpha=a, alpha0=a0))
            for om in (0.7, 1.0):
                report('omega %.1f V add' % om, H, Cycle(H, omega=om))
    for w in which:
        if w == 'which':
            for sl in ((), (1,), (2,), (1, 2)):
                H = Hier(A, t, agg_product(g, 3, 3), min_coarse=100, smooth_levels=sl)
                report('4 levels, smoothed transitions %s, V add' % (sl,), H, Cycle(H))
    for w in which:
        if w == 'omegap':
            for op in (0.5, 0.66, 0.9, 1.0):
                H = Hier(A, t, agg_product(g, 3, 3), min_coarse=100, smooth_levels=(1,), omega_p=op, verbose=False)
                report('4 levels, transition 1 smoothed, omega_p %.2f, V add' % op, H, Cycle(H))
