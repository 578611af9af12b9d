"""Research probe (CPU, scipy), round 5: what would MORE smoothing on the coarse levels buy, now that a level with a smoothed transition above costs two launches per cycle whatever
its smoother is (explicit operators v = S r, r_next = R r, x = v + R^T x_next can carry any polynomial S of the level matrix at the price of denser blocks, not of launches)?
Variants on the product-like hierarchy (level 1 = aggregates of 8 keyframes, smoothed transition 1 -> 2, additive fine level): nu = 1 / 2 / 3 damped-Jacobi steps on level 1 only
and on every coarse level; a degree-2 / degree-3 Chebyshev polynomial on level 1.  Input: a system cached by r3_cache_system.py.  Not part of the product or the tests.
  python scripts/research/r5_smoother_probe.py /tmp/r3_sys.npz 1e6"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse.linalg as spla
from scripts.research.precond_probe import block_diag_inv, pcg
from scripts.research.r3_cycle_probe import load, Hier, Cycle, agg_product


class CycleNu(Cycle):
    """nu[lvl] damped block-Jacobi steps before and after the coarse correction on level lvl (default 1); cheb[lvl] = degree of a Chebyshev polynomial in D^-1 A instead"""
    def __init__(self, H, nu=None, cheb=None, **kw):
        super().__init__(H, **kw)
        self.nu = nu or {}; self.cheb = cheb or {}; self.lmax = {}
    def smooth(self, lvl, x, r):
        L = self.H.levels[lvl]
        if lvl in self.cheb:
            if lvl not in self.lmax:      # power method on D^-1 A
                v = np.random.default_rng(0).normal(size=L['A'].shape[0])
                for _ in range(12): v = L['Dinv'] @ (L['A'] @ v); v /= np.linalg.norm(v)
                self.lmax[lvl] = 1.1 * float(v @ (L['Dinv'] @ (L['A'] @ v)))
            lmax = self.lmax[lvl]; lmin = lmax / 8.0
            theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
            sigma = theta / delta; rho = 1.0 / sigma
            res = r - self.mv(lvl, x) if np.any(x) else r.copy()
            d = (L['Dinv'] @ res) / theta
            x = x + d
            for _ in range(self.cheb[lvl] - 1):
                res = res - self.mv(lvl, d)
                rho_new = 1.0 / (2.0 * sigma - rho)
                d = rho_new * rho * d + (2.0 * rho_new / delta) * (L['Dinv'] @ res)
                x = x + d; rho = rho_new
            return x
        for _ in range(self.nu.get(lvl, 1)):
            x = x + self.omega * (L['Dinv'] @ (r - self.mv(lvl, x) if np.any(x) else r))
        return x
    def cyc(self, lvl, r):
        L = self.H.levels[lvl]; self.visits[lvl] += 1
        if 'lu' in L: return L['lu'].solve(r)
        if lvl in self.lu: return self.lu[lvl].solve(r)
        if lvl == 0 and self.fine == 'add':
            return L['Dinv'] @ r + self.alpha0 * (L['P'] @ self.coarse(1, L['P'].T @ r))
        x = self.smooth(lvl, np.zeros_like(r), r)
        rc = L['P'].T @ (r - self.mv(lvl, x))
        x = x + self.alpha * (L['P'] @ self.coarse(lvl + 1, rc))
        return self.smooth(lvl, x, r)


if __name__ == '__main__':
    path = sys.argv[1]; radius = float(sys.argv[2])
    g, t, A, b, s = load(path, radius)
    N = len(t)
    Dinv = block_diag_inv(A, N)
    xref, kbj = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000)
    print("radius %g: block-Jacobi %d its" % (radius, kbj), flush=True)
    H = Hier(A, t, agg_product(g, 3, 2), smooth_levels=(1,), omega_p=0.6)      # aggregates of 8 keyframes, then of 4, smoothed transition level 1 -> 2 (the product's shape)

    def report(name, M):
        t0 = time.time()
        x2, k2 = pcg(A, b, M, 1e-9, maxit=3000)
        err = np.abs(x2 - xref).max() / np.abs(xref).max()
        print('   %-58s its %4d  level-matvecs/it %s  err %.0e (%.0fs)' % (name, k2, ['%.1f' % (v / max(k2, 1)) for v in M.mvs], err, time.time() - t0), flush=True)
    nl = len(H.levels)
    report('product-like: nu = 1 everywhere', CycleNu(H))
    report('nu = 2 on level 1', CycleNu(H, nu={1: 2}))
    report('nu = 3 on level 1', CycleNu(H, nu={1: 3}))
    report('nu = 2 on levels 1, 2', CycleNu(H, nu={1: 2, 2: 2}))
    report('nu = 2 on every coarse level', CycleNu(H, nu={l: 2 for l in range(1, nl)}))
    report('nu = 3 on every coarse level', CycleNu(H, nu={l: 3 for l in range(1, nl)}))
    report('Chebyshev degree 2 on level 1', CycleNu(H, cheb={1: 2}))
    report('Chebyshev degree 3 on level 1', CycleNu(H, cheb={1: 3}))
    report('Chebyshev degree 2 on every coarse level', CycleNu(H, cheb={l: 2 for l in range(1, nl - 1)}))
    report('exact solve from level 1 (two-level bound)', CycleNu(H, exact_from=1))
    report('exact solve from level 2', CycleNu(H, exact_from=2))
