import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from scripts.research.precond_probe import block_diag_inv, pcg
from scripts.research.r3_cycle_probe import load, Hier, agg_product
from scripts.research.r5_smoother_probe import CycleNu
path = sys.argv[1]
for radius in (1e6, 1e4):
    g, t, A, b, s = load(path, radius)
    N = len(t)
    Dinv = block_diag_inv(A, N)
    xref, kbj = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000)
    print("radius %g: block-Jacobi %d its" % (radius, kbj), flush=True)
    for sl, op in (((1,), 0.6), ((0, 1), 0.6), ((0,), 0.6), ((0, 1), 0.45)):
        H = Hier(A, t, agg_product(g, 3, 2), smooth_levels=sl, omega_p=op)
        for name, M in (('V(1,1), additive fine level', CycleNu(H)), ('exact from level 1', CycleNu(H, exact_from=1))):
            t0 = time.time(); x2, k2 = pcg(A, b, M, 1e-9, maxit=3000)
            err = np.abs(x2 - xref).max() / np.abs(xref).max()
            print('   smoothed transitions %-8s omega_p %.2f  %-30s its %4d  err %.0e' % (sl, op, name, k2, err), flush=True)
