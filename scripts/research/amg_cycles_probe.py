"""Research probe (CPU, scipy): V / W / K cycles of the rigid-mode aggregation multigrid on a LATE linearisation (8 LM steps in) of a C3-structured graph, with the
product's hierarchy recipe (level 1 along odometry edges, 2^3 keyframes; levels above on summed couplings) — iterations per cycle type, fine level additive
(the product's form) or multiplicative.  Not part of the product or the tests."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from scripts.research import amg_probe as ap
from scripts.research.precond_probe import build_system, block_diag_inv, pcg, fpcg
from solve_keyframe_pose_graph_amd import graphgen
from oracle import binding as ob
from tests import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = graphgen.generate(n, n, odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
O = util.oracle_problem(g, True)
q, t, s, summ = O.solve(q, t, s, ob.default_options(max_num_iterations=8))
q = q.reshape(-1, 4); t = t.reshape(-1, 3)
print('late state cost', summ.final_cost, flush=True)
N = g.n_poses
for radius in [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1e6,1e8').split(',')]:
    A, b = build_system(g, q, t, s, radius)
    Dinv = block_diag_inv(A, N)
    x, k = pcg(A, b, lambda r: Dinv @ r, 1e-9, maxit=60000); print('radius %g block-Jacobi its %d' % (radius, k), flush=True)
    fn = lambda A_, N_, lvl: ap.topo_aggregates(g, N_, 3, loop_w=0.0) if lvl == 0 else ap.graph_aggregates(A_, N_, 3)
    H = ap.Hier(A, t, fn, min_coarse=500)
    for fa in (True, False):
        for (cyc, sm, nu) in (('V', ('jac', 0.9), 1), ('W', ('jac', 0.9), 1), ('K', ('jac', 0.9), 1), ('V', ('jac', 0.9), 2)):
            M = ap.Cycle(H, cyc, sm, nu=nu, fine_additive=fa)
            x2, k2 = (fpcg if cyc == 'K' else pcg)(A, b, M, 1e-9, maxit=3000)
            print('   %s nu=%d fine %s: its %d  work/it %.2f  level visits/it %d' % (cyc, nu, 'additive' if fa else 'multiplicative', k2, M.work / max(k2, 1), M.syncs / max(k2, 1)), flush=True)
