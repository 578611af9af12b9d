"""One-off soak of the library DEFAULTS (PCG tolerance 1e-9, two-stage early rejection) against the oracle's exact solve on many random
synthetic graphs with outliers (rejected steps!): accept/reject sequence, per-iteration cost, final cost after the 10-iteration budget."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from tests import util

n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad, rejected_total, early = [], 0, 0
t0 = time.time()
for k in range(n_graphs):
    n = int(rng.integers(80, 1500)); loops = int(rng.integers(5, max(6, n // 3))); f = int(rng.integers(1, 6))
    out = float(rng.choice([0.0, 0.1, 0.2, 0.4])); seed = int(rng.integers(1, 10**6))
    g = util.small_graph(n, loops, f=f, seed=seed, outlier_frac=out, min_loop_gap=int(rng.integers(5, 30)))
    if g.n_loops == 0:
        continue
    q, t, s = util.initial_state(g, True)
    O, P = util.oracle_problem(g, True), util.pgo_problem(g, True)
    qo, to, so, sumo = O.solve(q, t, s)
    qp, tp, sp, sump = P.solve(q, t, s)
    P.close()
    seq_o = [sumo.iterations[i].step_is_successful for i in range(sumo.num_logged)]
    seq_p = [sump.iterations[i].step_is_successful for i in range(sump.num_logged)]
    rejected_total += seq_o.count(0)
    ok = seq_o == seq_p and abs(sump.final_cost - sumo.final_cost) <= 1e-6 * max(sumo.final_cost, 1e-12)
    if ok:
        ok = all(abs(sumo.iterations[i].cost - sump.iterations[i].cost) <= 1e-6 * max(sumo.iterations[i].cost, 1e-12) for i in range(sumo.num_logged))
    if not ok:
        bad.append((k, n, g.n_loops, f, out, seed, seq_o, seq_p, sumo.final_cost, sump.final_cost))
        print("MISMATCH", bad[-1], flush=True)
print("graphs %d, rejected steps in total %d, mismatches %d, %.0f s" % (n_graphs, rejected_total, len(bad), time.time() - t0))
