"""One-off soak at medium size: C3-structured graphs (3k-15k keyframes, the default generator) with random seeds / loop density / outlier
share / odometry policy, library defaults vs the oracle's exact Cholesky for the 10-iteration budget."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

rng = np.random.default_rng(77)
bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    n = int(rng.integers(3000, 15000)); loops = int(n * rng.uniform(0.1, 1.2)); f = int(rng.integers(1, 4)); out = float(rng.choice([0.0, 0.1, 0.3])); seed = int(rng.integers(1, 10**6))
    g = graphgen.generate(n, loops, odom_f_max=f, seed=seed, outlier_frac=out)
    q, t, s = util.initial_state(g, True)
    O, P = util.oracle_problem(g, True), util.pgo_problem(g, True)
    t0 = time.time(); qo, to, so, sumo = O.solve(q, t, s); tc = time.time() - t0
    t0 = time.time(); qp, tp, sp, sump = P.solve(q, t, s); tg = time.time() - t0
    P.close()
    seq_o = [sumo.iterations[i].step_is_successful for i in range(sumo.num_logged)]
    seq_p = [sump.iterations[i].step_is_successful for i in range(sump.num_logged)]
    dev = max(abs(sumo.iterations[i].cost - sump.iterations[i].cost) / max(sumo.iterations[i].cost, 1e-12) for i in range(min(sumo.num_logged, sump.num_logged)))
    ok = seq_o == seq_p and dev <= 1e-6
    bad += 0 if ok else 1
    print('%s n %5d loops %5d f %d outliers %.1f: %s, max rel cost dev %.1e, cg %d, gpu %.2f s, oracle %.1f s' % ('ok ' if ok else 'BAD', n, g.n_loops, f, out,
          ''.join(map(str, seq_p)), dev, sump.cg_iterations, tg, tc), flush=True)
print('mismatches', bad)
