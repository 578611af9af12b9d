"""Soak of the hybrid schedule on multigrid-sized graphs (25k-70k keyframes: the oracle's exact solve is out of reach there): library defaults against the same solve with the
multigrid from the first iteration of every system at cg_rel_tolerance 1e-12 — accept/reject sequence and per-iteration costs within BASELINE.json's 1e-6 for the 12-iteration budget.
python scripts/gpu_fuzz_soak_multigrid.py [n_graphs]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    n = int(rng.integers(25000, 70000)); loops = int(n * rng.uniform(0.1, 1.2)); f = int(rng.integers(1, 4)); out = float(rng.choice([0.0, 0.1, 0.3])); seed = int(rng.integers(1, 10**6))
    g = graphgen.generate(n, loops, odom_f_max=f, seed=seed, outlier_frac=out)
    q, t, s = util.initial_state(g, True)
    kw = dict(max_num_iterations=12)
    P = util.pgo_problem(g, True, **kw); t0 = time.time(); _, tp, sp, sump = P.solve(q, t, s); tg = time.time() - t0; P.close()
    R = util.pgo_problem(g, True, cg_rel_tolerance=1e-12, mg_switch_iterations=0, cg_early_tolerance=0.0, cg_mid_tolerance=0.0, **kw); _, tr, sr, sumr = R.solve(q, t, s); R.close()
    seq_p = [sump.iterations[i].step_is_successful for i in range(sump.num_logged)]
    seq_r = [sumr.iterations[i].step_is_successful for i in range(sumr.num_logged)]
    dev = max(abs(sumr.iterations[i].cost - sump.iterations[i].cost) / max(sumr.iterations[i].cost, 1e-12) for i in range(min(sumr.num_logged, sump.num_logged)))
    ok = seq_p == seq_r and dev <= 1e-6 and sump.termination_type == sumr.termination_type
    bad += 0 if ok else 1
    print('%s n %5d loops %5d f %d outliers %.1f: %s, max rel cost dev %.1e, cg %d (multigrid %d) vs %d, %.2f s' % ('ok ' if ok else 'BAD', n, g.n_loops, f, out,
          ''.join(map(str, seq_p)), dev, sump.cg_iterations, sump.cg_iterations_multigrid, sumr.cg_iterations, tg), flush=True)
print('mismatches', bad)
