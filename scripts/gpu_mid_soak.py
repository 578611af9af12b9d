#!/usr/bin/env python3
"""Round 6 soak of the NEW default path for graphs of 5 000 - 80 000 keyframes (the multigrid from 5 000 keyframes with the smoothed keyframe transition decided by the density of its
levels): random graphs — size, loop density, odometry policy f = 1..5 with and without yaw weights, outliers, plain and switchable loops, 1-3 worlds — solved with library defaults
and, as the reference, with round 5's choices for that size (mg_smoothed_fine = 0, mg_min_keyframes = 24000, mg_min_keyframes_switchable = 8000) at cg_rel_tolerance 1e-12 and Ceres'
exact decision rule (pauses off).  Checked: same accept/reject sequence, per-iteration costs within BASELINE.json's 1e-6, NO PCG retry on the default path (a preconditioner that is
not positive definite would show there), and the time of both.
  python scripts/gpu_mid_soak.py [n_graphs] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402
from tests import util  # noqa: E402

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 66)
bad = 0
t_new = t_old = 0.0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    n = int(rng.choice([rng.integers(5000, 12000), rng.integers(12000, 40000), rng.integers(40000, 80000)], p=[0.5, 0.35, 0.15]))
    loops = int(n * rng.choice([0.05, 0.1, 0.3, 0.6, 1.0]))
    f = int(rng.choice([1, 2, 3, 5]))
    yaw = bool(f == 5 and rng.random() < 0.7)
    out = float(rng.choice([0.0, 0.1, 0.3]))
    sw = bool(rng.random() < 0.75)
    worlds = int(rng.choice([1, 1, 1, 3]))
    g = graphgen.generate(n, loops, odom_f_max=f, apply_yaw_weight=yaw, n_worlds=worlds, seed=int(rng.integers(1, 10 ** 6)), outlier_frac=out if sw else 0.0)
    q, t, s = util.initial_state(g, sw)
    kw = dict(max_num_iterations=10)
    P = util.pgo_problem(g, sw, **kw)
    P.solve(q, t, s)
    t0 = time.time(); _, tp, sp, sump = P.solve(q, t, s); tn = time.time() - t0; P.close()
    old = dict(mg_smoothed_fine=0, mg_min_keyframes=24000, mg_min_keyframes_switchable=8000)
    R = util.pgo_problem(g, sw, **old, **kw)
    R.solve(q, t, s)
    t0 = time.time(); R.solve(q, t, s); to = time.time() - t0; R.close()
    X = util.pgo_problem(g, sw, cg_rel_tolerance=1e-12, cg_early_tolerance=0.0, cg_mid_tolerance=0.0, **old, **kw)
    _, tr, sr, sumr = X.solve(q, t, s); X.close()
    seq_p = [sump.iterations[i].step_is_successful for i in range(sump.num_logged)]
    seq_r = [sumr.iterations[i].step_is_successful for i in range(sumr.num_logged)]
    dev = max(abs(sumr.iterations[i].cost - sump.iterations[i].cost) / max(sumr.iterations[i].cost, 1e-12) for i in range(min(sumr.num_logged, sump.num_logged)))
    ok = seq_p == seq_r and dev <= 1e-6 and sump.pcg_retries == 0
    bad += 0 if ok else 1
    t_new += tn; t_old += to
    print('%s n %5d loops %5d f %d%s outliers %.1f %s worlds %d: %s dev %.1e | defaults %.4f s cg %6d (mg %6d, retries %d) | round-5 choices %.4f s | x%.2f' % (
        'ok ' if ok else 'BAD', n, g.n_loops, f, '+yaw' if yaw else '', out if sw else 0.0, 'switchable' if sw else 'plain     ', worlds, ''.join(map(str, seq_p)), dev, tn, sump.cg_iterations,
        sump.cg_iterations_multigrid, sump.pcg_retries, to, to / tn), flush=True)
print('mismatches %d; total time defaults %.3f s, round-5 choices %.3f s (x%.2f)' % (bad, t_new, t_old, t_old / max(t_new, 1e-9)))
sys.exit(1 if bad else 0)
