"""The multigrid default (graphs >= 32 000 keyframes) against block-Jacobi and against round 1's policy on graph TYPES other than the benchmark's:
sparse loop closures (chain-like), no outliers, the reference's f = 1..5 odometry policy with yaw weights, plain (non-switchable) loops."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cases = [("60k keyframes, 6k loops (chain-like)", graphgen.generate(60000, 6000, odom_f_max=2, seed=7), True),
         ("60k keyframes, 60k loops, no outliers", graphgen.generate(60000, 60000, odom_f_max=2, seed=8, outlier_frac=0.0), True),
         ("50k keyframes, 25k loops, f=1..5 + yaw weights", graphgen.generate(50000, 25000, odom_f_max=5, apply_yaw_weight=True, seed=9), True),
         ("40k keyframes, 40k PLAIN loops (no switches)", graphgen.generate(40000, 40000, odom_f_max=2, seed=10, outlier_frac=0.0), False)]
for name, g, sw in cases:
    q, t, s = util.initial_state(g, sw)
    ref = None
    for label, kw in (("block-Jacobi only", dict(mg_min_keyframes=0, coarse_aggregates=0)), ("two-level policy (round 1)", dict(mg_min_keyframes=0, coarse_min_radius=1e5)), ("default (multigrid hybrid)", dict())):
        P = util.pgo_problem(g, sw, max_num_iterations=iters, cg_max_iterations=200000, **kw)
        _, _, _, sm = P.solve(q, t, s); P.close()
        its = [sm.iterations[k] for k in range(sm.num_logged)]
        if ref is None: ref = [i.cost for i in its]
        dev = max(abs(i.cost - r) / max(r, 1e-12) for i, r in zip(its, ref)) if len(its) == len(ref) else float('nan')
        print('%-48s %-28s %.3f s  cg %7d  LM %d  max rel cost dev %.1e  final %.6e' % (name, label, sm.seconds_device, sm.cg_iterations, sm.num_iterations, dev, sm.final_cost), flush=True)
