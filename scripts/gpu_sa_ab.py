"""Smoothed prolongators (mg_smoothed_levels = 0 / 1 / 2) on the benchmark graphs and on other graph types: device seconds and PCG iterations of the
library's default hybrid solve.  usage: gpu_sa_ab.py [c3,c4,c5,types]"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
which = (sys.argv[1] if len(sys.argv) > 1 else "c3,c4,types").split(",")
cases = []
if "c3" in which: cases += [("C3, 20 LM steps", graphgen.config("C3"), True, 20), ("C3, 10 LM steps", graphgen.config("C3"), True, 10)]
if "c4" in which: cases += [("C4, 20 LM steps", graphgen.config("C4"), True, 20)]
if "c5" in which: cases += [("C5, 10 LM steps", graphgen.config("C5"), True, 10)]
if "types" in which:
    cases += [("60k keyframes, 6k loops (chain-like), 20 steps", graphgen.generate(60000, 6000, odom_f_max=2, seed=7), True, 20),
              ("60k keyframes, 60k loops, no outliers, 20 steps", graphgen.generate(60000, 60000, odom_f_max=2, seed=8, outlier_frac=0.0), True, 20),
              ("50k keyframes, 25k loops, f=1..5 + yaw weights, 20 steps", graphgen.generate(50000, 25000, odom_f_max=5, apply_yaw_weight=True, seed=9), True, 20),
              ("40k keyframes, 40k PLAIN loops (no switches), 20 steps", graphgen.generate(40000, 40000, odom_f_max=2, seed=10, outlier_frac=0.0), False, 20)]
for name, g, sw, iters in cases:
    q, t, s = util.initial_state(g, sw)
    ref = None
    for sm_l in (0, 1, 2):
        P = util.pgo_problem(g, sw, max_num_iterations=iters, cg_max_iterations=200000, mg_smoothed_levels=sm_l, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
        _, _, _, sm = P.solve(q, t, s); P.close()
        its = [sm.iterations[k] for k in range(sm.num_logged)]
        if ref is None: ref = [i.cost for i in its]
        dev = max(abs(i.cost - r) / max(r, 1e-12) for i, r in zip(its, ref)) if len(its) == len(ref) else float('nan')
        print('%-58s smoothed %d: %.3f s  cg %7d (multigrid %7d)  LM %d  max rel cost dev %.1e' % (name, sm_l, sm.seconds_device, sm.cg_iterations, sm.cg_iterations_multigrid, sm.num_iterations, dev), flush=True)
