"""A/B of the two-level preconditioner (block-Jacobi + rigid-body coarse space): off / on from radius 1e6 (default) / always on."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

def cases():
    yield "mid 13380/6028 f2", graphgen.generate(13380, 6028, odom_f_max=2, seed=11, outlier_frac=0.1), True
    yield "mid 6918/1382 f3", graphgen.generate(6918, 1382, odom_f_max=3, seed=12, outlier_frac=0.3), True
    yield "C2", graphgen.config("C2"), False
    yield "C1F5", graphgen.config("C1F5"), True
    yield "C4", graphgen.config("C4"), True
    yield "C3", graphgen.config("C3"), True
for name, g, sw in cases():
    q, t, s = util.initial_state(g, sw)
    ref = None
    for label, kw in (("off", dict(coarse_aggregates=0)), ("default", dict()), ("always", dict(coarse_min_radius=0.0)), ("256 aggs", dict(coarse_aggregates=256))):
        P = util.pgo_problem(g, sw, **kw)
        P.solve(q, t, s)
        _, tt, ss, sm = P.solve(q, t, s)
        its = [sm.iterations[k] for k in range(sm.num_logged)]
        if ref is None: ref = [i.cost for i in its]
        dev = max(abs(i.cost - r) / max(r, 1e-12) for i, r in zip(its, ref))
        print('%-18s %-8s dev %.3f s cg %7d %s  max rel cost dev %.1e  %s' % (name, label, sm.seconds_device, sm.cg_iterations, [i.cg_iterations for i in its[1:]], dev, ''.join(str(i.step_is_successful) for i in its)), flush=True)
        P.close()
