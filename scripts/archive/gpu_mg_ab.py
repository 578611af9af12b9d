"""A/B of the aggregation multigrid (pgo.h: mg_min_keyframes): off (two-level / block-Jacobi policy of the library) vs on, same graphs, same
LM budget: PCG iterations per LM step, device seconds, per-iteration cost deviation."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

which = sys.argv[1].split(',') if len(sys.argv) > 1 else ['3k', '20k', 'C3']
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = sys.argv[3].split(';') if len(sys.argv) > 3 else ['off', 'on']

def cases():
    if '3k' in which: yield "3k f2", graphgen.generate(3000, 3000, odom_f_max=2, seed=5), True
    if '20k' in which: yield "20k f2", graphgen.generate(20000, 20000, odom_f_max=2, seed=3), True
    if 'C2' in which: yield "C2", graphgen.config("C2"), False
    if 'C4' in which: yield "C4", graphgen.config("C4"), True
    if 'C3' in which: yield "C3", graphgen.config("C3"), True
    if 'C5' in which: yield "C5", graphgen.config("C5"), True

def parse(v):
    if v == 'off': return dict(mg_min_keyframes=0)
    if v == 'on': return dict(mg_min_keyframes=1)
    kw = dict(mg_min_keyframes=1)
    for item in v.split(','):
        k, x = item.split('=')
        kw[k] = float(x) if '.' in x or 'e' in x else int(x)
    return kw

for name, g, sw in cases():
    q, t, s = util.initial_state(g, sw)
    ref = None
    for v in variants:
        t0 = time.time()
        P = util.pgo_problem(g, sw, max_num_iterations=iters, verbosity=1 if '-v' in sys.argv else 0, **parse(v))
        _, tt, ss, sm = P.solve(q, t, s)
        wall = time.time() - t0
        its = [sm.iterations[k] for k in range(sm.num_logged)]
        if ref is None: ref = [i.cost for i in its]
        dev = max(abs(i.cost - r) / max(r, 1e-12) for i, r in zip(its, ref))
        print('%-8s %-40s dev %.3f s (wall %.2f) cg %7d %s  max rel cost dev %.1e  %s final %.9e' % (name, v, sm.seconds_device, wall, sm.cg_iterations, [i.cg_iterations for i in its[1:]], dev,
                                                                                       ''.join(str(i.step_is_successful) for i in its), sm.final_cost), flush=True)
        P.close()
