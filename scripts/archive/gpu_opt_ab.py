"""A/B of solver options on one config: device seconds of the K-iteration solve per variant.  usage: gpu_opt_ab.py C3 20 "a=1,b=2;c=3" """
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

name = sys.argv[1]; iters = int(sys.argv[2]); variants = sys.argv[3].split(';')
g = graphgen.config(name) if name.startswith('C') else graphgen.generate(int(name), int(name), odom_f_max=2, seed=3)
sw = name != 'C2'
q, t, s = util.initial_state(g, sw)
for v in variants:
    kw = {}
    for item in v.split(','):
        if '=' in item:
            k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
    best = None
    for rep in range(2):
        P = util.pgo_problem(g, sw, max_num_iterations=iters, **kw)
        _, tt, ss, sm = P.solve(q, t, s)
        P.close()
        best = sm.seconds_device if best is None else min(best, sm.seconds_device)
    print('%-6s %-50s dev %.4f s  cg %6d  -> %.2f us per PCG iteration (all-in)  final %.9e' % (name, v, best, sm.cg_iterations, 1e6 * best / max(sm.cg_iterations, 1), sm.final_cost), flush=True)
