"""Scan one option over values on given configs: device seconds and PCG iterations of a 20-step solve.  python scripts/gpu_opt_scan.py C3,C4 mg_omega 0.7,0.8,0.9,1.0"""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for name in sys.argv[1].split(','):
    g = graphgen.config(name)
    q, t, s = util.initial_state(g, True)
    for v in sys.argv[3].split(','):
        val = float(v) if ('.' in v or 'e' in v) else int(v)
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, True, max_num_iterations=int(sys.argv[4]) if len(sys.argv) > 4 else 20, **{sys.argv[2]: val})
            _, _, _, sm = P.solve(q, t, s); P.close()
            if best is None or sm.seconds_device < best.seconds_device: best = sm
        print(name, sys.argv[2], v, 'dev %.4f s' % best.seconds_device, 'cg', best.cg_iterations, 'final %.9e' % best.final_cost, flush=True)
