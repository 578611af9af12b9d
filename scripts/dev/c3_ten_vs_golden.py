import sys, json
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
gold = json.load(open('/root/repo/tests/golden/c3_ten_iterations.json'))
g = graphgen.config('C3'); q, t, s = util.initial_state(g, True)
for name, kw in (('defaults', {}), ('no pauses', dict(cg_early_tolerance=0.0, cg_mid_tolerance=0.0)), ('tol 3e-10', dict(cg_rel_tolerance=3e-10)), ('tol 1e-10', dict(cg_rel_tolerance=1e-10)), ('tol 1e-12', dict(cg_rel_tolerance=1e-12))):
    P = util.pgo_problem(g, True, **kw); _, _, _, sm = P.solve(q, t, s); P.close()
    print('%-10s final %.12e  cg %5d  %.4f s' % (name, sm.final_cost, sm.cg_iterations, sm.seconds_device), [('%.3e' % (sm.iterations[k].cost)) for k in (8, 9, 10)], flush=True)
print({k: gold[k] for k in gold if not isinstance(gold[k], (list, dict))})
