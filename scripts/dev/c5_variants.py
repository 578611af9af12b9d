import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config("C5"); q, t, s = util.initial_state(g, True)
for spec in sys.argv[1].split(';'):
    kw = {}
    for item in (spec.split(',') if spec else []):
        k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
    P = util.pgo_problem(g, True, verbosity=0, **kw)
    _, _, _, sm = P.solve(q, t, s); P.close()
    print('C5 %-44s %.3f s  cg %6d (mg %6d)  final %.9e' % (spec or 'defaults', sm.seconds_device, sm.cg_iterations, sm.cg_iterations_multigrid, sm.final_cost), flush=True)
