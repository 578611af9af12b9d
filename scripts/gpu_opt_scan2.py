"""C3 / C4, 20 LM steps with library defaults plus one option set per line: device seconds, PCG iterations.  usage: gpu_opt_scan2.py C3 "mg_passes=2;mg_passes=3,mg_omega=0.8" """
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config(sys.argv[1]); q, t, s = util.initial_state(g, True)
for spec in [""] + (sys.argv[2].split(';') if len(sys.argv) > 2 else []):
    kw = {}
    for item in (spec.split(',') if spec else []):
        k, x = item.split('='); kw[k] = float(x) if '.' in x or 'e' in x else int(x)
    best = None
    for rep in range(2):
        P = util.pgo_problem(g, True, max_num_iterations=20, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, **kw)
        _, _, _, sm = P.solve(q, t, s); P.close()
        if best is None or sm.seconds_device < best.seconds_device: best = sm
    print('%-4s %-52s %.4f s  cg %6d (mg %6d)  final %.9e' % (sys.argv[1], spec or 'defaults', best.seconds_device, best.cg_iterations, best.cg_iterations_multigrid, best.final_cost), flush=True)
