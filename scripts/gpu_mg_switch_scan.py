import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for name in sys.argv[1].split(','):
    g = graphgen.config(name)
    q, t, s = util.initial_state(g, True)
    for sw in [int(x) for x in sys.argv[2].split(',')]:
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, True, max_num_iterations=20, mg_switch_iterations=sw)
            _, _, _, sm = P.solve(q, t, s); P.close()
            if best is None or sm.seconds_device < best.seconds_device: best = sm
        print(name, 'mg_switch_iterations', sw, 'dev %.4f s' % best.seconds_device, 'cg', best.cg_iterations, [best.iterations[k].cg_iterations for k in range(1, best.num_logged)], flush=True)
