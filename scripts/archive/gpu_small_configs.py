import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for name, sw in (("C1", True), ("C1F5", True), ("C2", False)):
    g = graphgen.config(name)
    q, t, s = util.initial_state(g, sw)
    P = util.pgo_problem(g, sw); P.solve(q, t, s); P.close()
    P = util.pgo_problem(g, sw)
    _, _, _, sm = P.solve(q, t, s)
    print(name, 'device %.4f s' % sm.seconds_device, 'cg', sm.cg_iterations, 'lm', sm.num_iterations, 'ok', sm.num_successful_steps, 'cost %.6g -> %.6g' % (sm.initial_cost, sm.final_cost), flush=True)
    P.close()
