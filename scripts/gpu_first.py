import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
from oracle import binding as ob
g = graphgen.config('C1')
O = util.oracle_problem(g); P = util.pgo_problem(g, verbosity=1)
q,t,s = util.initial_state(g, True, perturb=0.01, seed=3)
co,ro,go = O.evaluate(q,t,s); cp,rp,gp = P.evaluate(q,t,s)
print('cost', co, cp, 'res diff', np.abs(ro-rp).max(), 'grad diff', np.abs(go-gp).max())
q,t,s = util.initial_state(g, True)
qo,to,so,sumo = O.solve(q,t,s, ob.default_options(verbosity=1))
qp,tp,sp,sump = P.solve(q,t,s)
print('oracle', sumo.final_cost, sumo.num_iterations, 'gpu', sump.final_cost, sump.num_iterations, sump.message, 'cg', sump.cg_iterations)
print('pose diff', np.abs(tp-to).max(), np.abs(sp-so).max())
