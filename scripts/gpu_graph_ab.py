import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
for name, sw in (('C1', True), ('C2', False), ('C3', True)):
    g = graphgen.config(name); q,t,s = util.initial_state(g, sw)
    for ug in (0, 1):
        P = util.pgo_problem(g, sw, cg_use_graph=ug)
        P.solve(q,t,s)   # warm (graph capture, allocations)
        t0=time.time(); qq,tt,ss,summ = P.solve(q,t,s); dt=time.time()-t0
        print(name, 'graph', ug, 'LM', summ.num_iterations, 'cg', summ.cg_iterations, 'device %.4fs' % summ.seconds_device, 'us/cg-it %.2f' % (1e6*summ.seconds_device/max(1,summ.cg_iterations)), 'final %.10e' % summ.final_cost)
        P.close()
