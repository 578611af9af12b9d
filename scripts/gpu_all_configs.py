"""10-iteration solves of every BASELINE.json config on one GPU: LM iterations/s, CG iterations, chi^2, K1 bandwidth."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
for name, sw in (('C1', True), ('C1F5', True), ('C2', False), ('C3', True), ('C4', True), ('C5', True)):
    g = graphgen.config(name); q,t,s = util.initial_state(g, sw)
    P = util.pgo_problem(g, sw)
    P.solve(q,t,s)
    t0=time.time(); qq,tt,ss,summ = P.solve(q,t,s); dt=time.time()-t0
    P.solve_begin(q,t,s); ms,by = P.time_kernel(0, 20); P.solve_end()
    print('%-5s N %7d E %8d | LM %2d (%d ok) in %.4fs device = %.2f it/s | cg %6d | chi2 %.6e -> %.6e | K1 %.1f us %.0f GB/s | %s' % (
        name, g.n_poses, g.n_odom+g.n_loops, summ.num_iterations, summ.num_successful_steps, summ.seconds_device, summ.num_iterations/summ.seconds_device, summ.cg_iterations,
        2*summ.initial_cost, 2*summ.final_cost, ms*1e3, by/ms/1e6, summ.message.decode()), flush=True)
    P.close()
