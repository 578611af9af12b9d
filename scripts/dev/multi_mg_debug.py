"""Dev helper: single handle vs N in-process ranks with the multigrid forced on a small graph; prints the per-iteration logs side by side."""
import sys, threading
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding
from tests import util
from tests.test_gpu_two_ranks_one_gpu import InProcessAllReduce
world = int(sys.argv[1]); policy = sys.argv[2]; ls = int(sys.argv[3]); switch_at = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = graphgen.generate(6000, 3000, odom_f_max=2, seed=7)
q, t, s = util.initial_state(g, True)
mgmin = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
opts = dict(mg_min_keyframes=mgmin, mg_switch_iterations=switch_at, cg_rel_tolerance=float(sys.argv[6]) if len(sys.argv) > 6 else 1e-11, linear_solver=ls, max_num_iterations=int(sys.argv[7]) if len(sys.argv) > 7 else 8, mg_dense_max_nodes=64)
P = util.pgo_problem(g, True, **opts)
_, _, _, s1 = P.solve(q, t, s); P.close()
parts = sharding.partition(g, world, policy)
ar = InProcessAllReduce(world); out = [None] * world
def run(rank):
    try:
        Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **opts)
        Pr.comm_init_custom(rank, world, ar.make(rank)); out[rank] = Pr.solve(q, t, s); Pr.comm_destroy(); Pr.close()
    except Exception as e:
        print('rank', rank, 'failed', e); ar.barrier.abort()
th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[x.start() for x in th]; [x.join() for x in th]
sr = out[0][3]
for k in range(max(s1.num_logged, sr.num_logged)):
    a, b = s1.iterations[k], sr.iterations[k]
    print('%2d single: cost %.12e ok %d rho %.3e cg %5d res %.1e | ranks: cost %.12e ok %d rho %.3e cg %5d res %.1e' % (k, a.cost, a.step_is_successful, a.relative_decrease, a.cg_iterations, a.cg_residual, b.cost, b.step_is_successful, b.relative_decrease, b.cg_iterations, b.cg_residual))
