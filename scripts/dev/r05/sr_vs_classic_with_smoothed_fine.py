import sys; sys.path.insert(0, '.')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for n, seed, it in ((12000, 3, 14), (30000, 5, 16)):
    g = graphgen.generate(n, n, odom_f_max=2, seed=seed)
    q, t, s = util.initial_state(g, True)
    for fine in (0, 1):
        row = []
        for sr in (0, 1):
            P = util.pgo_problem(g, True, cg_single_reduction=sr, mg_smoothed_fine=fine, max_num_iterations=it)
            _, _, _, sm = P.solve(q, t, s); P.close()
            row.append((sm.cg_iterations, sm.cg_iterations_multigrid, [sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)]))
        print("G%d fine=%d  classic cg %d (mg %d)  single-reduction cg %d (mg %d)" % (n, fine, row[0][0], row[0][1], row[1][0], row[1][1]))
        print("   classic per step", row[0][2]); print("   single  per step", row[1][2], flush=True)
