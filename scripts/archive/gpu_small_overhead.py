"""Per-solve overheads on the reference-sized graphs: first solve (graph build + hipGraph capture), repeat, and after one added edge
(the rebuild every trigger pays); with and without hipGraph replay of the PCG chunks."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
for n, loops in ((200, 20), (1000, 150), (3000, 500)):
    g = graphgen.generate(n, loops, odom_f_max=5, apply_yaw_weight=1, seed=3, **graphgen._SMALL)
    q, t, s = util.initial_state(g, True)
    for use_graph in (1, 0):
        P = util.pgo_problem(g, True, cg_use_graph=use_graph)
        t0 = time.perf_counter(); _, _, _, sm = P.solve(q, t, s); first = time.perf_counter() - t0
        t0 = time.perf_counter(); _, _, _, sm2 = P.solve(q, t, s); second = time.perf_counter() - t0
        # a new edge forces a graph rebuild (what every trigger does)
        P.add_relpose_edges(g.odom_c1[:1], g.odom_c2[:1], g.odom_T[:1], g.odom_w[:1])
        t0 = time.perf_counter(); _, _, _, sm3 = P.solve(q, t, s); third = time.perf_counter() - t0
        print('n %5d hipGraph %d: first solve %.1f ms (device %.1f), repeat %.1f ms (device %.1f), after add_edge %.1f ms (device %.1f); cg %d, %d LM its' % (
            n, use_graph, first * 1e3, sm.seconds_device * 1e3, second * 1e3, sm2.seconds_device * 1e3, third * 1e3, sm3.seconds_device * 1e3, sm2.cg_iterations, sm2.num_iterations), flush=True)
        P.close()
