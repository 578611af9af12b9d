"""Session-sized graphs (the reference's f = 1..5 odometry policy, 2 degrees of yaw per keyframe): libpgo on one MI355X next to the CPU
oracle, fresh handle per graph (so the once-per-solve preconditioner comparison is included)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import binding as ob
from solve_keyframe_pose_graph_amd import graphgen
from tests import util

sizes = [int(a) for a in sys.argv[1:]] or [100, 200, 400, 512, 600, 1000, 2000, 4000]
for n in sizes:
    g = graphgen.generate(n, max(2, n // 6), odom_f_max=5, apply_yaw_weight=1, seed=n, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    P = util.pgo_problem(g, True)
    t0 = time.perf_counter(); _, _, _, sm = P.solve(q, t, s); wall = time.perf_counter() - t0
    t0 = time.perf_counter(); _, _, _, sm2 = P.solve(q, t, s); wall2 = time.perf_counter() - t0
    P.close()
    O = util.oracle_problem(g, True)
    t0 = time.perf_counter(); _, _, _, so = O.solve(q, t, s); cpu = time.perf_counter() - t0
    print('n %5d edges %6d: libpgo first solve %.1f ms, repeat %.1f ms (device %.1f ms, %d LM, %d PCG iterations) | CPU oracle %.1f ms (%d LM) | rel cost diff %.1e' % (
        n, g.n_edges if hasattr(g, 'n_edges') else len(g.odom_c1) + len(g.loop_c1), wall * 1e3, wall2 * 1e3, sm2.seconds_device * 1e3, sm2.num_iterations, sm2.cg_iterations, cpu * 1e3, so.num_iterations,
        abs(sm.final_cost - so.final_cost) / max(so.final_cost, 1e-300)), flush=True)
