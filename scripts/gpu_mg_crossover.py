"""Where the aggregation multigrid overtakes the two-level method: four graph types at several sizes, library defaults with the multigrid off (mg_min_keyframes = 0: two-level method) / forced on (1),
the reference's 10-iteration budget.  python scripts/gpu_mg_crossover.py 6000,10000,16000"""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '6000,10000,16000').split(',')]
for n in sizes:
    cases = [("chain-like (n/10 loops, f=1,2)", graphgen.generate(n, n // 10, odom_f_max=2, seed=7), True),
             ("no outliers (n loops)", graphgen.generate(n, n, odom_f_max=2, seed=8, outlier_frac=0.0), True),
             ("f=1..5 + yaw weights (n/2 loops)", graphgen.generate(n, n // 2, odom_f_max=5, apply_yaw_weight=True, seed=9), True),
             ("plain loops (n, no switches)", graphgen.generate(n, n, odom_f_max=2, seed=10, outlier_frac=0.0), False),
             ("session (n/5 loops, f=1..5, 2-degree turns)", graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0)), True)]
    for name, g, sw in cases:
        q, t, s = util.initial_state(g, sw)
        out = []
        for kw in (dict(mg_min_keyframes=0), dict(mg_min_keyframes=1)):
            best = None
            for rep in range(2):
                P = util.pgo_problem(g, sw, max_num_iterations=10, **kw); _, _, _, sm = P.solve(q, t, s); P.close()
                if best is None or sm.seconds_device < best.seconds_device: best = sm
            out.append(best)
        a, b = out
        print('%6d %-46s two-level %.4f s cg %6d | multigrid %.4f s cg %6d | ratio %.2f  cost dev %.1e' % (n, name, a.seconds_device, a.cg_iterations, b.seconds_device, b.cg_iterations,
              a.seconds_device / b.seconds_device, abs(a.final_cost - b.final_cost) / max(a.final_cost, 1e-12)), flush=True)
