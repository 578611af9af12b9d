"""A/B of option sets inside ONE process on one box: the timed region of bench.py (fresh handle, solve_begin outside, K LM steps) alternately with every option set, `reps` times.
  python scripts/dev/r05/ab_options.py C3 20 3 "" "cg_single_reduction=0" "cg_pause_always=1" ..."""
import sys
import time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
name, steps, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sets = sys.argv[4:] or [""]
if name.startswith('N'):
    g = graphgen.generate(int(name[1:]), int(name[1:]), odom_f_max=2, seed=3)
elif name.startswith('S'):      # session structure (scripts/research/session_step_times.py)
    g = graphgen.generate(int(name[1:]), int(name[1:]) // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
else:
    g = graphgen.config(name)
q0, t0, s0 = g.init_q, g.init_t, np.full(g.n_loops, 0.99)


def parse(txt):
    kw = {}
    for item in (txt.split(',') if txt else []):
        k, x = item.split('=')
        kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
    return kw


for r in range(reps + 1):      # round 0 = warm-up, not printed
    for txt in sets:
        P = capi.problem_from_graph(g, switchable=True, max_num_iterations=10 ** 6, **parse(txt))
        P.solve_begin(q0, t0, s0)
        P.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            P.lm_step(ignore_termination=True)
        P.synchronize()
        el = time.perf_counter() - t
        ms = {}
        for which, label in ((2, 'bj_it'), (4, 'matvec'), (5, 'update'), (6, 'mg_it'), (7, 'mg_cycle')) if not name.startswith('S') else ():
            try:
                ms[label] = P.time_kernel(which, 50)[0] * 1e3
            except capi.PgoError:
                ms[label] = ms['bj_it'] - ms['matvec'] if label == 'update' and 'matvec' in ms else float('nan')
        _, _, _, sm = P.solve_end()
        P.close()
        for label in ('bj_it', 'matvec', 'update', 'mg_it', 'mg_cycle'):
            ms.setdefault(label, float('nan'))
        if r:
            print("%-44s #%d  %.4f s  %.2f it/s  cg %d (mg %d)  final cost %.12e | us: bj %.2f (mv %.2f up %.2f) mg %.2f (cycle %.2f)" %
                  (txt or "(defaults)", r, el, steps / el, sm.cg_iterations, sm.cg_iterations_multigrid, sm.final_cost, ms['bj_it'], ms['matvec'], ms['update'], ms['mg_it'], ms['mg_cycle']), flush=True)
