"""Option sets on graph types other than the benchmark's (and on C4 / C5): 20 LM steps (C5: 10) with library defaults vs the given option sets, on fresh handles.
  python scripts/dev/r05/opt_types.py "types,C4,C5" "" "mg_first_passes=2" ..."""
import sys
import time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
which = sys.argv[1].split(',')
sets = sys.argv[2:] or [""]


def parse(txt):
    kw = {}
    for item in (txt.split(',') if txt else []):
        k, x = item.split('=')
        kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
    return kw


cases = []
if 'types' in which:
    cases += [("60k keyframes, 6k loops (chain-like)", lambda: graphgen.generate(60000, 6000, odom_f_max=2, seed=7), True, 20),
              ("60k keyframes, 60k loops, no outliers", lambda: graphgen.generate(60000, 60000, odom_f_max=2, seed=8, outlier_frac=0.0), True, 20),
              ("50k keyframes, 25k loops, f=1..5 + yaw weights", lambda: graphgen.generate(50000, 25000, odom_f_max=5, apply_yaw_weight=True, seed=9), True, 20),
              ("40k keyframes, 40k PLAIN loops (no switches)", lambda: graphgen.generate(40000, 40000, odom_f_max=2, seed=10, outlier_frac=0.0), False, 20),
              ("20k keyframes, 20k loops", lambda: graphgen.generate(20000, 20000, odom_f_max=2, seed=3), True, 20),
              ("12k keyframes, 12k loops", lambda: graphgen.generate(12000, 12000, odom_f_max=2, seed=3), True, 20)]
if 'C2' in which:      # BASELINE config 2 (plain loops, no switch variables) with the reference's 10-iteration budget
    cases.append(("C2 (10k keyframes, 1k PLAIN loops)", lambda: graphgen.config("C2"), False, 10))
if 'C2S' in which:     # the same graph with its loop closures switchable (what the reference's trigger builds)
    cases.append(("C2 graph, loops SWITCHABLE", lambda: graphgen.config("C2"), True, 10))
if 'mid' in which:     # graphs between the reference's sessions and the benchmark sizes
    cases += [("8k keyframes, 2k loops", lambda: graphgen.generate(8000, 2000, odom_f_max=2, seed=21), True, 10),
              ("16k keyframes, 4k loops, f=1..5 + yaw", lambda: graphgen.generate(16000, 4000, odom_f_max=5, apply_yaw_weight=True, seed=22), True, 10),
              ("30k keyframes, 6k PLAIN loops", lambda: graphgen.generate(30000, 6000, odom_f_max=2, seed=23, outlier_frac=0.0), False, 10)]
if 'scan' in which:    # where the multigrid (with the smoothed keyframe transition) takes over from the two-level method: plain and switchable loops, 10 LM iterations
    for n in (2000, 3000, 4000, 6000, 8000, 12000):
        cases.append(("%d keyframes, %d PLAIN loops" % (n, n // 5), (lambda n=n: graphgen.generate(n, n // 5, odom_f_max=2, seed=30 + n // 1000, outlier_frac=0.0)), False, 10))
        cases.append(("%d keyframes, %d switchable loops" % (n, n // 5), (lambda n=n: graphgen.generate(n, n // 5, odom_f_max=2, seed=30 + n // 1000)), True, 10))
        cases.append(("%d keyframes, %d switchable loops, f=1..5 + yaw" % (n, n // 2), (lambda n=n: graphgen.generate(n, n // 2, odom_f_max=5, apply_yaw_weight=True, seed=60 + n // 1000)), True, 10))
for c in ('C3', 'C4', 'C5'):
    if c in which:
        cases.append((c, (lambda c=c: graphgen.config(c)), True, 10 if c == 'C5' else 20))
for name, make, sw, iters in cases:
    g = make()
    q, t, s = util.initial_state(g, sw)
    ref = None
    for txt in sets:
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, sw, max_num_iterations=10 ** 6, cg_max_iterations=200000, **parse(txt))
            P.solve_begin(q, t, s)
            P.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                P.lm_step(ignore_termination=True)
            P.synchronize()
            el = time.perf_counter() - t0
            _, _, _, sm = P.solve_end(); P.close()
            best = el if best is None else min(best, el)
        its = [sm.iterations[k] for k in range(sm.num_logged)]
        if ref is None:
            ref = [i.cost for i in its]
        dev = max(abs(i.cost - r) / max(r, 1e-12) for i, r in zip(its, ref))
        print('%-48s %-34s %.4f s  cg %7d (mg %6d)  max rel cost dev %.1e  final %.9e' % (name, txt or "(defaults)", best, sm.cg_iterations, sm.cg_iterations_multigrid, dev, sm.final_cost), flush=True)
    del g
