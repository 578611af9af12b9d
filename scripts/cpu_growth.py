#!/usr/bin/env python3
"""How the CPU port's time per LM iteration grows with the graph (C3-structured graphs of increasing size, same generator and seed, 1 thread, 3 LM iterations each): the measured points behind
any extrapolation to the full C3 graph, whose own run (scripts/cpu_c3_full.py) takes many hours.  python scripts/cpu_growth.py 12000,24000,48000,72000 [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as ob
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12000,24000,48000").split(",")]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_cpu_growth.json")
rows = []
for n in sizes:
    g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    O = util.oracle_problem(g, True)
    q, t, s = util.initial_state(g, True)
    opt = ob.default_options(max_num_iterations=3, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, num_threads=1)
    t0 = time.time()
    _, _, _, sm = O.solve(q, t, s, opt)
    rows.append(dict(poses=n, edges=int(g.n_odom + g.n_loops), lm_iterations=int(sm.num_iterations), seconds_total=sm.seconds_total, seconds_per_lm_iteration=sm.seconds_total / max(1, sm.num_iterations),
                     seconds_linear_solver=sm.seconds_linear_solver, cholesky_fill_blocks=int(sm.chol_nnz_blocks), wall=time.time() - t0))
    print(rows[-1], flush=True)
    if len(rows) >= 2:
        a, b = rows[-2], rows[-1]
        print("  time exponent vs edges between the last two sizes: %.2f; fill exponent %.2f" % (np.log(b["seconds_per_lm_iteration"] / a["seconds_per_lm_iteration"]) / np.log(b["edges"] / a["edges"]),
                                                                                                np.log(b["cholesky_fill_blocks"] / a["cholesky_fill_blocks"]) / np.log(b["edges"] / a["edges"])), flush=True)
    cpu = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
    except Exception:
        pass
    with open(out, "w") as f:
        json.dump(dict(kind="port", threads=1, host_cpu=cpu, note="oracle/pgo_oracle.cpp, exact up-looking block Cholesky (AMD ordering), C3-structured graphs (graphgen.generate(n, n, odom_f_max=2, seed=3)), 3 LM iterations each", rows=rows), f, indent=1)
