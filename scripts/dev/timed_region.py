"""The timed region of bench.py alone, with the device idle for 0.3 s on both sides: the target of `rocprofv3 --kernel-trace` for scripts/rocpd_summary.py segments.
  python scripts/dev/timed_region.py [config | S<keyframes>] [steps] [verbosity] [opt=val,...]"""
import sys
import time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
verb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kw = {}      # optional fourth argument: library options "opt=val,..."
for item in (sys.argv[4].split(',') if len(sys.argv) > 4 and sys.argv[4] else []):
    key, x = item.split('='); kw[key] = float(x) if ('.' in x or 'e' in x) else int(x)
if name.startswith('S'):
    n = int(name[1:])
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))      # the session structure of scripts/research/session_step_times.py
else:
    g = graphgen.config(name)
q0, t0, s0 = g.init_q, g.init_t, np.full(g.n_loops, 0.99)
for leg in range(2):      # warm-up leg on its own handle, then the timed one (as bench.py does)
    P = capi.problem_from_graph(g, switchable=True, max_num_iterations=10 ** 6, verbosity=verb if leg == 1 else 0, **kw)
    P.solve_begin(q0, t0, s0)
    P.synchronize()
    time.sleep(0.3)
    t = time.perf_counter()
    for _ in range(steps):
        P.lm_step(ignore_termination=True)
    P.synchronize()
    el = time.perf_counter() - t
    time.sleep(0.3)
    _, _, _, sm = P.solve_end()
    P.close()
print(name, kw, 'steps', steps, 'seconds', el, 'it/s', steps / el, 'cg', sm.cg_iterations, 'mg', sm.cg_iterations_multigrid)
