"""rocprofv3 target: 20 LM steps of one graph type of scripts/dev/r05/opt_types.py with the given options (per-kernel times behind the gains and losses of mg_smoothed_fine).
  python scripts/dev/r05/smoothed_fine_type_profile.py <plain40k|f5_50k|noout60k> "<opt=val,...>" """
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
which = sys.argv[1]
kw = {}
for item in (sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] else []):
    k, x = item.split('='); kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
g, sw = {"plain40k": (lambda: graphgen.generate(40000, 40000, odom_f_max=2, seed=10, outlier_frac=0.0), False),
         "f5_50k": (lambda: graphgen.generate(50000, 25000, odom_f_max=5, apply_yaw_weight=True, seed=9), True),
         "noout60k": (lambda: graphgen.generate(60000, 60000, odom_f_max=2, seed=8, outlier_frac=0.0), True)}[which]
g = g()
q, t, s = util.initial_state(g, sw)
P = util.pgo_problem(g, sw, max_num_iterations=10 ** 6, cg_max_iterations=200000, verbosity=1, **kw)
P.solve_begin(q, t, s)
for _ in range(20): P.lm_step(ignore_termination=True)
_, _, _, sm = P.solve_end(); P.close()
print(which, kw, 'cg', sm.cg_iterations, 'mg', sm.cg_iterations_multigrid, flush=True)
