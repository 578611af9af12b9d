"""Does the 10-iteration chi^2 of C3 need every step solved to 1e-9?  Tolerance schedules via set_options between LM steps."""
import sys; sys.path.insert(0, '.')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config(sys.argv[1] if len(sys.argv) > 1 else "C3"); q, t, s = util.initial_state(g, True)
ref = None
for name, sched in (("1e-12 all", [1e-12] * 10), ("1e-9 all", [1e-9] * 10), ("1e-6 x3 then 1e-9", [1e-6] * 3 + [1e-9] * 7), ("1e-5 x3 then 1e-9", [1e-5] * 3 + [1e-9] * 7),
                    ("1e-7 x3 then 1e-9", [1e-7] * 3 + [1e-9] * 7), ("1e-6 x8 then 1e-9", [1e-6] * 8 + [1e-9] * 2)):
    P = util.pgo_problem(g, True, cg_max_iterations=20000)
    P.solve_begin(q, t, s)
    for k in range(10):
        P.set_options(cg_rel_tolerance=sched[k])
        P.lm_step()
    qq, tt, ss, sm = P.solve_end()
    its = [sm.iterations[k] for k in range(sm.num_logged)]
    if ref is None: ref = [i.cost for i in its]
    print('%-20s cg %6d final %.10e  max rel cost dev over iterations %.2e  final dev %.2e  %s' % (name, sm.cg_iterations, sm.final_cost,
          max(abs(i.cost - r) / r for i, r in zip(its, ref)), abs(sm.final_cost - ref[-1]) / ref[-1], ''.join(str(i.step_is_successful) for i in its)), flush=True)
    P.close()
