"""Per-LM-step wall time vs PCG iterations (fits step seconds = a + b x iterations): where the non-PCG time of a solve goes."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
for rep in range(2):
    P = util.pgo_problem(g, True, max_num_iterations=iters)
    _, _, _, sm = P.solve(q, t, s)
    P.close()
its = np.array([sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], float)
sec = np.array([sm.iterations[k].seconds for k in range(1, sm.num_logged)])
ok = np.array([sm.iterations[k].step_is_successful for k in range(1, sm.num_logged)])
A = np.stack([np.ones_like(its), its, ok], 1)
coef, *_ = np.linalg.lstsq(A, sec, rcond=None)
print('step seconds ~ %.3f ms + %.2f us x iterations + %.3f ms if accepted' % (coef[0] * 1e3, coef[1] * 1e6, coef[2] * 1e3))
for k in range(len(its)):
    print('  step %2d  cg %5d  %.3f ms  %s   (%.1f us/it)' % (k + 1, its[k], sec[k] * 1e3, 'ok' if ok[k] else 'REJ', 1e6 * sec[k] / max(its[k], 1)))
print('iteration 0: %.3f ms; device total %.4f s' % (sm.iterations[0].seconds * 1e3, sm.seconds_device))
