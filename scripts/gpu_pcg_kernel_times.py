"""HIP-event times of one block-Jacobi PCG iteration, its matvec and its update alone (pgo_time_kernel 2 / 4 / 5) on several configs."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
for name in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['C3']):
    g = graphgen.config(name)
    sw = name != 'C2'
    q, t, s = util.initial_state(g, sw)
    P = util.pgo_problem(g, sw)
    P.solve_begin(q, t, s)
    for _ in range(2): P.lm_step(ignore_termination=True)
    out = []
    for which in (2, 4, 5):
        try:
            best = min(P.time_kernel(which, 60)[0] for _ in range(3))
        except Exception:      # single-reduction form: the update cannot be launched without its matvec -> iteration minus matvec
            best = (out[0] - out[1]) * 1e-3
        out.append(best * 1e3)
    P.solve_end(); P.close()
    print('%-5s iteration %.2f us  matvec %.2f us  update %.2f us' % (name, out[0], out[1], out[2]), flush=True)
