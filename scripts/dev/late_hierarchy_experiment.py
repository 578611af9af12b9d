"""EXPERIMENT: what would the late LM systems of C3 cost if the multigrid hierarchy knew which loop closures end up switched off?  Pass 1 solves C3 (20 steps) and
writes the final switch values; pass 2 (PGO_MG_SW_FILE set) builds the hierarchy from them but solves the SAME problem from the same start."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
g = graphgen.config("C3"); q, t, s = util.initial_state(g, True)
def run(**kw):
    P = util.pgo_problem(g, True, max_num_iterations=20, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, **kw)
    out = P.solve(q, t, s); P.close(); return out
mode = sys.argv[1]
if mode == "write":
    _, _, sf, sm = run()
    sf.astype(np.float64).tofile("/tmp/c3_switches.bin")
    print("pass 1: %.4f s cg %d" % (sm.seconds_device, sm.cg_iterations), [sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], "dead (<0.1):", int((sf < 0.1).sum()))
else:
    for disc in (3.0, 0.0):
        _, _, sf, sm = run(mg_loop_discount=disc)
        print("pass 2 (hierarchy from final switches, discount %.0f): %.4f s cg %d" % (disc, sm.seconds_device, sm.cg_iterations), [sm.iterations[k].cg_iterations for k in range(1, sm.num_logged)], 'final %.9e' % sm.final_cost)
