"""Research helper (CPU): cache a LATE linearisation (after `steps` oracle LM iterations) of a C3-structured graph as .npz so that cycle probes start in seconds.
Not part of the product or the tests."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, scipy.sparse as sp
from scripts.research.precond_probe import build_system
from solve_keyframe_pose_graph_amd import graphgen
from oracle import binding as ob
from tests import util
n = int(sys.argv[1]); steps = int(sys.argv[2]); out = sys.argv[3]
radii = [float(x) for x in sys.argv[4].split(',')]
g = graphgen.generate(n, n, odom_f_max=2, seed=3)
q, t, s = util.initial_state(g, True)
if steps > 0:
    O = util.oracle_problem(g, True)
    q, t, s, summ = O.solve(q, t, s, ob.default_options(max_num_iterations=steps))
    print('late state cost', summ.final_cost, flush=True)
q = np.asarray(q).reshape(-1, 4); t = np.asarray(t).reshape(-1, 3)
d = dict(t=t, s=np.asarray(s), odom_c1=g.odom_c1, odom_c2=g.odom_c2, odom_w=g.odom_w, loop_c1=g.loop_c1, loop_c2=g.loop_c2)
for radius in radii:
    A, b = build_system(g, q, t, s, radius)
    A = A.tocsr()
    d['A_%g_data' % radius] = A.data; d['A_%g_indices' % radius] = A.indices; d['A_%g_indptr' % radius] = A.indptr; d['b_%g' % radius] = b
np.savez(out, **d)
print('saved', out)
