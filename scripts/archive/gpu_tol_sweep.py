import sys, time; sys.path.insert(0,'.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util
name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
g = graphgen.config(name)
q,t,s = util.initial_state(g, True)
res = {}
for tol in [float(x) for x in sys.argv[2:]] or [1e-10, 1e-4, 1e-2, 1e-1]:
    P = util.pgo_problem(g, True, cg_rel_tolerance=tol, cg_max_iterations=6000, cg_check_every=25, max_num_iterations=150, function_tolerance=1e-8)
    t0=time.time(); qq,tt,ss,summ = P.solve(q,t,s); dt=time.time()-t0
    its=[summ.iterations[k] for k in range(summ.num_logged)]
    print('tol',tol,'LM',summ.num_iterations,'succ',summ.num_successful_steps,'cg',summ.cg_iterations,'time %.3f'%dt,'dev %.3f'%summ.seconds_device,'cost %.10e'%summ.final_cost, summ.message.decode())
    print('   costs', ' '.join('%.6e'%i.cost for i in its[:40]))
    print('   cg   ', ' '.join('%d'%i.cg_iterations for i in its[:60]))
    res[tol]=(qq,tt,ss,summ.final_cost)
    P.close()
ks=list(res)
for k in ks[1:]:
    print('vs',ks[0],k,'dcost rel',abs(res[k][3]-res[ks[0]][3])/res[ks[0]][3],'dt',np.abs(res[k][1]-res[ks[0]][1]).max(),'ds',np.abs(res[k][2]-res[ks[0]][2]).max())
