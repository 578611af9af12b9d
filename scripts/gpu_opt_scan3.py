"""Option scan: 20-step solves of C3 / C4 (or others) under sets of pgo_options.  python scripts/gpu_opt_scan3.py C3,C4 "mg_first_passes=2" "mg_first_passes=2,mg_passes=3" ...
(the empty string "" = library defaults; S<n> = a session-structured graph of n keyframes; iters=<k> sets the LM iteration budget)."""
import sys
sys.path.insert(0, '/root/repo')
from solve_keyframe_pose_graph_amd import graphgen
from tests import util
names = sys.argv[1].split(',')
sets = sys.argv[2:] or [""]
for name in names:
    if name.startswith('S'):      # session-structured graph of that many keyframes (f = 1..5 odometry with yaw weights, one loop closure per 5 keyframes, 2-degree turns)
        n = int(name[1:]); g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    elif name.startswith('G'):    # C3-structured graph (the default generator, f = 1,2, one loop closure per keyframe, 10 % outliers) of that many keyframes
        n = int(name[1:]); g = graphgen.generate(n, n, odom_f_max=2, seed=3)
    else: g = graphgen.config(name)
    sw = name != 'C2'
    q, t, s = util.initial_state(g, sw)
    for st in sets:
        kw = {}
        for item in (st.split(',') if st else []):
            k, x = item.split('='); kw[k] = float(x) if ('.' in x or 'e' in x) else int(x)
        iters = kw.pop('iters', 20)
        best = None
        for rep in range(2):
            P = util.pgo_problem(g, sw, max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, **kw)
            _, _, _, sm = P.solve(q, t, s); P.close()
            if best is None or sm.seconds_device < best.seconds_device: best = sm
        print('%-4s %-40s %.4f s  cg %6d (multigrid %6d)  final %.9e' % (name, st or 'defaults', best.seconds_device, best.cg_iterations, best.cg_iterations_multigrid, best.final_cost), flush=True)
