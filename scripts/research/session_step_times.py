"""Session-sized graph (library defaults, 10 LM iterations): where an LM step's time goes.  The library's own host timers (verbosity 2) give, per step, the time of
building the LM system and its preconditioner (two-level method: Galerkin assembly + the dense Gauss-Jordan inverse), of the PCG and of the candidate evaluation and
the re-linearisation; steps are grouped by the preconditioner they ran with (pgo_iteration.preconditioner).  (Round 3's version fitted step seconds = a + b x iterations over
all steps: with one long block-Jacobi step and nine two-level steps of nearly equal length that fit says nothing about either.)"""
import os, re, subprocess, sys
sys.path.insert(0, '/root/repo')
import numpy as np

if len(sys.argv) > 2 and sys.argv[2] == 'child':
    from solve_keyframe_pose_graph_amd import graphgen
    from tests import util
    n = int(sys.argv[1])
    g = graphgen.generate(n, n // 5, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    for rep in range(2):
        P = util.pgo_problem(g, True, max_num_iterations=10, verbosity=2 if rep else 0)
        _, _, _, sm = P.solve(q, t, s)
        P.close()
    for k in range(1, sm.num_logged):
        it = sm.iterations[k]
        print('STEP %d %d %d %.9f' % (k, it.cg_iterations, it.preconditioner, it.seconds))
    print('TOTAL %.9f %.9f' % (sm.seconds_device, sm.iterations[0].seconds))
    sys.exit(0)

names = {0: 'block-Jacobi', 1: 'two-level', 2: 'multigrid'}
for n in [int(x) for x in sys.argv[1].split(',')]:
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), 'child'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sysms, pcgms, evalms, linms = {}, {}, {}, {}
    for ln in out.stderr.splitlines():
        m = re.search(r'it\s+(\d+) PCG: .*system \+ preconditioner ([0-9.]+) ms, PCG ([0-9.]+) ms', ln)
        if m: sysms[int(m.group(1))] = float(m.group(2)); pcgms[int(m.group(1))] = float(m.group(3))
        m = re.search(r'it\s+(\d+) candidate evaluated in ([0-9.]+) ms', ln)
        if m: evalms[int(m.group(1))] = float(m.group(2))
        m = re.search(r'it\s+(\d+) linearised in ([0-9.]+) ms', ln)
        if m: linms[int(m.group(1))] = float(m.group(2))
    steps = [ln.split() for ln in out.stdout.splitlines() if ln.startswith('STEP')]
    tot = [ln.split() for ln in out.stdout.splitlines() if ln.startswith('TOTAL')][0]
    print('%d keyframes: device total %.2f ms for %d LM steps, iteration 0 %.2f ms' % (n, float(tot[1]) * 1e3, len(steps), float(tot[2]) * 1e3))
    groups = {}
    for _, k, its, pre, sec in steps:
        groups.setdefault(int(pre) & 3, []).append((int(k), int(its), float(sec) * 1e3))
    for pre, rows in sorted(groups.items()):
        its = np.array([r[1] for r in rows], float); ks = [r[0] for r in rows]
        s_ = np.array([sysms.get(k, np.nan) for k in ks]); p_ = np.array([pcgms.get(k, np.nan) for k in ks])
        e_ = np.array([evalms.get(k, 0.0) + linms.get(k, 0.0) for k in ks]); w_ = np.array([r[2] for r in rows])
        print('    %-12s %2d steps: %4.0f PCG iterations per step; per step %.2f ms = system + preconditioner %.2f + PCG %.2f (%.1f us per iteration) + evaluation and re-linearisation %.2f'
              % (names.get(pre, str(pre)), len(rows), its.mean(), w_.mean(), np.nanmean(s_), np.nanmean(p_), 1e3 * np.nansum(p_) / max(its.sum(), 1), e_.mean()))
