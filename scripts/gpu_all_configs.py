"""The reference's 10-iteration budget (src/PoseGraphSLAM.cpp:1272) on every BASELINE.json config with library defaults on one GPU — LM iterations/s, PCG iterations, chi^2, K1 bandwidth —
and, beside it, the CPU port (oracle/: 1 thread, exact block Cholesky) on the SAME graph where that finishes in test time (C1, C1F5, C2) or on a same-structure sample
(C3: 24 000 poses; C4: 4 worlds x 6 000 poses), with both final chi^2.  The full-size C3 number of the port is measured once per round by scripts/cpu_c3_full.py
(profiles/r04_cpu_c3_full.json).  kind = "port": context for the GPU number, never "vs Ceres"."""
import json, os, sys, time; sys.path.insert(0, '.')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util


def gpu(g, sw):
    q, t, s = util.initial_state(g, sw)
    P = util.pgo_problem(g, sw)
    P.solve(q, t, s)
    t0 = time.time()
    qq, tt, ss, summ = P.solve(q, t, s)
    wall = time.time() - t0
    P.solve_begin(q, t, s); ms, by = P.time_kernel(0, 20); P.solve_end()
    P.close()
    return summ, ms, by, (qq.reshape(-1, 4), tt.reshape(-1, 3)), wall


def cpu(g, sw, threads=1):
    from oracle import binding as ob
    O = util.oracle_problem(g, sw)
    q, t, s = util.initial_state(g, sw)
    t0 = time.time()
    qo, to, so, sm = O.solve(q, t, s, ob.default_options(num_threads=threads))
    return sm, time.time() - t0, (np.asarray(qo).reshape(-1, 4), np.asarray(to).reshape(-1, 3))


rows = [('C1', graphgen.config('C1'), True, 'same'), ('C1F5', graphgen.config('C1F5'), True, 'same'), ('C2', graphgen.config('C2'), False, 'same'),
        ('C3', graphgen.config('C3'), True, graphgen.generate(24000, 24000, odom_f_max=2, seed=3)),
        ('C4', graphgen.config('C4'), True, graphgen.generate(24000, 2400, odom_f_max=5, apply_yaw_weight=True, n_worlds=4, seed=4)),
        ('C5', graphgen.config('C5'), True, None)]
for name, g, sw, sample in rows:
    summ, ms, by, pose, wall = gpu(g, sw)
    print('%-5s N %7d E %8d | GPU: LM %2d (%d ok) in %.4f s device = %.2f it/s | PCG %6d (retried systems %d) | chi2 %.6e -> %.9e | K1 %.1f us %.0f GB/s | incl. upload + write-back %.4f s = %.2f it/s | %s' % (
        name, g.n_poses, g.n_odom + g.n_loops, summ.num_iterations, summ.num_successful_steps, summ.seconds_device, summ.num_iterations / summ.seconds_device, summ.cg_iterations, summ.pcg_retries,
        2 * summ.initial_cost, 2 * summ.final_cost, ms * 1e3, by / ms / 1e6, wall, summ.num_iterations / wall, summ.message.decode()), flush=True)
    if sample is None:
        print('      CPU port: not run at this size (C3 at full size: profiles/r04_cpu_c3_full.json)', flush=True)
        continue
    gs = g if isinstance(sample, str) else sample
    if gs is not g:
        ssum, _, _, pose, _ = gpu(gs, sw)
        print('      sample N %7d E %8d | GPU: LM %2d in %.4f s = %.2f it/s | chi2 -> %.9e' % (gs.n_poses, gs.n_odom + gs.n_loops, ssum.num_iterations, ssum.seconds_device, ssum.num_iterations / ssum.seconds_device, 2 * ssum.final_cost), flush=True)
    else:
        ssum = summ
    osum, wall, opose = cpu(gs, sw)
    dt = float(np.linalg.norm(pose[1] - opose[1], axis=1).max()); dr = float(util.rot_angle(pose[0], opose[0]).max())
    nt = min(os.cpu_count() or 1, 32)
    asum, _, _ = cpu(gs, sw, nt)
    print('      CPU port (1 thread, exact Cholesky, %s graph): LM %2d (%d ok) in %.3f s = %.3f it/s (linear solver %.3f s, Jacobians %.3f s, fill %d blocks) | chi2 -> %.9e | GPU/CPU chi2 rel diff %.2e | max pose diff %.2e m / %.2e rad | GPU %.1fx (vs %d threads for the residual blocks: %.3f s, %.1fx)' % (
        'the same' if gs is g else 'the sample', osum.num_iterations, osum.num_successful_steps, osum.seconds_total, osum.num_iterations / osum.seconds_total, osum.seconds_linear_solver, osum.seconds_jacobian,
        osum.chol_nnz_blocks, 2 * osum.final_cost, abs(ssum.final_cost - osum.final_cost) / osum.final_cost, dt, dr, osum.seconds_total / ssum.seconds_device, nt, asum.seconds_total, asum.seconds_total / ssum.seconds_device), flush=True)
