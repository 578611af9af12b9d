#!/usr/bin/env python3
"""Generates tests/golden/c3_ten_iterations.json — an INDEPENDENT CPU trajectory of the reference's 10-iteration solve on the full C3
graph (100 000 keyframes / 300 000 edges), used by the GPU suite as the full-size parity anchor.

The oracle's exact block Cholesky cannot factor the C3 normal matrix in reasonable time (fill > 2e7 blocks), so the linear systems
are solved here with scipy's conjugate gradients to a relative residual of 1e-12 — the same mathematical step as SPARSE_NORMAL_CHOLESKY
to ~1e-12 — on a matrix assembled in scipy from the ORACLE's Jet-autodiff Jacobian blocks.  The trust-region logic is the Python
restatement of oracle/pgo_oracle.cpp (Ceres trust_region_minimizer.cc / levenberg_marquardt_strategy.cc).  Nothing of libpgo is used.

Run (about 30-60 min, one core):  python tests/golden/make_c3_trajectory.py [n_iterations] [config] [output name] [preconditioner] [converge]
With a fifth argument "converge" the run does not stop at n_iterations but at Ceres' own convergence tests with the reference's options (src/PoseGraphSLAM.cpp:1268-1272 set
none of them, so the defaults hold: function_tolerance 1e-6, parameter_tolerance 1e-8) — n_iterations is then only the safety cap — and the
output (tests/golden/c3_converged.json) also carries every 100th keyframe's pose and all switches thresholded at 0.5: the converged-minimum anchor SURVEY.md 8d(ii) asks for.
The state is checkpointed after every iteration (<output>.state.npz, not committed), so an interrupted run resumes where it stopped.
The optional preconditioner "mg" (an aggregation-multigrid V-cycle in scipy, scripts/research/amg_probe.py) only shortens the CG runs of the
long trajectories (20 iterations: the trust region grows to 1e6 and block-Jacobi needs ~10^4 iterations per step); the steps are the same
to the CG tolerance of 1e-12.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402
from tests import util  # noqa: E402


def linearize(O, g, q, t, s):
    N, S = g.n_poses, g.n_loops
    J1r, J2r, _ = O.jacobian_blocks(q, t, s, 0)
    J1s, J2s, dss = O.jacobian_blocks(q, t, s, 1)
    J1p, _, _ = O.jacobian_blocks(q, t, s, 2)
    cost, res, grad = O.evaluate(q, t, s)
    Hd = np.zeros((N, 6, 6))
    np.add.at(Hd, g.odom_c1, np.einsum('eia,eib->eab', J1r, J1r)); np.add.at(Hd, g.odom_c2, np.einsum('eia,eib->eab', J2r, J2r))
    np.add.at(Hd, g.loop_c1, np.einsum('eia,eib->eab', J1s, J1s)); np.add.at(Hd, g.loop_c2, np.einsum('eia,eib->eab', J2s, J2s))
    np.add.at(Hd, g.reg_node, np.einsum('eia,eib->eab', J1p, J1p))
    Hoff_r = np.einsum('eia,eib->eab', J1r, J2r)
    Hoff_s = np.einsum('eia,eib->eab', J1s, J2s)
    c1v = np.einsum('eia,ei->ea', J1s, dss[:, :6]); c2v = np.einsum('eia,ei->ea', J2s, dss[:, :6])
    hss = (dss ** 2).sum(1)
    rs = res[6 * g.n_odom:6 * g.n_odom + 7 * S].reshape(S, 7)
    gs = np.einsum('ei,ei->e', dss, rs)
    return dict(cost=cost, grad=grad, Hd=Hd, Hoff_r=Hoff_r, Hoff_s=Hoff_s, c1=c1v, c2=c2v, hss=hss, gs=gs,
                J=(J1r, J2r, J1s, J2s, dss, J1p), r=(res[:6 * g.n_odom].reshape(-1, 6), rs, res[6 * g.n_odom + 7 * S:].reshape(-1, 6)))


def solve_step(g, L, scale_p, scale_s, diag_p, diag_s, radius, precond='bj', t_now=None):
    N, S = g.n_poses, g.n_loops
    lam_p = diag_p / (radius * scale_p ** 2)
    lam_s = diag_s / (radius * scale_s ** 2)
    a = L['hss'] + lam_s
    Hd = L['Hd'].copy()
    Hoff_s = L['Hoff_s'] - np.einsum('ea,eb->eab', L['c1'], L['c2']) / a[:, None, None]
    np.add.at(Hd, g.loop_c1, -np.einsum('ea,eb->eab', L['c1'], L['c1']) / a[:, None, None])
    np.add.at(Hd, g.loop_c2, -np.einsum('ea,eb->eab', L['c2'], L['c2']) / a[:, None, None])
    Hd[np.arange(N)[:, None], np.arange(6), np.arange(6)] += lam_p.reshape(N, 6)
    r_ = np.concatenate([np.arange(N), g.odom_c1, g.odom_c2, g.loop_c1, g.loop_c2])
    c_ = np.concatenate([np.arange(N), g.odom_c2, g.odom_c1, g.loop_c2, g.loop_c1])
    b_ = np.concatenate([Hd, L['Hoff_r'], L['Hoff_r'].transpose(0, 2, 1), Hoff_s, Hoff_s.transpose(0, 2, 1)])
    ii = (r_[:, None, None] * 6 + np.arange(6)[None, :, None]) + 0 * np.arange(6)[None, None, :]
    jj = (c_[:, None, None] * 6 + np.arange(6)[None, None, :]) + 0 * np.arange(6)[None, :, None]
    A = sp.coo_matrix((b_.ravel(), (ii.ravel(), jj.ravel())), shape=(6 * N, 6 * N)).tocsr()
    b = -L['grad'][:6 * N].copy()
    np.add.at(b.reshape(N, 6), g.loop_c1, L['c1'] * (L['gs'] / a)[:, None]); np.add.at(b.reshape(N, 6), g.loop_c2, L['c2'] * (L['gs'] / a)[:, None])
    Di = np.linalg.inv(Hd)
    M = spla.LinearOperator((6 * N, 6 * N), matvec=lambda v: np.einsum('nab,nb->na', Di, v.reshape(N, 6)).ravel())
    if precond == 'mg':
        from scripts.research import amg_probe as mg
        H = mg.Hier(A, t_now, lambda A_, N_, lvl: mg.graph_aggregates(A_, N_, 3 if lvl == 0 else 2), min_coarse=500, verbose=False)
        cyc = mg.Cycle(H, 'V', ('jac', 0.9), fine_additive=True)
        M = spla.LinearOperator((6 * N, 6 * N), matvec=lambda v: cyc(v))
    it = [0]
    x, info = spla.cg(A, b, rtol=1e-12, atol=0.0, maxiter=100000, M=M, callback=lambda xk: it.__setitem__(0, it[0] + 1))
    dp = x.reshape(N, 6)
    ds = -(L['gs'] + np.einsum('ea,ea->e', L['c1'], dp[g.loop_c1]) + np.einsum('ea,ea->e', L['c2'], dp[g.loop_c2])) / a
    return dp, ds, it[0], info


def model_change(g, L, dp, ds):
    J1r, J2r, J1s, J2s, dss, J1p = L['J']
    rr, rs, rp = L['r']
    u = np.einsum('eia,ea->ei', J1r, dp[g.odom_c1]) + np.einsum('eia,ea->ei', J2r, dp[g.odom_c2])
    mc = np.sum(u * (rr + 0.5 * u))
    u6 = np.einsum('eia,ea->ei', J1s, dp[g.loop_c1]) + np.einsum('eia,ea->ei', J2s, dp[g.loop_c2])
    u7 = np.concatenate([u6, np.zeros((len(u6), 1))], axis=1) + dss * ds[:, None]
    mc += np.sum(u7 * (rs + 0.5 * u7))
    up = np.einsum('eia,ea->ei', J1p, dp[g.reg_node])
    mc += np.sum(up * (rp + 0.5 * up))
    return -mc


def plus(q, t, s, dp, ds):
    n = np.linalg.norm(dp[:, :3], axis=1)
    sbd = np.where(n > 0, np.sin(n) / np.where(n > 0, n, 1), 1.0)
    dq = np.concatenate([dp[:, :3] * sbd[:, None], np.cos(n)[:, None]], axis=1)
    ax, ay, az, aw = dq.T; bx, by, bz, bw = q.T
    qn = np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz], axis=1)
    return qn, t + dp[:, 3:], s + ds


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    name = sys.argv[2] if len(sys.argv) > 2 else "C3"
    out_name = sys.argv[3] if len(sys.argv) > 3 else "%s_ten_iterations.json" % name.lower()
    precond = sys.argv[4] if len(sys.argv) > 4 else "bj"
    converge = len(sys.argv) > 5 and sys.argv[5] == "converge"
    function_tolerance, parameter_tolerance = 1e-6, 1e-8      # Ceres defaults
    here = os.path.dirname(os.path.abspath(__file__))
    state_path = os.path.join(here, out_name + '.state.npz')
    g = graphgen.config(name)
    O = util.oracle_problem(g, True)
    q, t, s = util.initial_state(g, True)
    N, S = g.n_poses, g.n_loops
    L = linearize(O, g, q, t, s)
    diagH = np.einsum('naa->na', L['Hd']).reshape(-1)
    scale_p = 1.0 / (1.0 + np.sqrt(diagH)); scale_s = 1.0 / (1.0 + np.sqrt(L['hss']))     # Jacobi scaling: fixed at iteration 0
    radius, decrease = 1e4, 2.0
    x_cost = L['cost']
    log = [dict(iteration=0, cost=x_cost, successful=1, radius=radius)]
    reuse = False
    first_it = 1
    diag_p = diag_s = None
    if converge and os.path.exists(state_path):      # resume an interrupted run
        st = np.load(state_path, allow_pickle=True)
        q, t, s = st['q'], st['t'], st['s']
        radius, decrease, reuse, first_it = float(st['radius']), float(st['decrease']), bool(st['reuse']), int(st['next_it'])
        log = json.loads(str(st['log']))
        L = linearize(O, g, q, t, s)
        x_cost = L['cost']
        if reuse:
            diag_p, diag_s = st['diag_p'], st['diag_s']
        print('resumed at iteration %d, cost %.12e, radius %.3e' % (first_it, x_cost, radius), flush=True)
    t00 = time.time()
    termination = "NO_CONVERGENCE: maximum number of iterations"
    for it in range(first_it, n_iter + 1):
        if not reuse:
            diag_p = np.clip(scale_p ** 2 * np.einsum('naa->na', L['Hd']).reshape(-1), 1e-6, 1e32)
            diag_s = np.clip(scale_s ** 2 * L['hss'], 1e-6, 1e32)
        t0 = time.time()
        dp, ds, cgit, info = solve_step(g, L, scale_p, scale_s, diag_p, diag_s, radius, precond, t)
        mc = model_change(g, L, dp, ds)
        qc, tc, sc = plus(q, t, s, dp, ds)
        cand = O.evaluate(qc, tc, sc, want_residuals=False, want_gradient=False)[0]
        rho = (x_cost - cand) / mc
        rec = dict(iteration=it, radius=radius, model_cost_change=mc, candidate_cost=cand, relative_decrease=rho, cg_iterations=cgit)
        if converge:
            # trust_region_minimizer.cc: the parameter-tolerance test, then the function-tolerance test, both BEFORE the step is accepted or rejected
            step_norm = np.sqrt(np.sum((qc - q) ** 2) + np.sum((tc - t) ** 2) + np.sum((sc - s) ** 2))
            x_norm = np.sqrt(np.sum(q ** 2) + np.sum(t ** 2) + np.sum(s ** 2))
            rec['step_norm'] = float(step_norm)
            stop = None
            if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
                stop = "CONVERGENCE: parameter tolerance"
            elif abs(x_cost - cand) <= function_tolerance * x_cost:
                stop = "CONVERGENCE: function tolerance"
            if stop:
                rec['successful'] = 0; rec['cost'] = x_cost; rec['terminated'] = stop
                log.append(rec)
                termination = stop
                print('it %2d cost %.12e |dcost| %.3e -> %s' % (it, x_cost, abs(x_cost - cand), stop), flush=True)
                break
        if rho > 1e-3:
            q, t, s = qc, tc, sc
            L = linearize(O, g, q, t, s)
            x_cost = L['cost']
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3)); decrease = 2.0; reuse = False
            rec['successful'] = 1
        else:
            radius /= decrease; decrease *= 2.0; reuse = True
            rec['successful'] = 0
        rec['cost'] = x_cost
        log.append(rec)
        print('it %2d cost %.12e rho %.3e cg %d (%s) %.0fs' % (it, x_cost, rho, cgit, 'ok' if rec['successful'] else 'REJ', time.time() - t0), flush=True)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), out_name + '.partial'), "w") as f:   # a long run survives an interruption
            json.dump(dict(config=name, iterations=log), f)
        if converge:
            np.savez(state_path, q=q, t=t, s=s, radius=radius, decrease=decrease, reuse=reuse, next_it=it + 1, log=json.dumps(log),
                     diag_p=diag_p, diag_s=diag_s)
    out = dict(note="generated by tests/golden/make_c3_trajectory.py: oracle Jacobians + scipy CG (rtol 1e-12, preconditioner %s) + Python restatement of the Ceres LM loop" % precond,
               config=name, n_poses=N, n_edges=g.n_odom + g.n_loops, iterations=log, seconds=time.time() - t00,
               final_t_sample=t[::997].tolist(), final_s_sample=s[::997].tolist())
    if converge:
        on = (s > 0.5).astype(np.uint8)
        out.update(termination=termination, function_tolerance=function_tolerance, parameter_tolerance=parameter_tolerance, final_cost=x_cost, final_chi2=2.0 * x_cost,
                   pose_sample_stride=100, final_q_sample_100=q[::100].tolist(), final_t_sample_100=t[::100].tolist(),
                   switches_on_hex=np.packbits(on).tobytes().hex(), n_switches=int(S), n_switches_on=int(on.sum()),
                   switch_margin=float(np.abs(s - 0.5).min()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), out_name)
    with open(path, "w") as f:
        json.dump(out, f)
    if os.path.exists(path + '.partial'):
        os.remove(path + '.partial')
    print("wrote", path)


if __name__ == "__main__":
    main()
