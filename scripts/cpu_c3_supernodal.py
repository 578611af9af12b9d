#!/usr/bin/env python3
"""A credible CPU row: ONE Levenberg-Marquardt iteration of the benchmark graph on the host with a SUPERNODAL sparse direct solver — scipy's SuperLU
(supernodal, BLAS-3 panels, symmetric mode, MMD ordering on A + A^T) on the Schur-reduced normal matrix assembled from the oracle's Jet Jacobians — instead of the oracle
port's own up-looking block Cholesky (oracle/sparse_chol.hpp: 1 856 s for the same system).  Not Ceres + CHOLMOD (neither is in this image, CMakeLists.txt:22-23 of the
reference asks for both), but the same class of method: a fill-reducing ordering and a supernodal factorisation; LU instead of Cholesky costs it about a factor two.

One LM iteration as Ceres runs it: residuals + Jacobians, normal equations, factorise + solve, candidate cost.  Single thread (OMP / BLAS threads pinned to 1).

  python scripts/cpu_c3_supernodal.py [config | n_poses] [output.json]"""
import json
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(v, "1")

import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402
from tests import util  # noqa: E402
import make_c3_trajectory as M  # noqa: E402  (the independent CPU trajectory's linearisation and LM algebra)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "C3"
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    if what.isdigit():
        n = int(what)
        g = graphgen.generate(n, int(round(n * 100003 / 100000)), odom_f_max=2, seed=3)
        name = "C3-structured %d poses" % n
    else:
        g = graphgen.config(what); name = what
    N, S = g.n_poses, g.n_loops
    O = util.oracle_problem(g, True)
    q, t, s = util.initial_state(g, True)
    t0 = time.perf_counter()
    L = M.linearize(O, g, q, t, s)
    t_lin = time.perf_counter() - t0
    diagH = np.einsum('naa->na', L['Hd']).reshape(-1)
    scale_p = 1.0 / (1.0 + np.sqrt(diagH)); scale_s = 1.0 / (1.0 + np.sqrt(L['hss']))
    diag_p = np.clip(scale_p ** 2 * diagH, 1e-6, 1e32); diag_s = np.clip(scale_s ** 2 * L['hss'], 1e-6, 1e32)
    radius = 1e4
    # ---- the damped, Schur-reduced normal matrix (the switch columns eliminated per edge: exactly a Cholesky that orders them first)
    t0 = time.perf_counter()
    lam_p = diag_p / (radius * scale_p ** 2); lam_s = diag_s / (radius * scale_s ** 2)
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
    A = sp.coo_matrix((b_.ravel(), (ii.ravel(), jj.ravel())), shape=(6 * N, 6 * N)).tocsc()
    b = -L['grad'][:6 * N].copy()
    np.add.at(b.reshape(N, 6), g.loop_c1, L['c1'] * (L['gs'] / a)[:, None]); np.add.at(b.reshape(N, 6), g.loop_c2, L['c2'] * (L['gs'] / a)[:, None])
    t_asm = time.perf_counter() - t0
    # ---- supernodal factorisation + solve
    t0 = time.perf_counter()
    lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    t_fac = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = lu.solve(b)
    t_sol = time.perf_counter() - t0
    res = float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    fill = int(lu.L.nnz + lu.U.nnz)
    dp = x.reshape(N, 6)
    ds = -(L['gs'] + np.einsum('ea,ea->e', L['c1'], dp[g.loop_c1]) + np.einsum('ea,ea->e', L['c2'], dp[g.loop_c2])) / a
    t0 = time.perf_counter()
    mc = M.model_change(g, L, dp, ds)
    qc, tc, sc = M.plus(q, t, s, dp, ds)
    cand = O.evaluate(qc, tc, sc, want_residuals=False, want_gradient=False)[0]
    t_eval = time.perf_counter() - t0
    rho = (L['cost'] - cand) / mc
    total = t_lin + t_asm + t_fac + t_sol + t_eval
    rec = {"what": "one LM iteration on the host, 1 thread: oracle Jet Jacobians + scipy SuperLU (supernodal, SymmetricMode, MMD_AT_PLUS_A) on the Schur-reduced normal matrix",
           "workload": "%s: %d poses / %d edges" % (name, N, g.n_odom + g.n_loops), "unknowns": 6 * N, "nnz_A": int(A.nnz), "nnz_L_plus_U": fill,
           "seconds": {"jacobians_and_gradient": t_lin, "normal_matrix_assembly_numpy": t_asm, "factorisation": t_fac, "triangular_solves": t_sol, "model_change_and_candidate_cost": t_eval, "total": total},
           "lm_iters_per_s": 1.0 / total, "linear_residual_rel": res, "cost": L['cost'], "candidate_cost": cand, "relative_decrease": rho,
           "cores": 1, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"}
    txt = json.dumps(rec)
    print(txt)
    if out_path:
        with open(out_path, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
