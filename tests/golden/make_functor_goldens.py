#!/usr/bin/env python3
"""Generates tests/golden/functor_goldens.json — the pin for the oracle and the HIP kernels.

The reference (mpkuse/solve_keyframe_pose_graph) has NO tests, fixtures or golden vectors for its solver
path and cannot be built in this image (Ceres/Eigen/ROS absent) — parity is unpinned by the reference
itself.  These vectors are therefore derived INDEPENDENTLY of both the oracle's
Jet arithmetic and the kernels' analytic Jacobians:

  * residuals: closed-form SE(3) algebra of what reference src/CeresResidues.h:32-69,104-127,158-201
    compute, evaluated with 50 significant digits (mpmath);
  * Jacobians: 50-digit central differences through the Ceres `EigenQuaternionParameterization::Plus`
    retraction (q <- [sin|d| d/|d|, cos|d|] (x) q), step 1e-20 -> truncation error ~1e-40.

Only the Eigen rotation-matrix -> quaternion branch rule (sign convention of q_obs / of the regulariser's
delta_q) is shared knowledge; it is pinned separately by known answers at 180-degree rotations.

Run:  python tests/golden/make_functor_goldens.py      (deterministic; numpy seed 20260928)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
H = mp.mpf(10) ** -20


def qmul(a, b):  # Hamilton product, coefficients x,y,z,w
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz]


def qconj(q):
    return [-q[0], -q[1], -q[2], q[3]]


def rotmat(q):
    """Polynomial rotation matrix of Eigen::toRotationMatrix (no normalisation)."""
    x, y, z, w = q
    return mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mat_to_quat_eigen(R):
    """Eigen's Quaternion(Matrix3) branch rule; returns x,y,z,w."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = [mp.mpf(0)] * 4
    if t > 0:
        t = mp.sqrt(t + 1)
        q[3] = t / 2
        t = mp.mpf(1) / (2 * t)
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = mp.sqrt(R[i, i] - R[j, j] - R[k, k] + 1)
        q[i] = t / 2
        t = mp.mpf(1) / (2 * t)
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def plus(q, d):
    n = mp.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    if n == 0:
        return list(q)
    s = mp.sin(n) / n
    return qmul([s * d[0], s * d[1], s * d[2], mp.cos(n)], q)


def mpv(a):
    return [mp.mpf(float(x)) for x in a]


def T_to_obs(T16):
    T = mp.matrix(4, 4)
    for c in range(4):
        for r in range(4):
            T[r, c] = mp.mpf(float(T16[c * 4 + r]))
    R = T[0:3, 0:3]
    return mat_to_quat_eigen(R), [T[0, 3], T[1, 3], T[2, 3]], T


def res_relpose(q1, t1, q2, t2, qo, to, w):
    R1 = rotmat(q1)
    R2 = rotmat(q2)
    v = mp.matrix(t1) + R1 * mp.matrix(to) - mp.matrix(t2)
    dt = R2.T * v
    dq = qmul(qmul(qconj(q2), q1), qo)
    return [w * dt[0], w * dt[1], w * dt[2], w * 2 * dq[0], w * 2 * dq[1], w * 2 * dq[2]]


def res_switch(q1, t1, q2, t2, s, qo, to):
    r6 = res_relpose(q1, t1, q2, t2, qo, to, mp.mpf(1))
    return [s * x for x in r6] + [s * (1 - s)]


def res_prior(q1, t1, T, w):
    # delta = f^-1 * [R(q1) p1; 0 1]; general inverse as the reference does (CeresResidues.h:117)
    n = mp.eye(4)
    n[0:3, 0:3] = rotmat(q1)
    for i in range(3):
        n[i, 3] = t1[i]
    d = mp.inverse(T) * n
    dq = mat_to_quat_eigen(d[0:3, 0:3])
    return [w * d[0, 3], w * d[1, 3], w * d[2, 3], w * 2 * dq[0], w * 2 * dq[1], w * 2 * dq[2]]


def num_jac(f, nres, blocks):
    """blocks: list of (kind, getter/setter index) handled by caller through f(perturbation dict)."""
    out = {}
    for name, dim in blocks:
        J = [[None] * dim for _ in range(nres)]
        for c in range(dim):
            dp = [mp.mpf(0)] * dim
            dm = [mp.mpf(0)] * dim
            dp[c] = H
            dm[c] = -H
            rp = f({name: dp})
            rm = f({name: dm})
            for i in range(nres):
                J[i][c] = (rp[i] - rm[i]) / (2 * H)
        out[name] = J
    return out


def fl(x):
    if isinstance(x, list):
        return [fl(y) for y in x]
    return float(x)


def rand_unit_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def quat_to_R_np(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def make_T(q, t):
    T = np.eye(4)
    T[:3, :3] = quat_to_R_np(q)
    T[:3, 3] = t
    return T.flatten(order="F")  # column-major, as Eigen stores Matrix4d


def axis_angle_quat(axis, ang):
    a = np.asarray(axis, float)
    a = a / np.linalg.norm(a)
    return np.concatenate([a * np.sin(ang / 2), [np.cos(ang / 2)]])


def case_relpose(q1, t1, q2, t2, T16, w):
    qo, to, _ = T_to_obs(T16)
    Q1, T1, Q2, T2, W = mpv(q1), mpv(t1), mpv(q2), mpv(t2), mp.mpf(float(w))

    def f(p):
        a1 = p.get("th1", [0, 0, 0]); b1 = p.get("p1", [0, 0, 0]); a2 = p.get("th2", [0, 0, 0]); b2 = p.get("p2", [0, 0, 0])
        return res_relpose(plus(Q1, a1), [T1[i] + b1[i] for i in range(3)], plus(Q2, a2), [T2[i] + b2[i] for i in range(3)], qo, to, W)
    r = f({})
    J = num_jac(f, 6, [("th1", 3), ("p1", 3), ("th2", 3), ("p2", 3)])
    J1 = [J["th1"][i] + J["p1"][i] for i in range(6)]
    J2 = [J["th2"][i] + J["p2"][i] for i in range(6)]
    return dict(kind="relpose", q1=fl(list(q1)), t1=fl(list(t1)), q2=fl(list(q2)), t2=fl(list(t2)), T=fl(list(T16)), w=float(w),
                q_obs=fl(qo), r=fl(r), J1=fl(J1), J2=fl(J2))


def case_switch(q1, t1, q2, t2, s, T16, w):
    qo, to, _ = T_to_obs(T16)
    Q1, T1, Q2, T2, S0 = mpv(q1), mpv(t1), mpv(q2), mpv(t2), mp.mpf(float(s))

    def f(p):
        a1 = p.get("th1", [0, 0, 0]); b1 = p.get("p1", [0, 0, 0]); a2 = p.get("th2", [0, 0, 0]); b2 = p.get("p2", [0, 0, 0]); ds = p.get("s", [0])
        return res_switch(plus(Q1, a1), [T1[i] + b1[i] for i in range(3)], plus(Q2, a2), [T2[i] + b2[i] for i in range(3)], S0 + ds[0], qo, to)
    r = f({})
    J = num_jac(f, 7, [("th1", 3), ("p1", 3), ("th2", 3), ("p2", 3), ("s", 1)])
    J1 = [J["th1"][i] + J["p1"][i] for i in range(7)]
    J2 = [J["th2"][i] + J["p2"][i] for i in range(7)]
    Js = [J["s"][i][0] for i in range(7)]
    return dict(kind="switch", q1=fl(list(q1)), t1=fl(list(t1)), q2=fl(list(q2)), t2=fl(list(t2)), s=float(s), T=fl(list(T16)), w=float(w),
                q_obs=fl(qo), r=fl(r), J1=fl(J1), J2=fl(J2), Js=fl(Js))


def case_prior(q1, t1, T16, w):
    _, _, T = T_to_obs(T16)
    Q1, T1, W = mpv(q1), mpv(t1), mp.mpf(float(w))

    def f(p):
        a1 = p.get("th1", [0, 0, 0]); b1 = p.get("p1", [0, 0, 0])
        return res_prior(plus(Q1, a1), [T1[i] + b1[i] for i in range(3)], T, W)
    r = f({})
    J = num_jac(f, 6, [("th1", 3), ("p1", 3)])
    J1 = [J["th1"][i] + J["p1"][i] for i in range(6)]
    return dict(kind="prior", q1=fl(list(q1)), t1=fl(list(t1)), T=fl(list(T16)), w=float(w), r=fl(r), J1=fl(J1))


def main():
    rng = np.random.default_rng(20260928)
    cases = []
    # ---- random cases
    for _ in range(80):
        q1, q2, qo = rand_unit_quat(rng), rand_unit_quat(rng), rand_unit_quat(rng)
        t1, t2, to = rng.normal(size=3) * 5, rng.normal(size=3) * 5, rng.normal(size=3) * 2
        cases.append(case_relpose(q1, t1, q2, t2, make_T(qo, to), rng.uniform(0.05, 1.5)))
    for _ in range(80):
        q1, q2, qo = rand_unit_quat(rng), rand_unit_quat(rng), rand_unit_quat(rng)
        t1, t2, to = rng.normal(size=3) * 5, rng.normal(size=3) * 5, rng.normal(size=3) * 2
        cases.append(case_switch(q1, t1, q2, t2, rng.uniform(-0.2, 1.4), make_T(qo, to), rng.uniform(0.05, 1.5)))
    for _ in range(30):
        # regulariser: pose near its target (the reference sets target = current pose, PoseGraphSLAM.cpp:1844)
        qf, tf = rand_unit_quat(rng), rng.normal(size=3) * 5
        dq = axis_angle_quat(rng.normal(size=3), rng.uniform(0, 0.6))
        q1 = np.array(fl(qmul(list(map(mp.mpf, map(float, dq))), list(map(mp.mpf, map(float, qf))))))
        q1 = q1 / np.linalg.norm(q1)
        if rng.uniform() < 0.5:
            q1 = -q1  # antipodal representation of the same rotation
        cases.append(case_prior(q1, tf + rng.normal(size=3) * 0.3, make_T(qf, tf), rng.uniform(1.1, 6.0)))
    for _ in range(10):  # far from target: exercises the non-trace branches
        qf, tf = rand_unit_quat(rng), rng.normal(size=3) * 5
        cases.append(case_prior(rand_unit_quat(rng), rng.normal(size=3) * 5, make_T(qf, tf), rng.uniform(1.1, 6.0)))
    # ---- adversarial: observation rotations hitting every Eigen matrix->quat branch
    ident = np.array([0, 0, 0, 1.0])
    special_obs = [ident,
                   axis_angle_quat([1, 0, 0], np.pi / 2), axis_angle_quat([0, 1, 0], np.pi / 2), axis_angle_quat([0, 0, 1], np.pi / 2),
                   axis_angle_quat([1, 0, 0], np.pi), axis_angle_quat([0, 1, 0], np.pi), axis_angle_quat([0, 0, 1], np.pi),
                   axis_angle_quat([1, 1, 0], np.pi), axis_angle_quat([0, 1, 1], np.pi), axis_angle_quat([1, 0, 1], np.pi),
                   axis_angle_quat([1, 2, 3], 2.5), axis_angle_quat([3, 2, 1], 3.0), axis_angle_quat([1, -3, 2], 2.2),
                   axis_angle_quat([1, 1, 1], 2 * np.pi / 3)]
    for qo in special_obs:
        q1, q2 = rand_unit_quat(rng), rand_unit_quat(rng)
        t1, t2, to = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        cases.append(case_relpose(q1, t1, q2, t2, make_T(qo, to), 0.9))
        cases.append(case_relpose(-q1, t1, q2, t2, make_T(qo, to), 0.81))      # antipodal q1
        cases.append(case_relpose(ident, np.zeros(3), ident, np.zeros(3), make_T(qo, to), 1.0))  # identity poses
    # noise-free ground truth -> residual exactly ~0 (known answer)
    for _ in range(6):
        q1, q2 = rand_unit_quat(rng), rand_unit_quat(rng)
        t1, t2 = rng.normal(size=3) * 3, rng.normal(size=3) * 3
        R1 = quat_to_R_np(q1)
        R2 = quat_to_R_np(q2)
        T = np.eye(4)
        T[:3, :3] = R1.T @ R2
        T[:3, 3] = R1.T @ (t2 - t1)
        cases.append(case_relpose(q1, t1, q2, t2, T.flatten(order="F"), 0.9))
        cases.append(case_switch(q1, t1, q2, t2, 0.99, T.flatten(order="F"), 1.0))
    # switch values named in SURVEY.md §8c
    for s in [0.0, 0.5, 0.99, 1.0, 1.3]:
        q1, q2, qo = rand_unit_quat(rng), rand_unit_quat(rng), rand_unit_quat(rng)
        t1, t2, to = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        cases.append(case_switch(q1, t1, q2, t2, s, make_T(qo, to), 1.0))
    # regulariser exactly at its target (the state at the start of every reference solve)
    for _ in range(4):
        qf, tf = rand_unit_quat(rng), rng.normal(size=3) * 5
        cases.append(case_prior(qf, tf, make_T(qf, tf), 5.4099))
        cases.append(case_prior(-qf, tf, make_T(qf, tf), 1.1))

    # ---- Plus and its Jacobian
    plus_cases = []
    for _ in range(12):
        q = rand_unit_quat(rng)
        d = rng.normal(size=3) * rng.choice([1e-9, 1e-3, 0.3, 1.5])
        Q, D = mpv(q), mpv(d)
        out = plus(Q, D)
        jac = [[None] * 3 for _ in range(4)]
        for c in range(3):
            dp = [mp.mpf(0)] * 3; dm = [mp.mpf(0)] * 3
            dp[c] = H; dm[c] = -H
            a, b = plus(Q, dp), plus(Q, dm)
            for i in range(4):
                jac[i][c] = (a[i] - b[i]) / (2 * H)
        plus_cases.append(dict(q=fl(list(q)), delta=fl(list(d)), q_plus=fl(out), jac0=fl(jac)))
    plus_cases.append(dict(q=fl(list(ident)), delta=[0.0, 0.0, 0.0], q_plus=fl(list(ident)), jac0=[[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [0, 0, 0]]))

    # ---- known answers for the Eigen matrix->quaternion branch rule at 180-degree rotations
    m2q = []
    for name, R, q in [("rot180_x", np.diag([1.0, -1, -1]), [1.0, 0, 0, 0]),
                       ("rot180_y", np.diag([-1.0, 1, -1]), [0, 1.0, 0, 0]),
                       ("rot180_z", np.diag([-1.0, -1, 1]), [0, 0, 1.0, 0]),
                       ("identity", np.eye(3), [0, 0, 0, 1.0]),
                       ("rot90_z", np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]]), [0, 0, np.sqrt(0.5), np.sqrt(0.5)]),
                       ("rot120_111", np.array([[0.0, 0, 1], [1, 0, 0], [0, 1, 0]]), [0.5, 0.5, 0.5, 0.5])]:
        T = np.eye(4)
        T[:3, :3] = R
        m2q.append(dict(name=name, T=fl(list(T.flatten(order="F"))), q=fl(q)))

    out = dict(note="generated by tests/golden/make_functor_goldens.py (mpmath, 50 digits); see its docstring",
               cases=cases, plus=plus_cases, mat_to_quat=m2q)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "functor_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, "cases:", len(cases), "bytes:", os.path.getsize(path))


if __name__ == "__main__":
    main()
