"""CPU: the per-lane algebra of the K1 kernels (csrc/pgo_device_math.hpp — closed-form residuals and analytic
tangent-space Jacobian blocks) instantiated on the host by tests/native/device_math_host.cpp and checked against the
golden vectors.  This is host logic coverage; the product never evaluates edges on the CPU."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(HERE, "native", "libdevice_math_host.so")
    src = os.path.join(HERE, "native", "device_math_host.cpp")
    hdr = os.path.join(ROOT, "solve_keyframe_pose_graph_amd", "csrc", "pgo_device_math.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.dirname(hdr), "-o", so, src])
    return C.CDLL(so)


def A(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def P(a):
    return a.ctypes.data_as(dp)


def test_analytic_blocks_match_goldens(shim):
    with open(os.path.join(HERE, "golden", "functor_goldens.json")) as f:
        g = json.load(f)
    for c in g["cases"]:
        k = c["kind"]
        if k == "relpose":
            r, J1, J2 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
            a = [A(c[x]) for x in ("q1", "t1", "q2", "t2", "T")]
            shim.dm_relpose(P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), C.c_double(c["w"]), P(r), P(J1), P(J2))
            got, want = [r, J1, J2], [c["r"], c["J1"], c["J2"]]
            r2 = np.zeros(6)
            shim.dm_relpose_cost_only(P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), C.c_double(c["w"]), P(r2))
            assert np.array_equal(r, r2)
        elif k == "switch":
            r, J1, J2, Js = np.zeros(7), np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(7)
            a = [A(c[x]) for x in ("q1", "t1", "q2", "t2", "T")]
            shim.dm_switch(P(a[0]), P(a[1]), P(a[2]), P(a[3]), C.c_double(c["s"]), P(a[4]), P(r), P(J1), P(J2), P(Js))
            G1, G2 = np.array(c["J1"]), np.array(c["J2"])
            assert np.abs(G1[6]).max() == 0 and np.abs(G2[6]).max() == 0   # the 7th Jacobian row w.r.t. poses is identically zero
            got, want = [r, J1, J2, Js], [c["r"], G1[:6], G2[:6], c["Js"]]
        else:
            r, J1 = np.zeros(6), np.zeros((6, 6))
            a = [A(c[x]) for x in ("q1", "t1", "T")]
            shim.dm_prior(P(a[0]), P(a[1]), P(a[2]), C.c_double(c["w"]), P(r), P(J1))
            got, want = [r, J1], [c["r"], c["J1"]]
        for x, y in zip(got, want):
            y = np.array(y)
            assert np.abs(x - y).max() <= 2e-13 * max(1.0, np.abs(y).max()), k
    for p in g["plus"]:
        o = np.zeros(4)
        shim.dm_plus(P(A(p["q"])), P(A(p["delta"])), P(o))
        assert np.abs(o - p["q_plus"]).max() <= 1e-15
    for m in g["mat_to_quat"]:
        o = np.zeros(4)
        shim.dm_mat_to_quat(P(A(m["T"])), P(o))
        assert np.abs(o - m["q"]).max() <= 1e-15


def test_matrix_free_edge_operator_equals_jtj(shim):
    """compact_apply (the per-edge-side body of the matrix-free PCG matvec) == J_side^T (J1 p1 + J2 p2), incl. the switch Schur term."""
    rng = np.random.default_rng(7)
    for trial in range(60):
        q1, q2, qo = (rng.normal(size=4) for _ in range(3))
        q1 /= np.linalg.norm(q1); q2 /= np.linalg.norm(q2); qo /= np.linalg.norm(qo)
        t1, t2, to = rng.normal(size=3) * 3, rng.normal(size=3) * 3, rng.normal(size=3)
        from tests.golden.make_functor_goldens import make_T
        T = A(make_T(qo, to))
        p1, p2 = rng.normal(size=6), rng.normal(size=6)
        is_sw = trial % 2
        if not is_sw:
            w = rng.uniform(0.1, 1.5)
            r, J1, J2 = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
            shim.dm_relpose(P(A(q1)), P(A(t1)), P(A(q2)), P(A(t2)), P(T), C.c_double(w), P(r), P(J1), P(J2))
            u = J1 @ p1 + J2 @ p2
            want1, want2 = J1.T @ u, J2.T @ u
            ws, kscale = w, 0.0
        else:
            s = rng.uniform(0.05, 1.2)
            r, J1, J2, Js = np.zeros(7), np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(7)
            shim.dm_switch(P(A(q1)), P(A(t1)), P(A(q2)), P(A(t2)), C.c_double(s), P(T), P(r), P(J1), P(J2), P(Js))
            a_inv = 1.0 / (Js @ Js + rng.uniform(0, 0.1))
            u = J1 @ p1 + J2 @ p2
            u = u - Js[:6] * (Js[:6] @ u) * a_inv          # Schur complement of the switch column
            want1, want2 = J1.T @ u, J2.T @ u
            ws, kscale = s, np.sqrt(a_inv)
        y1, y2 = np.zeros(6), np.zeros(6)
        shim.dm_compact_apply(P(A(q1)), P(A(t1)), P(A(q2)), P(A(t2)), P(T), C.c_double(ws), C.c_int(is_sw), C.c_double(kscale), P(A(p1)), P(A(p2)), P(y1), P(y2))
        sc = max(1.0, np.abs(want1).max(), np.abs(want2).max())
        assert np.abs(y1 - want1).max() <= 1e-12 * sc and np.abs(y2 - want2).max() <= 1e-12 * sc
