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
