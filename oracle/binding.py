"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libpgo_oracle.so (the CPU restatement of the reference's Ceres path; see
pgo_oracle.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED by the reference (no reference tests/fixtures exist); pinned by tests/golden/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libpgo_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force=False):
    """Compiles the oracle with g++ (Makefile in this directory)."""
    srcs = [os.path.join(_HERE, f) for f in ("pgo_oracle.cpp", "functors.hpp", "sparse_chol.hpp")]
    if (not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs if os.path.exists(s))):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


class IterLog(C.Structure):
    _fields_ = [("iteration", C.c_int), ("step_is_valid", C.c_int), ("step_is_successful", C.c_int), ("reserved", C.c_int),
                ("cost", C.c_double), ("cost_change", C.c_double), ("model_cost_change", C.c_double), ("relative_decrease", C.c_double),
                ("gradient_max_norm", C.c_double), ("step_norm", C.c_double), ("trust_region_radius", C.c_double), ("seconds", C.c_double)]


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int), ("num_iterations", C.c_int), ("num_successful_steps", C.c_int), ("num_unsuccessful_steps", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("seconds_total", C.c_double), ("seconds_linear_solver", C.c_double),
                ("seconds_jacobian", C.c_double), ("chol_nnz_blocks", C.c_longlong), ("num_logged", C.c_int), ("reserved", C.c_int),
                ("iterations", IterLog * 256), ("message", C.c_char * 256)]


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("jacobi_scaling", C.c_int), ("max_num_consecutive_invalid_steps", C.c_int),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double), ("verbosity", C.c_int), ("num_threads", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_create.restype = C.c_void_p
        _lib.orc_destroy.argtypes = [C.c_void_p]
        _lib.orc_last_cholesky_flops.restype = C.c_double
        _lib.orc_cholesky_symbolic.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        _lib.orc_sizeof_summary.restype = C.c_size_t
        _lib.orc_sizeof_options.restype = C.c_size_t
        _lib.orc_odometry_edges_from_vio.restype = C.c_int64
        assert _lib.orc_sizeof_summary() == C.sizeof(Summary), (_lib.orc_sizeof_summary(), C.sizeof(Summary))
        assert _lib.orc_sizeof_options() == C.sizeof(Options), (_lib.orc_sizeof_options(), C.sizeof(Options))
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def default_options(**kw):
    o = Options()
    lib().orc_options_init(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ---- single block evaluation ----
def eval_relpose(q1, t1, q2, t2, T16, w):
    r = np.zeros(6); Jamb = np.zeros((6, 14)); J1 = np.zeros((6, 6)); J2 = np.zeros((6, 6))
    a = [_d(x) for x in (q1, t1, q2, t2, T16)]
    lib().orc_eval_relpose(a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], C.c_double(w), _d(r)[1], Jamb.ctypes.data_as(_dp), J1.ctypes.data_as(_dp), J2.ctypes.data_as(_dp))
    return r, Jamb, J1, J2


def eval_switch(q1, t1, q2, t2, s, T16, w):
    r = np.zeros(7); Jamb = np.zeros((7, 15)); J1 = np.zeros((7, 6)); J2 = np.zeros((7, 6)); Js = np.zeros(7)
    a = [_d(x) for x in (q1, t1, q2, t2, [s], T16)]
    lib().orc_eval_switch(a[0][1], a[1][1], a[2][1], a[3][1], a[4][1], a[5][1], C.c_double(w), r.ctypes.data_as(_dp), Jamb.ctypes.data_as(_dp),
                          J1.ctypes.data_as(_dp), J2.ctypes.data_as(_dp), Js.ctypes.data_as(_dp))
    return r, Jamb, J1, J2, Js


def eval_prior(q1, t1, T16, w):
    r = np.zeros(6); Jamb = np.zeros((6, 7)); J1 = np.zeros((6, 6))
    a = [_d(x) for x in (q1, t1, T16)]
    lib().orc_eval_prior(a[0][1], a[1][1], a[2][1], C.c_double(w), r.ctypes.data_as(_dp), Jamb.ctypes.data_as(_dp), J1.ctypes.data_as(_dp))
    return r, Jamb, J1


def mat_to_quat(T16):
    q = np.zeros(4)
    lib().orc_mat_to_quat(_d(T16)[1], q.ctypes.data_as(_dp))
    return q


def quat_plus(q, d):
    o = np.zeros(4)
    lib().orc_quat_plus(_d(q)[1], _d(d)[1], o.ctypes.data_as(_dp))
    return o


def quat_plus_jacobian(q):
    o = np.zeros((4, 3))
    lib().orc_quat_plus_jacobian(_d(q)[1], o.ctypes.data_as(_dp))
    return o


def odometry_edges_from_vio(w_M, set_id, u_begin, u_end, f_max=5, use_yaw=True):
    """The reference's odometry-residue loop (src/PoseGraphSLAM.cpp:1570-1639) -> (c1, c2, T[n,16], weight)."""
    w_M = np.ascontiguousarray(w_M, dtype=np.float64).reshape(-1, 16)
    cap = max(0, (u_end - u_begin) * f_max)
    c1 = np.zeros(cap, np.int32); c2 = np.zeros(cap, np.int32); T = np.zeros((cap, 16)); w = np.zeros(cap)
    sid = None if set_id is None else np.ascontiguousarray(set_id, dtype=np.int32)
    n = lib().orc_odometry_edges_from_vio(C.c_int64(len(w_M)), w_M.ctypes.data_as(_dp), None if sid is None else sid.ctypes.data_as(_ip),
                                          C.c_int64(u_begin), C.c_int64(u_end), C.c_int(f_max), C.c_int(1 if use_yaw else 0),
                                          c1.ctypes.data_as(_ip), c2.ctypes.data_as(_ip), T.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
    return c1[:n].copy(), c2[:n].copy(), T[:n].copy(), w[:n].copy()


def initial_guess_from_vio(left, left_of_node, w_M, u_begin, u_end, quat, t):
    """Initial guesses (src/PoseGraphSLAM.cpp:1770-1786): quat/t (full arrays) updated in place for u in [u_begin, u_end)."""
    left = np.ascontiguousarray(left, dtype=np.float64).reshape(-1, 16)
    w_M = np.ascontiguousarray(w_M, dtype=np.float64).reshape(-1, 16)
    sel = np.ascontiguousarray(left_of_node, dtype=np.int32)
    rc = lib().orc_initial_guess_from_vio(C.c_int64(len(left)), left.ctypes.data_as(_dp), sel.ctypes.data_as(_ip), w_M.ctypes.data_as(_dp),
                                          C.c_int64(u_begin), C.c_int64(u_end), quat.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
    assert rc == 0
    return quat, t


class OracleProblem:
    """Mirror of the pgo_* problem API on the CPU oracle."""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_create())
        self.n_rel = 0
        self.n_sw = 0
        self.n_pri = 0

    def __del__(self):
        try:
            if self.h:
                lib().orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def add_relpose_edges(self, c1, c2, T, w):
        c1a, c1p = _i(c1); c2a, c2p = _i(c2); Ta, Tp = _d(T); wa, wp = _d(w)
        n = len(c1a)
        assert Ta.size == 16 * n and wa.size == n
        lib().orc_add_relpose_edges(self.h, C.c_int64(n), c1p, c2p, Tp, wp)
        self.n_rel += n

    def add_switchable_edges(self, c1, c2, T, w, sw):
        c1a, c1p = _i(c1); c2a, c2p = _i(c2); Ta, Tp = _d(T); swa, swp = _i(sw)
        n = len(c1a)
        wa, wp = _d(w if w is not None else np.ones(n))
        lib().orc_add_switchable_edges(self.h, C.c_int64(n), c1p, c2p, Tp, wp, swp)
        self.n_sw += n

    def set_node_regularizers(self, node, T, w):
        na, np_ = _i(node); Ta, Tp = _d(T); wa, wp = _d(w)
        lib().orc_set_node_regularizers(self.h, C.c_int64(len(na)), np_, Tp, wp)
        self.n_pri = len(na)

    def set_nodes_constant(self, node):
        na, np_ = _i(node)
        lib().orc_set_nodes_constant(self.h, C.c_int64(len(na)), np_)

    def cholesky_symbolic(self, n_nodes):
        """(fill in 6x6 blocks, flops) of the exact block Cholesky of this problem's Schur-reduced normal matrix under the port's AMD ordering — symbolic phase only, nothing factorised"""
        nnz = C.c_longlong(0); fl = C.c_double(0)
        rc = lib().orc_cholesky_symbolic(self.h, C.c_int64(n_nodes), C.byref(nnz), C.byref(fl))
        assert rc == 0
        return int(nnz.value), float(fl.value)

    def evaluate(self, q, t, s, want_residuals=True, want_gradient=True):
        qa, qp = _d(q); ta, tp = _d(t); sa, sp = _d(s)
        N = qa.size // 4; S = sa.size
        cost = C.c_double(0)
        res = np.zeros(6 * self.n_rel + 7 * self.n_sw + 6 * self.n_pri) if want_residuals else None
        grad = np.zeros(6 * N + S) if want_gradient else None
        lib().orc_evaluate(self.h, qp, tp, sp, C.c_int64(N), C.c_int64(S), C.byref(cost),
                           res.ctypes.data_as(_dp) if res is not None else None, grad.ctypes.data_as(_dp) if grad is not None else None)
        return cost.value, res, grad

    def jacobian_blocks(self, q, t, s, kind):
        qa, qp = _d(q); ta, tp = _d(t); sa, sp = _d(s)
        N = qa.size // 4; S = sa.size
        n = [self.n_rel, self.n_sw, self.n_pri][kind]
        J1 = np.zeros((n, 6, 6)); J2 = np.zeros((n, 6, 6)); ds = np.zeros((n, 7))
        lib().orc_jacobian_blocks(self.h, qp, tp, sp, C.c_int64(N), C.c_int64(S), C.c_int(kind), J1.ctypes.data_as(_dp), J2.ctypes.data_as(_dp), ds.ctypes.data_as(_dp))
        return J1, J2, ds

    def dense_normal_matrix(self, q, t, s):
        qa, qp = _d(q); ta, tp = _d(t); sa, sp = _d(s)
        N = qa.size // 4; S = sa.size
        n = 6 * N + S
        H = np.zeros((n, n))
        lib().orc_dense_normal_matrix(self.h, qp, tp, sp, C.c_int64(N), C.c_int64(S), H.ctypes.data_as(_dp))
        return H

    def solve(self, q, t, s, options=None):
        """Returns (q, t, s, Summary) — inputs are not modified."""
        q = np.array(q, dtype=np.float64).reshape(-1).copy(); t = np.array(t, dtype=np.float64).reshape(-1).copy(); s = np.array(s, dtype=np.float64).reshape(-1).copy()
        N = q.size // 4; S = s.size
        o = options if options is not None else default_options()
        summ = Summary()
        lib().orc_solve(self.h, C.byref(o), q.ctypes.data_as(_dp), t.ctypes.data_as(_dp), s.ctypes.data_as(_dp), C.c_int64(N), C.c_int64(S), C.byref(summ))
        return q, t, s, summ
