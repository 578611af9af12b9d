"""ctypes binding of libpgo.so (include/pgo.h) — the drop-in boundary for the reference's Ceres path.

The library is the product; this module only marshals numpy arrays across the C-ABI.  It fails loudly when
libpgo.so cannot be built/loaded or when no HIP device is present: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _build

PGO_MAX_ITERATION_LOG = 256
PGO_COMM_ID_BYTES = 128
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Options(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int32), ("linear_solver", C.c_int32), ("jacobi_scaling", C.c_int32),
                ("max_num_consecutive_invalid_steps", C.c_int32),
                ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("cg_max_iterations", C.c_int32), ("cg_check_every", C.c_int32), ("cg_rel_tolerance", C.c_double),
                ("cg_warm_start", C.c_int32), ("cg_use_graph", C.c_int32), ("cg_early_tolerance", C.c_double), ("cg_early_reject_rho", C.c_double), ("cg_mid_tolerance", C.c_double), ("cg_mid_reject_rho", C.c_double), ("coarse_aggregates", C.c_int32), ("mg_min_keyframes", C.c_int32), ("coarse_min_radius", C.c_double),
                ("mg_omega", C.c_double), ("mg_correction_scale", C.c_double), ("mg_first_passes", C.c_int32), ("mg_passes", C.c_int32), ("mg_dense_max_nodes", C.c_int32), ("mg_switch_iterations", C.c_int32),
                ("mg_loop_discount", C.c_double), ("mg_regroup_fraction", C.c_double), ("mg_prolongation_damping", C.c_double), ("mg_smoothed_levels", C.c_int32), ("mg_min_keyframes_switchable", C.c_int32),
                ("device_id", C.c_int32), ("verbosity", C.c_int32), ("cg_single_reduction", C.c_int32), ("mg_explicit_transfer", C.c_int32), ("cg_end_game", C.c_int32), ("cg_pause_always", C.c_int32), ("mg_smoothed_fine", C.c_int32),
                ("mg_dist_min_rows", C.c_int32), ("mg_fine_filter", C.c_int32), ("mg_dist_setup", C.c_int32)]


class Iteration(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("step_is_valid", C.c_int32), ("step_is_successful", C.c_int32), ("cg_iterations", C.c_int32),
                ("cost", C.c_double), ("cost_change", C.c_double), ("model_cost_change", C.c_double), ("relative_decrease", C.c_double),
                ("gradient_max_norm", C.c_double), ("step_norm", C.c_double), ("trust_region_radius", C.c_double), ("cg_residual", C.c_double),
                ("seconds", C.c_double), ("reason", C.c_int32), ("preconditioner", C.c_int32),
                ("seconds_system", C.c_double), ("seconds_pcg", C.c_double), ("seconds_evaluate", C.c_double), ("seconds_linearize", C.c_double),
                ("cg_iterations_multigrid", C.c_int32), ("single_reduction", C.c_int32)]


# pgo_iteration.reason / .preconditioner (include/pgo.h)
STEP_ACCEPTED, STEP_REJECTED_RHO, STEP_REJECTED_AT_PAUSE, STEP_INVALID_FACTORIZATION, STEP_INVALID_BREAKDOWN, STEP_INVALID_MODEL, STEP_CONVERGED = range(7)
STEP_REASONS = ["accepted", "rejected-rho", "rejected-at-pause", "invalid-factorization", "invalid-breakdown", "invalid-model", "converged"]
PRECOND_BLOCK_JACOBI, PRECOND_TWO_LEVEL, PRECOND_MULTIGRID, PRECOND_RETRIED = 0, 1, 2, 16


class Summary(C.Structure):
    _fields_ = [("termination_type", C.c_int32), ("num_iterations", C.c_int32), ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
                ("cg_iterations", C.c_int64), ("initial_cost", C.c_double), ("final_cost", C.c_double), ("seconds_total", C.c_double),
                ("seconds_device", C.c_double), ("num_logged", C.c_int32), ("reserved_", C.c_int32),
                ("iterations", Iteration * PGO_MAX_ITERATION_LOG), ("message", C.c_char * 256), ("cg_iterations_multigrid", C.c_int64), ("pcg_retries", C.c_int32), ("reserved2_", C.c_int32)]


class ShardingStats(C.Structure):
    """pgo_sharding_stats (include/pgo.h): the rank-local handle and what its exchanges moved"""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("keyframes_local", C.c_int64), ("keyframes_owned", C.c_int64), ("keyframes_shared", C.c_int64), ("shared_global", C.c_int64),
                ("mg_levels", C.c_int32), ("mg_levels_distributed", C.c_int32), ("mg_rows_total", C.c_int64), ("mg_rows_own", C.c_int64), ("mg_blocks_total", C.c_int64), ("mg_blocks_own", C.c_int64),
                ("pcg_iterations", C.c_int64), ("exchanges", C.c_int64), ("allreduces", C.c_int64), ("bytes_sent_neighbour", C.c_double), ("bytes_allreduce", C.c_double),
                ("bytes_sent_per_mg_iteration", C.c_double), ("bytes_sent_per_bj_iteration", C.c_double), ("bytes_round5_per_mg_iteration", C.c_double), ("bytes_round5_per_bj_iteration", C.c_double),
                ("exchanges_per_mg_iteration", C.c_int32), ("exchanges_per_bj_iteration", C.c_int32),
                ("mg_setup_levels_own_rows", C.c_int32), ("mg_setup_exchanges", C.c_int32), ("mg_setup_blocks_total", C.c_int64), ("mg_setup_blocks_own", C.c_int64),
                ("bytes_sent_per_mg_setup", C.c_double), ("bytes_allreduce_replicated_setup", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/pgo.h declares (checked by tests/test_capi_symbols.py against the header text)
ABI_VERSION = 7      # PGO_ABI_VERSION of the include/pgo.h this view mirrors
EXPORTS = [
    "pgo_abi_version", "pgo_abi_sizeof", "pgo_options_init", "pgo_create", "pgo_destroy", "pgo_set_options", "pgo_reserve",
    "pgo_add_relpose_edges", "pgo_add_switchable_edges", "pgo_set_node_regularizers", "pgo_set_nodes_constant",
    "pgo_num_relpose_edges", "pgo_num_switchable_edges", "pgo_num_regularizers",
    "pgo_set_vio_poses", "pgo_num_vio_poses", "pgo_add_odometry_edges_from_vio", "pgo_initial_guess_from_vio", "pgo_get_relpose_edge_records",
    "pgo_solve", "pgo_solve_begin", "pgo_lm_step", "pgo_solve_end", "pgo_evaluate",
    "pgo_get_jacobian_blocks", "pgo_get_normal_blocks", "pgo_apply_normal_operator", "pgo_manifold_plus",
    "pgo_comm_get_unique_id", "pgo_comm_init", "pgo_comm_destroy", "pgo_comm_init_custom", "pgo_comm_set_exchange", "pgo_local_group_create", "pgo_local_group_abort", "pgo_local_group_destroy", "pgo_comm_init_local",
    "pgo_get_sharding_stats", "pgo_mg_level_norms", "pgo_partition_edges",
    "pgo_time_linearize_kernel", "pgo_time_kernel", "pgo_time_vio_odometry_kernel", "pgo_dense_spd_inverse", "pgo_device_synchronize", "pgo_strerror", "pgo_last_error", "pgo_build_info",
]

_lib = None


class PgoError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libpgo error %d: %s" % (code, text))
        self.code = code


def load(build=True):
    """Loads libpgo.so (building it in-tree with hipcc when stale).  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PGO_LIBPGO_OVERRIDE") or (_build.build_libpgo() if build else _build.LIBPGO)   # override: kernel A/B experiments (scripts/) load a variant build
    if not os.path.exists(path):
        raise RuntimeError("libpgo.so is missing (%s): the HIP library must be built; there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.pgo_strerror.restype = C.c_char_p
    lib.pgo_last_error.restype = C.c_char_p
    lib.pgo_last_error.argtypes = [C.c_void_p]
    lib.pgo_strerror.argtypes = [C.c_int]
    if hasattr(lib, "pgo_build_info"):
        lib.pgo_build_info.restype = C.c_char_p
        lib.pgo_build_info.argtypes = []
    for f in EXPORTS:
        if not hasattr(lib, f):
            raise RuntimeError("libpgo.so does not export %s" % f)
    lib.pgo_abi_version.restype = C.c_int32
    if lib.pgo_abi_version() != ABI_VERSION:
        raise RuntimeError("libpgo.so speaks ABI %d, this ctypes view %d: rebuild the library from this tree" % (lib.pgo_abi_version(), ABI_VERSION))
    lib.pgo_abi_sizeof.restype = C.c_int64
    lib.pgo_abi_sizeof.argtypes = [C.c_int32]
    for which, T in enumerate((Options, Iteration, Summary)):
        if lib.pgo_abi_sizeof(which) != C.sizeof(T):
            raise RuntimeError("capi.%s is %d bytes, libpgo.so's struct %d: the ctypes view is out of date with include/pgo.h" % (T.__name__, C.sizeof(T), lib.pgo_abi_sizeof(which)))
    _lib = lib
    return lib


def build_info():
    """(sha256 the loaded library says its sources had, sha256 of the sources in this checkout): equal when the library was built from this tree by _build.py."""
    txt = load().pgo_build_info().decode()
    return txt.rsplit(":", 1)[-1], _build.source_tree_hash()


def default_options(**kw):
    o = Options()
    load().pgo_options_init(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError("unknown option %r" % k)
        setattr(o, k, v)
    return o


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pd(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _pi(a):
    return a.ctypes.data_as(_ip) if a is not None else None


class Problem:
    """Persistent solver problem = the reference's `ceres::Problem reint_problem` + `ceres::Solve`."""

    def __init__(self, options=None, **opt_kw):
        self.lib = load()
        self.h = C.c_void_p()
        o = options if options is not None else default_options(**opt_kw)
        rc = self.lib.pgo_create(C.byref(self.h), C.byref(o))
        if rc != 0:
            self.h = C.c_void_p()
            raise PgoError(rc, self.lib.pgo_strerror(rc).decode())
        self.options = o
        self.n_rel = self.n_sw = self.n_reg = 0

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.pgo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise PgoError(rc, "%s — %s" % (self.lib.pgo_strerror(rc).decode(), self.lib.pgo_last_error(self.h).decode()))

    def set_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.options, k):
                raise TypeError("unknown option %r" % k)
            setattr(self.options, k, v)
        self._check(self.lib.pgo_set_options(self.h, C.byref(self.options)))

    # ---- problem construction ----
    def add_relpose_edges(self, c1, c2, c1_T_c2, weight):
        c1, c2, T, w = _i(c1), _i(c2), _d(c1_T_c2), _d(weight)
        n = len(c1)
        assert len(c2) == n and T.size == 16 * n and w.size == n
        self._check(self.lib.pgo_add_relpose_edges(self.h, C.c_int64(n), _pi(c1), _pi(c2), _pd(T), _pd(w)))
        self.n_rel += n

    def add_switchable_edges(self, c1, c2, c1_T_c2, weight, switch_idx):
        c1, c2, T, s = _i(c1), _i(c2), _d(c1_T_c2), _i(switch_idx)
        n = len(c1)
        w = _d(weight) if weight is not None else None
        assert len(c2) == n and T.size == 16 * n and len(s) == n
        self._check(self.lib.pgo_add_switchable_edges(self.h, C.c_int64(n), _pi(c1), _pi(c2), _pd(T), _pd(w), _pi(s)))
        self.n_sw += n

    def set_node_regularizers(self, node, target, weight):
        node, T, w = _i(node), _d(target), _d(weight)
        n = len(node)
        assert T.size == 16 * n and w.size == n
        self._check(self.lib.pgo_set_node_regularizers(self.h, C.c_int64(n), _pi(node), _pd(T), _pd(w)))
        self.n_reg = n

    def set_nodes_constant(self, node):
        node = _i(node)
        self._check(self.lib.pgo_set_nodes_constant(self.h, C.c_int64(len(node)), _pi(node)))

    # ---- graph construction on the device from raw VIO poses ----
    def set_vio_poses(self, first, w_M):
        M = _d(w_M).reshape(-1, 16)
        self._check(self.lib.pgo_set_vio_poses(self.h, C.c_int64(first), C.c_int64(len(M)), _pd(M)))

    def num_vio_poses(self):
        n = C.c_int64()
        self._check(self.lib.pgo_num_vio_poses(self.h, C.byref(n)))
        return n.value

    def add_odometry_edges_from_vio(self, node_set_id, u_begin, u_end, f_max=5, use_yaw_weight=True):
        sid = _i(node_set_id) if node_set_id is not None else None
        n = C.c_int64()
        self._check(self.lib.pgo_add_odometry_edges_from_vio(self.h, _pi(sid), C.c_int64(u_begin), C.c_int64(u_end), C.c_int32(f_max),
                                                              C.c_int32(1 if use_yaw_weight else 0), C.byref(n)))
        self.n_rel += n.value
        return n.value

    def initial_guess_from_vio(self, left, left_of_node, u_begin, u_end, quat, t):
        L = _d(left).reshape(-1, 16)
        sel = _i(left_of_node)
        assert quat.dtype == np.float64 and t.dtype == np.float64 and quat.flags.c_contiguous and t.flags.c_contiguous
        self._check(self.lib.pgo_initial_guess_from_vio(self.h, C.c_int64(len(L)), _pd(L), _pi(sel), C.c_int64(u_begin), C.c_int64(u_end), _pd(quat), _pd(t)))

    def relpose_edge_records(self, first, n):
        c1 = np.zeros(n, np.int32); c2 = np.zeros(n, np.int32); rec = np.zeros((n, 8))
        self._check(self.lib.pgo_get_relpose_edge_records(self.h, C.c_int64(first), C.c_int64(n), _pi(c1), _pi(c2), _pd(rec)))
        return c1, c2, rec

    # ---- solve ----
    @staticmethod
    def _state(quat, t, sw):
        q = np.array(quat, dtype=np.float64).reshape(-1).copy()
        tt = np.array(t, dtype=np.float64).reshape(-1).copy()
        s = np.array(sw, dtype=np.float64).reshape(-1).copy() if sw is not None else np.zeros(0)
        assert q.size % 4 == 0 and tt.size == q.size // 4 * 3
        return q, tt, s

    def solve(self, quat, t, sw=None):
        """= ceres::Solve.  Returns (quat, t, sw, Summary); the inputs are not modified."""
        q, tt, s = self._state(quat, t, sw)
        summ = Summary()
        self._check(self.lib.pgo_solve(self.h, _pd(q), _pd(tt), _pd(s) if s.size else None, C.c_int64(q.size // 4), C.c_int64(s.size), C.byref(summ)))
        return q, tt, s, summ

    def solve_begin(self, quat, t, sw=None):
        q, tt, s = self._state(quat, t, sw)
        self._shape = (q.size // 4, s.size)
        self._check(self.lib.pgo_solve_begin(self.h, _pd(q), _pd(tt), _pd(s) if s.size else None, C.c_int64(q.size // 4), C.c_int64(s.size)))

    def lm_step(self, ignore_termination=False):
        done = C.c_int32(0)
        self._check(self.lib.pgo_lm_step(self.h, C.c_int32(1 if ignore_termination else 0), C.byref(done)))
        return bool(done.value)

    def solve_end(self):
        N, S = self._shape
        q, tt, s = np.zeros(4 * N), np.zeros(3 * N), np.zeros(S)
        summ = Summary()
        self._check(self.lib.pgo_solve_end(self.h, _pd(q), _pd(tt), _pd(s) if S else None, C.byref(summ)))
        return q, tt, s, summ

    def evaluate(self, quat, t, sw=None, want_residuals=True, want_gradient=True):
        q, tt, s = self._state(quat, t, sw)
        N, S = q.size // 4, s.size
        cost = C.c_double(0)
        res = np.zeros(6 * self.n_rel + 7 * self.n_sw + 6 * self.n_reg) if want_residuals else None
        grad = np.zeros(6 * N + S) if want_gradient else None
        self._check(self.lib.pgo_evaluate(self.h, _pd(q), _pd(tt), _pd(s) if S else None, C.c_int64(N), C.c_int64(S), C.byref(cost), _pd(res), _pd(grad)))
        self._shape = (N, S)
        return cost.value, res, grad

    def jacobian_blocks(self, kind, first=0, count=None):
        n = [self.n_rel, self.n_sw, self.n_reg][kind]
        count = n - first if count is None else count
        J1 = np.zeros((count, 6, 6)); J2 = np.zeros((count, 6, 6)); ds = np.zeros((count, 7))
        self._check(self.lib.pgo_get_jacobian_blocks(self.h, C.c_int32(kind), C.c_int64(first), C.c_int64(count), _pd(J1), _pd(J2), _pd(ds)))
        return J1, J2, ds

    def normal_blocks(self):
        N, _ = self._shape
        E = self.n_rel + self.n_sw
        diag = np.zeros((N, 6, 6)); grad = np.zeros((N, 6)); off = np.zeros((E, 6, 6))
        c = np.zeros((self.n_sw, 12)); hss = np.zeros(self.n_sw); gs = np.zeros(self.n_sw)
        self._check(self.lib.pgo_get_normal_blocks(self.h, _pd(diag), _pd(grad), _pd(off), _pd(c), _pd(hss), _pd(gs)))
        return diag, grad, off, c, hss, gs

    def apply_normal_operator(self, x):
        x = _d(x).reshape(-1)
        y = np.zeros_like(x)
        self._check(self.lib.pgo_apply_normal_operator(self.h, _pd(x), _pd(y)))
        return y

    def manifold_plus(self, quat, t, delta):
        """EigenQuaternionParameterization::Plus of every keyframe on the device (parity hook): returns (quat_out, t_out)"""
        q = _d(quat).reshape(-1); tt = _d(t).reshape(-1); d = _d(delta).reshape(-1)
        n = q.size // 4
        qo = np.zeros_like(q); to = np.zeros_like(tt)
        self._check(self.lib.pgo_manifold_plus(self.h, C.c_int64(n), _pd(q), _pd(tt), _pd(d), _pd(qo), _pd(to)))
        return qo.reshape(n, 4), to.reshape(n, 3)

    def time_kernel(self, which, launches=20):
        ms = C.c_double(0); by = C.c_double(0)
        self._check(self.lib.pgo_time_kernel(self.h, C.c_int32(which), C.c_int32(launches), C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def time_vio_odometry_kernel(self, f_max=5, launches=20):
        ms = C.c_double(0); by = C.c_double(0)
        self._check(self.lib.pgo_time_vio_odometry_kernel(self.h, C.c_int32(f_max), C.c_int32(launches), C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def dense_spd_inverse(self, a, launches=1):
        """K6's blocked Gauss-Jordan inverse of a symmetric positive definite matrix; returns (inverse, average milliseconds)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = a.shape[0]
        out = np.empty_like(a); ms = C.c_double(0)
        self._check(self.lib.pgo_dense_spd_inverse(self.h, C.c_int32(n), _pd(a), _pd(out), C.c_int32(launches), C.byref(ms)))
        return out, ms.value

    def synchronize(self):
        self._check(self.lib.pgo_device_synchronize(self.h))

    # ---- multi-GPU ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * PGO_COMM_ID_BYTES)()
        rc = load().pgo_comm_get_unique_id(buf)
        if rc != 0:
            raise PgoError(rc, load().pgo_strerror(rc).decode())
        return bytes(buf)

    def comm_init(self, rank, world_size, unique_id):
        buf = (C.c_uint8 * PGO_COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self.lib.pgo_comm_init(self.h, C.c_int32(rank), C.c_int32(world_size), buf))

    def comm_init_custom(self, rank, world_size, fn):
        """fn(device_ptr:int, count:int, op:int, stream:int) -> 0 on success; all-reduces `count` doubles in device memory in place."""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)

        def tramp(ctx, buf, count, op, stream):
            try:
                return int(fn(buf, count, op, stream))
            except Exception:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._custom_cb = CB(tramp)   # keep alive
        self._world = world_size
        self._check(self.lib.pgo_comm_init_custom(self.h, C.c_int32(rank), C.c_int32(world_size), self._custom_cb, None))

    def comm_set_exchange(self, fn):
        """fn(send_ptr:int, send_off:list, recv_ptr:int, recv_off:list, stream:int) -> 0: the neighbour exchange of a caller-supplied collective (MPI_Alltoallv in doubles on device buffers)."""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.c_void_p)
        world = self._world

        def tramp(ctx, sbuf, soff, rbuf, roff, stream):
            try:
                return int(fn(sbuf, [soff[i] for i in range(world + 1)], rbuf, [roff[i] for i in range(world + 1)], stream))
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._custom_xcb = CB(tramp)
        self._check(self.lib.pgo_comm_set_exchange(self.h, self._custom_xcb))

    def comm_init_local(self, rank, world_size, group):
        """in-process communicator: `group` from local_group_create(world_size); every rank from its own thread"""
        self._check(self.lib.pgo_comm_init_local(self.h, C.c_int32(rank), C.c_int32(world_size), group))

    def sharding_stats(self):
        st = ShardingStats()
        self._check(self.lib.pgo_get_sharding_stats(self.h, C.byref(st)))
        return st

    def mg_level_norms(self, level):
        """diagnostic: sums of squares of what this rank's cycle kernels read of multigrid level `level` (1-based) after the last set-up"""
        out = np.zeros(8)
        self._check(self.lib.pgo_mg_level_norms(self.h, C.c_int32(level), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def comm_destroy(self):
        self._check(self.lib.pgo_comm_destroy(self.h))


def local_group_create(world_size):
    g = C.c_void_p()
    rc = load().pgo_local_group_create(C.c_int32(world_size), C.byref(g))
    if rc != 0:
        raise PgoError(rc, load().pgo_strerror(rc).decode())
    return g


def local_group_abort(group):
    load().pgo_local_group_abort(group)


def local_group_destroy(group):
    load().pgo_local_group_destroy(group)


PARTITION = {"contiguous": 0, "chain": 1, "spatial": 2}


def partition_edges(policy, world, positions, rel_c1, rel_c2, sw_c1, sw_c2):
    """pgo_partition_edges (host only): -> (node_part [n_nodes], rel_rank [n_rel], sw_rank [n_sw]); the policies of sharding.py behind the C-ABI."""
    lib = load()
    pos = _d(positions).reshape(-1, 3)
    rc1, rc2, sc1, sc2 = _i(rel_c1), _i(rel_c2), _i(sw_c1), _i(sw_c2)
    part = np.zeros(len(pos), np.int32); rr = np.zeros(len(rc1), np.int32); sr = np.zeros(len(sc1), np.int32)
    rc = lib.pgo_partition_edges(C.c_int32(PARTITION[policy]), C.c_int32(world), C.c_int64(len(pos)), _pd(pos), C.c_int64(len(rc1)), _pi(rc1), _pi(rc2),
                                 C.c_int64(len(sc1)), _pi(sc1), _pi(sc2), _pi(part), _pi(rr), _pi(sr))
    if rc != 0:
        raise PgoError(rc, lib.pgo_strerror(rc).decode())
    return part, rr, sr


def problem_from_graph(g, switchable=True, options=None, edge_slice=None, **opt_kw):
    """Builds a Problem from a graphgen.PoseGraph the way the reference's trigger adds residual blocks
    (odometry -> SixDOFError, loop closures -> switchable / plain, regularisers).  `edge_slice(kind, n)`
    optionally returns the index subset this rank owns (edge sharding)."""
    P = Problem(options, **opt_kw)
    sel = (lambda kind, n: np.arange(n)) if edge_slice is None else edge_slice
    io = sel("odom", g.n_odom)
    if len(io):
        P.add_relpose_edges(g.odom_c1[io], g.odom_c2[io], g.odom_T[io], g.odom_w[io])
    il = sel("loop", g.n_loops)
    if len(il):
        if switchable:
            P.add_switchable_edges(g.loop_c1[il], g.loop_c2[il], g.loop_T[il], g.loop_w[il], il.astype(np.int32))
        else:
            P.add_relpose_edges(g.loop_c1[il], g.loop_c2[il], g.loop_T[il], g.loop_w[il])
    ir = sel("reg", len(g.reg_node))
    if len(ir):
        P.set_node_regularizers(g.reg_node[ir], g.reg_T[ir], g.reg_w[ir])
    return P
