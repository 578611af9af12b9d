"""ctypes view of the synthetic pose-graph generator (include/pgo_graphgen.h).  Workload tool, not solver code."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _build


class GenConfig(C.Structure):
    _fields_ = [("n_poses", C.c_int64), ("n_loops", C.c_int64), ("odom_f_max", C.c_int32), ("apply_yaw_weight", C.c_int32),
                ("n_worlds", C.c_int32), ("reserved_", C.c_int32), ("inter_world_frac", C.c_double), ("outlier_frac", C.c_double),
                ("odom_sigma_t", C.c_double), ("odom_sigma_r", C.c_double), ("loop_sigma_t", C.c_double), ("loop_sigma_r", C.c_double),
                ("box_scale", C.c_double), ("turn_deg_per_keyframe", C.c_double), ("loop_radius", C.c_double),
                ("straight_min", C.c_int32), ("straight_max", C.c_int32), ("min_loop_gap", C.c_int32), ("reserved2_", C.c_int32), ("seed", C.c_uint64)]


@dataclass
class PoseGraph:
    """What the reference's caller layer hands the solver (see include/pgo_graphgen.h)."""
    n_poses: int
    truth_q: np.ndarray
    truth_t: np.ndarray
    init_q: np.ndarray
    init_t: np.ndarray
    world: np.ndarray
    odom_c1: np.ndarray
    odom_c2: np.ndarray
    odom_T: np.ndarray
    odom_w: np.ndarray
    loop_c1: np.ndarray
    loop_c2: np.ndarray
    loop_T: np.ndarray
    loop_w: np.ndarray
    loop_is_outlier: np.ndarray
    reg_node: np.ndarray
    reg_T: np.ndarray
    reg_w: np.ndarray

    @property
    def n_odom(self):
        return len(self.odom_c1)

    @property
    def n_loops(self):
        return len(self.loop_c1)


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build_graphgen())
        for f in ("pgo_gen_num_poses", "pgo_gen_num_odom", "pgo_gen_num_loops", "pgo_gen_num_regularizers"):
            getattr(_lib, f).restype = C.c_int64
            getattr(_lib, f).argtypes = [C.c_void_p]
        _lib.pgo_gen_destroy.argtypes = [C.c_void_p]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def generate(n_poses, n_loops, odom_f_max=1, apply_yaw_weight=False, n_worlds=1, seed=1, **kw) -> PoseGraph:
    lib = _load()
    cfg = GenConfig()
    lib.pgo_gen_config_init(C.byref(cfg))
    cfg.n_poses, cfg.n_loops, cfg.odom_f_max, cfg.apply_yaw_weight, cfg.n_worlds, cfg.seed = n_poses, n_loops, odom_f_max, int(apply_yaw_weight), n_worlds, seed
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown generator option %r" % k)
        setattr(cfg, k, v)
    h = C.c_void_p()
    rc = lib.pgo_gen_create(C.byref(cfg), C.byref(h))
    if rc != 0:
        raise ValueError("pgo_gen_create failed: %d" % rc)
    try:
        N, EO, EL, ER = (lib.pgo_gen_num_poses(h), lib.pgo_gen_num_odom(h), lib.pgo_gen_num_loops(h), lib.pgo_gen_num_regularizers(h))
        d, i = np.float64, np.int32
        g = PoseGraph(N, np.zeros((N, 4), d), np.zeros((N, 3), d), np.zeros((N, 4), d), np.zeros((N, 3), d), np.zeros(N, i),
                      np.zeros(EO, i), np.zeros(EO, i), np.zeros((EO, 16), d), np.zeros(EO, d),
                      np.zeros(EL, i), np.zeros(EL, i), np.zeros((EL, 16), d), np.zeros(EL, d), np.zeros(EL, i),
                      np.zeros(ER, i), np.zeros((ER, 16), d), np.zeros(ER, d))
        lib.pgo_gen_get_poses(h, _p(g.truth_q, C.c_double), _p(g.truth_t, C.c_double), _p(g.init_q, C.c_double), _p(g.init_t, C.c_double), _p(g.world, C.c_int32))
        lib.pgo_gen_get_odom(h, _p(g.odom_c1, C.c_int32), _p(g.odom_c2, C.c_int32), _p(g.odom_T, C.c_double), _p(g.odom_w, C.c_double))
        lib.pgo_gen_get_loops(h, _p(g.loop_c1, C.c_int32), _p(g.loop_c2, C.c_int32), _p(g.loop_T, C.c_double), _p(g.loop_w, C.c_double), _p(g.loop_is_outlier, C.c_int32))
        lib.pgo_gen_get_regularizers(h, _p(g.reg_node, C.c_int32), _p(g.reg_T, C.c_double), _p(g.reg_w, C.c_double))
        return g
    finally:
        lib.pgo_gen_destroy(h)


# small graphs turn faster in a small box so that 200 keyframes already revisit places
_SMALL = dict(box_scale=1.0, turn_deg_per_keyframe=15.0, straight_min=2, straight_max=6, min_loop_gap=20, odom_sigma_r=0.002, odom_sigma_t=0.01)


# BASELINE.json configs.  C3 is the headline workload.
def config(name, seed=None):
    name = name.upper()
    if name == "C1":      # 200 poses / 199 odom + 20 switchable loops
        return generate(200, 20, odom_f_max=1, seed=seed or 1, **_SMALL)
    if name == "C1F5":    # same with the reference's f = 1..5 policy and yaw weights -> 985 odom edges
        return generate(200, 20, odom_f_max=5, apply_yaw_weight=True, seed=seed or 1, loop_radius=15.0, min_loop_gap=20, box_scale=1.0,
                        odom_sigma_r=0.002, odom_sigma_t=0.01)  # 2 deg/keyframe turns keep the yaw-weighted chain connected
    if name == "C2":      # 10k poses / 9 999 odom + 1 000 plain loop edges (no switches, no outliers)
        return generate(10000, 1000, odom_f_max=1, seed=seed or 2, outlier_frac=0.0)
    if name == "C3":      # 100k poses / 199 997 odom (f=1,2) + 100 003 switchable loops = 300 000 edges
        return generate(100000, 100003, odom_f_max=2, seed=seed or 3)
    if name == "C4":      # 4 worlds x 50k poses, f = 1..5 within worlds, 20k switchable loops (25% inter-world)
        return generate(200000, 20000, odom_f_max=5, apply_yaw_weight=True, n_worlds=4, seed=seed or 4)
    if name == "C5":      # 1M poses / 1 999 997 odom + 1 000 003 loops
        return generate(1000000, 1000003, odom_f_max=2, seed=seed or 5)
    raise KeyError(name)
