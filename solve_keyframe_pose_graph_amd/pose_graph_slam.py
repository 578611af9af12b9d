"""ctypes view of the C++ host side (csrc/host/PoseGraphSLAM.{hpp,cpp}) — the ROS-free mirror of the reference's
`PoseGraphSLAM` solver-facing interface and of one wake-up of reinit_ceres_problem_onnewloopedge_optimize6DOF()
(reference src/PoseGraphSLAM.cpp:1287-1940).  Used by the parity tests and examples; production callers use the C++ class."""
import ctypes as C

import numpy as np

from . import _build, capi

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None


def _load():
    global _lib
    if _lib is None:
        capi.load()
        _lib = C.CDLL(_build.build_host())
        _lib.pgo_host_create.restype = C.c_void_p
        _lib.pgo_host_create_source_only.restype = C.c_void_p
        _lib.pgo_host_switch.restype = C.c_double
        for f in ("destroy", "add_node", "add_loop_edge", "set_kidnapped", "trigger", "n_nodes", "solved_until", "node_pose_exists", "get_node_pose", "switch",
                  "n_added_edges", "get_added_edges", "n_regularizers", "get_regularizers", "get_initial_guess", "get_summary", "last_error"):
            fn = getattr(_lib, "pgo_host_" + f)
            fn.argtypes = None
    return _lib


class GraphSource:
    """The data-source half of a host session (NodeDataManager + Worlds stand-in) without a solver: needs no GPU.  Reads and writes
    the reference's `log_posegraph.json` (NodeDataManager::saveAsJSON / loadFromJSON, reference src/NodeDataManager.cpp:503-754) and
    exports .g2o; `attach_solver()` turns it into a PoseGraphSLAM session over the same data."""

    def __init__(self):
        self.lib = _load()
        self.h = C.c_void_p(self.lib.pgo_host_create_source_only())
        self._owned = True

    def close(self):
        if self._owned and self.h and self.h.value:
            self.lib.pgo_host_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_node(self, world, w_M_i_colmajor16, stamp=-1.0):
        a = np.ascontiguousarray(w_M_i_colmajor16, dtype=np.float64)
        self.lib.pgo_host_add_node_stamped(self.h, C.c_int(world), a.ctypes.data_as(_dp), C.c_double(stamp))

    def add_loop_edge(self, a, b, b_T_a_colmajor16, weight=1.0, description=""):
        T = np.ascontiguousarray(b_T_a_colmajor16, dtype=np.float64)
        self.lib.pgo_host_add_loop_edge_described(self.h, C.c_int(a), C.c_int(b), T.ctypes.data_as(_dp), C.c_double(weight), description.encode())

    def n_nodes(self):
        return self.lib.pgo_host_source_n_nodes(self.h)

    def n_edges(self):
        return self.lib.pgo_host_source_n_edges(self.h)

    def node(self, i):
        w = C.c_int(); st = C.c_double(); T = np.zeros(16)
        self.lib.pgo_host_source_get_node(self.h, C.c_int(i), C.byref(w), C.byref(st), T.ctypes.data_as(_dp))
        return w.value, st.value, T

    def edge(self, e):
        a = C.c_int(); b = C.c_int(); w = C.c_double(); T = np.zeros(16); buf = C.create_string_buffer(512)
        self.lib.pgo_host_source_get_edge(self.h, C.c_int(e), C.byref(a), C.byref(b), C.byref(w), T.ctypes.data_as(_dp), buf, C.c_int(512))
        return a.value, b.value, w.value, T, buf.value.decode()

    def save_posegraph_json(self, base_path):
        return bool(self.lib.pgo_host_save_posegraph_json(self.h, str(base_path).encode()))

    def load_posegraph_json(self, base_path, edge_mask=None):
        m = np.ascontiguousarray(edge_mask, dtype=np.uint8) if edge_mask is not None else np.zeros(0, np.uint8)
        err = C.create_string_buffer(512)
        ok = self.lib.pgo_host_load_posegraph_json(self.h, str(base_path).encode(), m.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(len(m)), err, C.c_int(512))
        if not ok:
            raise ValueError("log_posegraph.json: " + err.value.decode())
        return self

    def export_g2o(self, path, optimized=False, f_max=5):
        return bool(self.lib.pgo_host_export_g2o(self.h, str(path).encode(), C.c_int(1 if optimized else 0), C.c_int(f_max)))

    # ---- solved_posegraph.json: the saved state of a finished session (Composer::saveStateToDisk / loadStateFromDisk) ----
    def save_solved_posegraph_json(self, base_path):
        return bool(self.lib.pgo_host_save_solved_posegraph_json(self.h, str(base_path).encode()))

    def load_solved_posegraph_json(self, base_path):
        err = C.create_string_buffer(512)
        if not self.lib.pgo_host_load_solved_posegraph_json(self.h, str(base_path).encode(), err, C.c_int(512)):
            raise ValueError("solved_posegraph.json: " + err.value.decode())
        return self

    def loaded_poses(self):
        """corrected poses w_T_c of the loaded map, n x 16 column-major (the keyframes a caller then marks constant)"""
        n = self.lib.pgo_host_n_loaded_poses(self.h)
        out = np.zeros((n, 16))
        for i in range(n):
            self.lib.pgo_host_get_loaded_pose(self.h, C.c_int(i), out[i].ctypes.data_as(_dp))
        return out

    def set_id_of_world(self, w):
        return self.lib.pgo_host_source_set_id_of_world(self.h, C.c_int(w))

    def pose_between_worlds(self, m, n):
        T = np.zeros(16)
        self.lib.pgo_host_source_pose_between_worlds(self.h, C.c_int(m), C.c_int(n), T.ctypes.data_as(_dp))
        return T

    def merge_worlds(self, m, n, m_T_n_colmajor16):
        T = np.ascontiguousarray(m_T_n_colmajor16, dtype=np.float64)
        self.lib.pgo_host_source_merge_worlds(self.h, C.c_int(m), C.c_int(n), T.ctypes.data_as(_dp))

    def attach_solver(self, **opt_kw):
        """-> PoseGraphSLAM over this source (needs a GPU).  The returned object owns the session."""
        opt = capi.default_options(**opt_kw)
        if not self.lib.pgo_host_attach_solver(self.h, C.byref(opt)):
            raise capi.PgoError(-2, "PoseGraphSLAM: pgo_create failed (no GPU / libpgo missing): there is no CPU fallback")
        S = PoseGraphSLAM.__new__(PoseGraphSLAM)
        S.lib, S.opt, S.h = self.lib, opt, self.h
        self._owned = False
        return S


class PoseGraphSLAM:
    def __init__(self, **opt_kw):
        self.lib = _load()
        self.opt = capi.default_options(**opt_kw)
        self.h = C.c_void_p(self.lib.pgo_host_create(C.byref(self.opt)))
        if not self.h.value:
            raise capi.PgoError(-2, "PoseGraphSLAM: pgo_create failed (no GPU / libpgo missing): there is no CPU fallback")

    def close(self):
        if self.h and self.h.value:
            self.lib.pgo_host_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the caller side (NodeDataManager stand-in) ----
    def add_node(self, world, w_M_i_colmajor16, stamp=-1.0):
        a = np.ascontiguousarray(w_M_i_colmajor16, dtype=np.float64)
        self.lib.pgo_host_add_node_stamped(self.h, C.c_int(world), a.ctypes.data_as(_dp), C.c_double(stamp))

    def add_loop_edge(self, a, b, b_T_a_colmajor16, weight=1.0, description=""):
        T = np.ascontiguousarray(b_T_a_colmajor16, dtype=np.float64)
        self.lib.pgo_host_add_loop_edge_described(self.h, C.c_int(a), C.c_int(b), T.ctypes.data_as(_dp), C.c_double(weight), description.encode())

    def set_kidnapped(self, k):
        self.lib.pgo_host_set_kidnapped(self.h, C.c_int(1 if k else 0))

    def set_device_graph_construction(self, on):
        """Steps -3-/-4- of the trigger (odometry measurements, yaw weights, VIO-derived guesses) as device kernels (default) or on the host."""
        self.lib.pgo_host_set_device_graph_construction(self.h, C.c_int(1 if on else 0))

    # ---- PoseGraphSLAM interface ----
    def load_state(self, optimization_variable_as_constants=True):
        """PoseGraphSLAM::load_state: the keyframes already in the data source are a previously solved map (kept constant)."""
        return bool(self.lib.pgo_host_load_state(self.h, C.c_int(1 if optimization_variable_as_constants else 0)))

    def reinit_ceres_problem_onnewloopedge_optimize6DOF_once(self):
        return bool(self.lib.pgo_host_trigger(self.h))

    def nNodes(self):
        return self.lib.pgo_host_n_nodes(self.h)

    def solvedUntil(self):
        return self.lib.pgo_host_solved_until(self.h)

    def nodePoseExists(self, i):
        return bool(self.lib.pgo_host_node_pose_exists(self.h, C.c_int(i)))

    def getNodePose(self, i):
        T = np.zeros(16)
        self.lib.pgo_host_get_node_pose(self.h, C.c_int(i), T.ctypes.data_as(_dp))
        return T.reshape(4, 4, order="F")

    def get_loopedge_switching_variable_val(self, e):
        return self.lib.pgo_host_switch(self.h, C.c_int(e))

    # ---- introspection ----
    def added_edges(self):
        n = self.lib.pgo_host_n_added_edges(self.h)
        c1 = np.zeros(n, np.int32); c2 = np.zeros(n, np.int32); w = np.zeros(n); sw = np.zeros(n, np.int32)
        if n:
            self.lib.pgo_host_get_added_edges(self.h, c1.ctypes.data_as(_ip), c2.ctypes.data_as(_ip), w.ctypes.data_as(_dp), sw.ctypes.data_as(_ip))
        return c1, c2, w, sw

    def regularizers(self):
        n = self.lib.pgo_host_n_regularizers(self.h)
        node = np.zeros(n, np.int32); w = np.zeros(n); T = np.zeros((n, 16))
        if n:
            self.lib.pgo_host_get_regularizers(self.h, node.ctypes.data_as(_ip), w.ctypes.data_as(_dp), T.ctypes.data_as(_dp))
        return node, w, T

    def initial_guess(self):
        n = self.nNodes()
        q = np.zeros((n, 4)); t = np.zeros((n, 3))
        self.lib.pgo_host_get_initial_guess(self.h, q.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
        return q, t

    def summary(self):
        s = capi.Summary()
        self.lib.pgo_host_get_summary(self.h, C.byref(s))
        return s

    def saveAsJSON(self, base_path):
        return bool(self.lib.pgo_host_save_as_json(self.h, str(base_path).encode()))

    def save_posegraph_json(self, base_path):
        return bool(self.lib.pgo_host_save_posegraph_json(self.h, str(base_path).encode()))

    def save_solved_posegraph_json(self, base_path):
        """Composer::saveStateToDisk's solved_posegraph.json with the optimised poses of this session"""
        return bool(self.lib.pgo_host_save_solved_posegraph_json(self.h, str(base_path).encode()))

    def export_g2o(self, path, optimized=True, f_max=5):
        return bool(self.lib.pgo_host_export_g2o(self.h, str(path).encode(), C.c_int(1 if optimized else 0), C.c_int(f_max)))

    def last_error(self):
        return self.lib.pgo_host_last_error(self.h)


def read_log_optimized_poses(path):
    """Reader for the reference's `log_optimized_poses.json` (PoseGraphSLAM::saveAsJSON, reference src/PoseGraphSLAM.cpp:1111-1207;
    matrices are Eigen CSVFormat strings "r0c0,r0c1,..;r1c0,.." — RawFileIO.h:90-101).  Returns a dict of numpy arrays."""
    import json

    def mat(sv):
        return np.array([[float(x) for x in row.split(",")] for row in sv.split(";")])
    with open(path) as f:
        d = json.load(f)
    nodes = sorted(d.get("PoseGraphSLAM_nodes", []), key=lambda x: x["node_i"])
    edges = sorted(d.get("PoseGraphSLAM_loopedgeinfo", []), key=lambda x: x["getEdge_i"])
    return {
        "nNodes": d["meta_data"]["nNodes"],
        "wTc_opt": np.array([mat(n["wTc_opt"]) for n in nodes]).reshape(-1, 4, 4),
        "w_T_c_odom": np.array([mat(n["w_T_c_odom"]) for n in nodes]).reshape(-1, 4, 4),
        "edge_a": np.array([e["a"] for e in edges], dtype=np.int32), "edge_b": np.array([e["b"] for e in edges], dtype=np.int32),
        "edge_weight": np.array([e["weight"] for e in edges]),
        "edge_world_of_a": np.array([e["world_of_a"] for e in edges], dtype=np.int32), "edge_world_of_b": np.array([e["world_of_b"] for e in edges], dtype=np.int32),
        "switching_var_after_opt": np.array([e.get("switching_var_after_opt", np.nan) for e in edges]),
    }
