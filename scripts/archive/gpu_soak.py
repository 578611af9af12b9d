"""Soak: device memory across repeated solves on one handle, create/solve/destroy cycles and host-shim sessions (hipMemGetInfo)."""
import sys; sys.path.insert(0,'.')
import ctypes as C, numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from solve_keyframe_pose_graph_amd.pose_graph_slam import PoseGraphSLAM
from tests import util
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
g = util.small_graph(600, 80, f=2, seed=3); q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True)
P.solve(q, t, s)
m0 = free_mb()
for i in range(300):
    P.solve(q, t, s)
    if i % 50 == 0: P.evaluate(q, t, s); P.time_vio_odometry_kernel if False else None
m1 = free_mb()
print("repeated solves on one handle: free MB before %.1f after %.1f" % (m0, m1))
for rep in range(4):
    for i in range(60):
        Pi = util.pgo_problem(g, True); Pi.solve(q, t, s); Pi.close()
    m2 = free_mb()
    print("create/solve/destroy x60 (round %d): free MB after %.1f" % (rep, m2))
w_M = util.poses_to_matrices(g.init_q, g.init_t)
for i in range(20):
    S = PoseGraphSLAM()
    for k in range(g.n_poses): S.add_node(0, w_M[k])
    for e in range(g.n_loops): S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once(); S.close()
m3 = free_mb()
print("host shim sessions x20: free MB after %.1f" % m3)
assert abs(m1 - m0) < 8 and abs(m3 - m2) < 64
print("ok")
