"""CPU: the synthetic workload generator — determinism, BASELINE.json config shapes, and that it derives the solver's
inputs the way the reference's trigger does (reference src/PoseGraphSLAM.cpp:1570-1633, 1550-1556, 1817-1849)."""
import numpy as np

from solve_keyframe_pose_graph_amd import graphgen
from tests.golden.make_functor_goldens import quat_to_R_np


def T_of(q, t):
    T = np.eye(4); T[:3, :3] = quat_to_R_np(q); T[:3, 3] = t
    return T


def test_deterministic():
    a, b = graphgen.config("C1"), graphgen.config("C1")
    for f in ("init_q", "init_t", "odom_T", "loop_T", "loop_c1", "odom_w"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    c = graphgen.config("C1", seed=99)
    assert not np.array_equal(a.init_t, c.init_t)


def test_config_shapes():
    g = graphgen.config("C1")
    assert (g.n_poses, g.n_odom, g.n_loops, len(g.reg_node)) == (200, 199, 20, 1)
    g = graphgen.config("C1F5")
    assert (g.n_poses, g.n_odom, g.n_loops) == (200, 985, 20)      # 199+198+197+196+195
    g = graphgen.config("C2")
    assert (g.n_poses, g.n_odom, g.n_loops) == (10000, 9999, 1000) and g.loop_is_outlier.sum() == 0
    g = graphgen.generate(5000, 5000, odom_f_max=2, seed=3)
    assert g.n_odom == 2 * 5000 - 3 and g.n_loops == 5000


def test_odometry_edges_follow_the_reference_policy():
    g = graphgen.config("C1F5")
    # c1 = u, c2 = u - f, measurement = w_M_u^-1 w_M_umf from the VIO (= initial, single world) poses, weight 0.9^f exp(-yaw^2/6)
    for e in [0, 5, 100, 500, 984]:
        u, v = g.odom_c1[e], g.odom_c2[e]
        f = u - v
        assert 1 <= f <= 5
        M = np.linalg.inv(T_of(g.init_q[u], g.init_t[u])) @ T_of(g.init_q[v], g.init_t[v])
        assert np.abs(M - g.odom_T[e].reshape(4, 4, order="F")).max() <= 1e-9
        yaw = np.degrees(np.arctan2(M[1, 0], M[0, 0]))
        assert abs(g.odom_w[e] - 0.9 ** f * np.exp(-yaw * yaw / 6.0)) <= 1e-9
    # at the initial guess every odometry residual is zero (all edges derive from one pose array)
    from oracle import binding as ob
    for e in [3, 77, 640]:
        u, v = g.odom_c1[e], g.odom_c2[e]
        r = ob.eval_relpose(g.init_q[u], g.init_t[u], g.init_q[v], g.init_t[v], g.odom_T[e], g.odom_w[e])[0]
        assert np.abs(r).max() <= 1e-9


def test_loop_edges_and_regulariser():
    g = graphgen.config("C1")
    assert np.all(g.loop_c1 < g.loop_c2)                       # c1 = older keyframe b, c2 = newer a (PoseGraphSLAM.cpp:1552-1555)
    assert np.all(g.loop_c2 - g.loop_c1 > 20)
    inl = np.where(g.loop_is_outlier == 0)[0]
    for e in inl[:5]:
        b, a = g.loop_c1[e], g.loop_c2[e]
        bTa = np.linalg.inv(T_of(g.truth_q[b], g.truth_t[b])) @ T_of(g.truth_q[a], g.truth_t[a])
        assert np.abs(bTa[:3, 3] - g.loop_T[e].reshape(4, 4, order="F")[:3, 3]).max() < 0.3   # truth + measurement noise
    assert g.reg_node[0] == 0 and abs(g.reg_w[0] - max(1.1, np.log(1 + 199) / 2)) < 1e-12
    assert np.abs(g.reg_T[0].reshape(4, 4, order="F") - T_of(g.init_q[0], g.init_t[0])).max() < 1e-12


def test_multi_world_merge():
    g = graphgen.generate(2000, 300, odom_f_max=5, apply_yaw_weight=True, n_worlds=4, seed=4, loop_radius=6.0)
    assert set(np.unique(g.world)) == {0, 1, 2, 3}
    # no odometry edge crosses a kidnap; each world's VIO restarts at identity before merging
    assert np.all(g.world[g.odom_c1] == g.world[g.odom_c2])
    inter = g.world[g.loop_c1] != g.world[g.loop_c2]
    assert inter.sum() > 0
    assert len(g.reg_node) >= 1 and g.reg_node[0] == 0
