"""GPU: the C++ host side above the C-ABI (csrc/host/PoseGraphSLAM.*) — the ROS-free mirror of the reference's
PoseGraphSLAM trigger.  A scripted arrival sequence is pushed through it and the problem it builds is compared with
expectations derived by hand from the reference's rules (src/PoseGraphSLAM.cpp:1340-1367, 1550-1556, 1570-1633,
1649-1793, 1817-1849); every trigger's solve is re-done by the oracle from the same initial guess."""
import numpy as np
import pytest

from oracle import binding as ob
from solve_keyframe_pose_graph_amd import capi, graphgen
from solve_keyframe_pose_graph_amd.pose_graph_slam import PoseGraphSLAM
from tests import util
from tests.golden.make_functor_goldens import quat_to_R_np

pytestmark = pytest.mark.gpu


def T_of(q, t):
    T = np.eye(4); T[:3, :3] = quat_to_R_np(q); T[:3, 3] = t
    return T


def quat_of(T):
    return ob.mat_to_quat(T.flatten(order="F"))


def expected_odom(vio, lo, hi, worlds=None):
    """reference policy for u in [lo, hi): f = 1..5, weight 0.9^f exp(-yaw_deg^2/6), measurement w_M_u^-1 w_M_umf"""
    out = []
    for u in range(lo, hi):
        for f in range(1, 6):
            if u - f < 0:
                continue
            if worlds is not None and (worlds[u] < 0 or worlds[u - f] < 0):
                continue
            M = np.linalg.inv(vio[u]) @ vio[u - f]
            yaw = np.degrees(np.arctan2(M[1, 0], M[0, 0]))
            out.append((u, u - f, 0.9 ** f * np.exp(-yaw * yaw / 6.0), M))
    return out


@pytest.mark.parametrize("device_k0", [True, False], ids=["k0_on_device", "k0_on_host"])
def test_single_world_incremental_triggers_match_oracle(device_k0):
    g = graphgen.config("C1F5")
    vio = [T_of(g.init_q[i], g.init_t[i]) for i in range(g.n_poses)]
    S = PoseGraphSLAM(max_num_iterations=10, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    S.set_device_graph_construction(device_k0)      # steps -3-/-4- as K0 kernels (default) or on the host thread
    O = ob.OracleProblem()
    # arrival script: keyframes stream in; loop-closure messages arrive when their newer keyframe exists; trigger after each batch
    order = np.argsort(g.loop_c2, kind="stable")
    batches = [order[:7], order[7:13], order[13:]]
    n_seen, n_edges_seen, prev_solved = 0, 0, 0
    exp_edges = []
    assert not S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()       # nothing to do yet
    for bi, batch in enumerate(batches):
        upto = int(g.loop_c2[batch].max()) + 1 + 3
        upto = min(upto, g.n_poses)
        for i in range(n_seen, upto):
            S.add_node(0, vio[i].flatten(order="F"))
        n_seen = upto
        for e in batch:
            S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], float(g.loop_w[e]))   # (a = newer, b = older, b_T_a)
        su_before = S.solvedUntil()
        assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
        assert S.solvedUntil() == n_seen - 1 and S.nNodes() == n_seen
        # ---- expected residual blocks of this trigger: new loop edges first (switch id = message index), then odometry
        new_loops = [(int(g.loop_c1[e]), int(g.loop_c2[e]), float(g.loop_w[e]), n_edges_seen + k) for k, e in enumerate(batch)]
        n_edges_seen += len(batch)
        new_odom = expected_odom(vio, su_before + 1, n_seen)
        exp_edges += [(b, a, w, sw) for (b, a, w, sw) in new_loops] + [(u, v, w, -1) for (u, v, w, _) in new_odom]
        c1, c2, w, sw = S.added_edges()
        assert len(c1) == len(exp_edges)
        for k, (a_, b_, w_, s_) in enumerate(exp_edges):
            assert (c1[k], c2[k], sw[k]) == (a_, b_, s_) and abs(w[k] - w_) <= 1e-12 * max(1.0, w_)
        # ---- regulariser: node 0, weight max(1.1, ln(1 + end - start)/2), target = its current optimised pose
        node, rw, rT = S.regularizers()
        assert list(node) == [0] and abs(rw[0] - max(1.1, np.log(1 + (n_seen - 1)) / 2)) <= 1e-12
        q0, t0 = S.initial_guess()
        assert np.abs(rT[0].reshape(4, 4, order="F") - T_of(q0[0], t0[0])).max() <= 1e-12
        # ---- initial guess: solved nodes keep their optimised pose, new nodes chain odometry from the last solved one
        if bi == 0:
            for i in range(n_seen):
                assert np.abs(T_of(q0[i], t0[i]) - vio[i]).max() <= 1e-9
        else:
            last = su_before
            Tl = T_of(q0[last], t0[last])
            for i in range(last + 1, n_seen):
                want = Tl @ np.linalg.inv(vio[last]) @ vio[i]
                assert np.abs(T_of(q0[i], t0[i]) - want).max() <= 1e-9
        # ---- the same accumulated problem, same initial guess, solved by the oracle
        O.add_switchable_edges([x[0] for x in new_loops], [x[1] for x in new_loops], g.loop_T[batch], [x[2] for x in new_loops], [x[3] for x in new_loops])
        if new_odom:
            O.add_relpose_edges([x[0] for x in new_odom], [x[1] for x in new_odom], np.array([x[3].flatten(order="F") for x in new_odom]), [x[2] for x in new_odom])
        O.set_node_regularizers(node, rT, rw)
        s0 = np.array([0.99] * n_edges_seen) if bi == 0 else np.concatenate([s_prev, [0.99] * len(batch)])
        qo, to, so, sumo = O.solve(q0, t0, s0)
        summ = S.summary()
        assert summ.num_iterations == sumo.num_iterations
        assert abs(summ.final_cost - sumo.final_cost) <= 1e-6 * max(sumo.final_cost, 1e-12)
        got_t = np.array([S.getNodePose(i)[:3, 3] for i in range(n_seen)])
        assert np.abs(got_t - to.reshape(-1, 3)).max() <= 1e-5
        s_prev = np.array([S.get_loopedge_switching_variable_val(e) for e in range(n_edges_seen)])
        assert np.abs(s_prev - so).max() <= 1e-5
    S.close()


@pytest.mark.parametrize("device_k0", [True, False], ids=["k0_on_device", "k0_on_host"])
def test_kidnap_two_worlds_merge_on_first_inter_world_edge(device_k0):
    rng = np.random.default_rng(5)
    g = util.small_graph(120, 0, f=1, seed=8, turn_deg_per_keyframe=2.0)
    truth = [T_of(g.truth_q[i], g.truth_t[i]) for i in range(120)]
    # world 0 = keyframes 0..59, kidnap, world 1 = 60..119 with its own odometry frame (restarts at identity)
    vio = [truth[i] for i in range(60)] + [np.linalg.inv(truth[60]) @ truth[i] for i in range(60, 120)]
    world = [0] * 60 + [1] * 60
    S = PoseGraphSLAM(max_num_iterations=10, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    S.set_device_graph_construction(device_k0)
    for i in range(120):
        S.add_node(world[i], vio[i].flatten(order="F"))
    S.set_kidnapped(True)
    S.add_loop_edge(40, 5, (np.linalg.inv(truth[5]) @ truth[40]).flatten(order="F"))
    assert not S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()        # kidnapped: the trigger sleeps (reference :1314-1319)
    S.set_kidnapped(False)
    assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    node, rw, _ = S.regularizers()
    assert sorted(node) == [0, 60]                                             # both worlds are their own set root
    c1, c2, w, sw = S.added_edges()
    odom = [(a, b) for a, b, s in zip(c1, c2, sw) if s < 0]
    assert all(world[a] == world[b] or True for a, b in odom)
    # the reference adds odometry edges across the kidnap too when both worlds are alive
    assert (60, 59) in odom
    # first inter-world loop edge: a = 100 (world 1), b = 20 (world 0)
    bTa = np.linalg.inv(truth[20]) @ truth[100]
    S.add_loop_edge(100, 20, bTa.flatten(order="F"))
    assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    node, rw, _ = S.regularizers()
    assert list(node) == [0]                                                   # merged: only the root world keeps a regulariser
    assert abs(rw[0] - max(1.1, np.log(1 + 59) / 2)) <= 1e-12
    # after the merge every world-1 pose is expressed in world 0's frame: close to the ground truth (noise-free script)
    err = max(np.abs(S.getNodePose(i) - truth[i]).max() for i in range(120))
    assert err <= 1e-6, err
    assert S.get_loopedge_switching_variable_val(1) > 0.9
    S.close()


def test_save_as_json_round_trip(tmp_path):
    """log_optimized_poses.json with the reference's keys (src/PoseGraphSLAM.cpp:1111-1207) and a lossless CSV matrix encoding."""
    from solve_keyframe_pose_graph_amd.pose_graph_slam import read_log_optimized_poses
    g = graphgen.config("C1")
    vio = [T_of(g.init_q[i], g.init_t[i]) for i in range(g.n_poses)]
    S = PoseGraphSLAM()
    for i in range(g.n_poses):
        S.add_node(0, vio[i].flatten(order="F"))
    for e in range(g.n_loops):
        S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    assert S.saveAsJSON(tmp_path)
    d = read_log_optimized_poses(tmp_path / "log_optimized_poses.json")
    assert d["nNodes"] == g.n_poses and d["wTc_opt"].shape == (g.n_poses, 4, 4)
    for i in (0, 7, 199):
        assert np.array_equal(d["wTc_opt"][i], S.getNodePose(i))            # %.17g round-trips doubles exactly
        assert np.abs(d["w_T_c_odom"][i] - vio[i]).max() == 0.0
    assert np.array_equal(d["edge_a"], g.loop_c2) and np.array_equal(d["edge_b"], g.loop_c1)
    sw = np.array([S.get_loopedge_switching_variable_val(e) for e in range(g.n_loops)])
    assert np.array_equal(d["switching_var_after_opt"], sw)
    S.close()


def test_replay_of_a_recorded_session_from_log_posegraph_json(tmp_path):
    """a session written in the reference's log_posegraph.json format is replayed through the trigger (incremental
    solves), and the optimised trajectory written as log_optimized_poses.json equals a direct solve of the same final problem."""
    from solve_keyframe_pose_graph_amd import replay
    from solve_keyframe_pose_graph_amd.pose_graph_slam import GraphSource, read_log_optimized_poses
    g = util.small_graph(400, 60, f=1, seed=31)
    w_M = util.poses_to_matrices(g.init_q, g.init_t)
    rec = GraphSource()
    for i in range(g.n_poses):
        rec.add_node(0, w_M[i])
    for e in range(g.n_loops):
        rec.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0, "synthetic")
    (tmp_path / "in").mkdir()
    assert rec.save_posegraph_json(tmp_path / "in")
    assert replay.main([str(tmp_path / "in"), "--out", str(tmp_path / "out"), "--every", "80"]) == 0
    out = read_log_optimized_poses(tmp_path / "out" / "log_optimized_poses.json")
    assert out["nNodes"] == g.n_poses and len(out["edge_a"]) == g.n_loops
    assert np.abs(out["w_T_c_odom"].reshape(-1, 4, 4).transpose(0, 2, 1).reshape(-1, 16) - w_M).max() < 1e-12
    assert (tmp_path / "out" / "optimized.g2o").exists() and (tmp_path / "out" / "log_posegraph.json").exists()
    # the last trigger solved the complete graph: its cost is a local minimum of the same objective the oracle sees
    src = GraphSource().load_posegraph_json(tmp_path / "out")
    S, log = replay.replay(src, every=80, max_num_iterations=10)
    assert len(log) >= 3 and all(r["final_cost"] <= r["initial_cost"] for r in log)
    assert log[-1]["keyframes"] == g.n_poses and log[-1]["loop_edges"] == g.n_loops
    got = np.array([S.getNodePose(i) for i in range(g.n_poses)])
    assert np.abs(got - out["wTc_opt"]).max() < 1e-9                          # the replay is deterministic
    inl = g.loop_is_outlier == 0
    order = np.argsort(np.maximum(g.loop_c1, g.loop_c2), kind="stable")       # arrival order = switch index
    s_fin = np.array([S.get_loopedge_switching_variable_val(k) for k in range(g.n_loops)])
    assert (s_fin[np.argsort(order)][inl] > 0.5).mean() > 0.9                 # inlier loop closures stay switched on
    S.close()


def test_cpp_example_replays_a_session_like_the_python_tool(tmp_path):
    """examples/replay_session.cpp drives the C++ host classes directly (PoseGraphSLAM + GraphFormats): same session, same trajectory."""
    import subprocess
    from solve_keyframe_pose_graph_amd import _build, replay
    from solve_keyframe_pose_graph_amd.pose_graph_slam import GraphSource, read_log_optimized_poses
    exe = _build.build_examples()
    g = util.small_graph(300, 40, f=1, seed=33)
    w_M = util.poses_to_matrices(g.init_q, g.init_t)
    rec = GraphSource()
    for i in range(g.n_poses):
        rec.add_node(0, w_M[i])
    for e in range(g.n_loops):
        rec.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    (tmp_path / "in").mkdir(); (tmp_path / "cpp").mkdir()
    assert rec.save_posegraph_json(tmp_path / "in")
    r = subprocess.run([exe, str(tmp_path / "in"), str(tmp_path / "cpp"), "60"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "LM iterations" in r.stdout
    assert replay.main([str(tmp_path / "in"), "--out", str(tmp_path / "py"), "--every", "60"]) == 0
    a = read_log_optimized_poses(tmp_path / "cpp" / "log_optimized_poses.json")
    b = read_log_optimized_poses(tmp_path / "py" / "log_optimized_poses.json")
    assert a["nNodes"] == b["nNodes"] == g.n_poses
    assert np.abs(a["wTc_opt"] - b["wTc_opt"]).max() < 1e-9
    assert np.abs(a["switching_var_after_opt"] - b["switching_var_after_opt"]).max() < 1e-9


def test_readers_run_concurrently_with_the_trigger_thread():
    """The reference's threading contract: one thread runs the trigger, others read getNodePose / nNodes / solvedUntil /
    get_loopedge_switching_variable_val under mutex_opt_vars at any time; the solve itself holds no lock and writes back once."""
    import threading
    g = util.small_graph(1500, 200, f=2, seed=41)
    w_M = util.poses_to_matrices(g.init_q, g.init_t)
    S = PoseGraphSLAM(max_num_iterations=10)
    order = np.argsort(np.maximum(g.loop_c1, g.loop_c2), kind="stable")
    stop = threading.Event()
    seen = {"reads": 0, "bad": 0, "max_solved": -1}

    def reader():
        rng = np.random.default_rng(0)
        while not stop.is_set():
            n = S.nNodes()
            su = S.solvedUntil()
            seen["max_solved"] = max(seen["max_solved"], su)
            if n > 0:
                i = int(rng.integers(0, n))
                if S.nodePoseExists(i):
                    T = S.getNodePose(i)
                    ok = np.isfinite(T).all() and abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-6 and np.array_equal(T[3], [0, 0, 0, 1])
                    seen["bad"] += 0 if ok else 1
                    seen["reads"] += 1
            S.get_loopedge_switching_variable_val(0)
    th = [threading.Thread(target=reader) for _ in range(3)]
    for x in th:
        x.start()
    k = 0
    try:
        for i in range(g.n_poses):
            S.add_node(0, w_M[i])
            while k < len(order) and max(g.loop_c1[order[k]], g.loop_c2[order[k]]) <= i:
                e = order[k]
                S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
                k += 1
            if (i + 1) % 250 == 0:
                S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
        # the readers get a moment to observe the state after the last trigger (they only run between its return and stop.set())
        import time
        deadline = time.time() + 5.0
        while seen["max_solved"] < g.n_poses - 1 and time.time() < deadline:
            time.sleep(0.01)
    finally:
        stop.set()
        for x in th:
            x.join(timeout=60)
    assert seen["reads"] > 100 and seen["bad"] == 0 and seen["max_solved"] == g.n_poses - 1
    S.close()


def test_continue_on_top_of_a_saved_map_load_state(tmp_path):
    """The reference's relocalisation-in-a-previous-map flow: session A is solved and saved as solved_posegraph.json
    (Composer::saveStateToDisk); session B loads it (loadStateFromDisk + PoseGraphSLAM::load_state: the old keyframes become CONSTANT
    optimisation variables), drives on in a new world with its own odometry frame, and the first loop closures into the old map
    merge the worlds and place the new keyframes in the old map's frame while the map itself stays bit-exact."""
    from solve_keyframe_pose_graph_amd.pose_graph_slam import GraphSource
    g = util.small_graph(160, 0, f=1, seed=8, turn_deg_per_keyframe=2.0)
    truth = [T_of(g.truth_q[i], g.truth_t[i]) for i in range(160)]
    # ---- session A: keyframes 0..79 with two loop closures, solved, saved
    A = PoseGraphSLAM(max_num_iterations=20, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    for i in range(80):
        A.add_node(0, truth[i].flatten(order="F"))
    A.add_loop_edge(70, 10, (np.linalg.inv(truth[10]) @ truth[70]).flatten(order="F"))
    A.add_loop_edge(60, 25, (np.linalg.inv(truth[25]) @ truth[60]).flatten(order="F"))
    assert A.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    old_map = np.array([A.getNodePose(i) for i in range(80)])
    assert A.save_solved_posegraph_json(tmp_path)
    A.close()
    # ---- session B: load the map, continue in world 1 (odometry restarts at identity)
    src = GraphSource().load_solved_posegraph_json(tmp_path)
    assert src.n_nodes() == 80 and np.abs(src.loaded_poses().reshape(80, 4, 4).transpose(0, 2, 1) - old_map).max() < 1e-12
    B = src.attach_solver(max_num_iterations=20, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    assert B.load_state(True)
    assert B.solvedUntil() == 79 and B.nNodes() == 80
    as_loaded = np.array([B.getNodePose(i) for i in range(80)])
    assert np.abs(as_loaded - old_map).max() < 1e-12           # matrix -> (xyzw, t) -> matrix round trip of the storage
    for i in range(80, 160):
        B.add_node(1, (np.linalg.inv(truth[80]) @ truth[i]).flatten(order="F"))
    # loop closures from the new drive into the old map: a = new keyframe, b = old keyframe, b_T_a
    for a, b in ((100, 20), (130, 50), (150, 75)):
        B.add_loop_edge(a, b, (np.linalg.inv(truth[b]) @ truth[a]).flatten(order="F"))
    assert B.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    # the loaded map is constant: bit-exact
    now = np.array([B.getNodePose(i) for i in range(80)])
    assert np.array_equal(now, as_loaded)
    # no odometry residue was added among or onto the map's keyframes except the reference's u <-> u-f across the boundary
    c1, c2, w, sw = B.added_edges()
    odom = [(a, b) for a, b, s_ in zip(c1, c2, sw) if s_ < 0]
    assert all(a >= 80 for a, b in odom) and min(a for a, b in odom) == 80
    # the new keyframes land in the old map's frame (noise-free script -> close to the ground truth expressed in the map frame)
    err = max(np.abs(B.getNodePose(i) - truth[i]).max() for i in range(80, 160))
    assert err <= 1e-5, err
    node, rw, _ = B.regularizers()
    assert list(node) == [0]                      # merged into world 0's set: one regulariser, on the (constant) root keyframe
    B.close()


def _generic_se3(seed):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return T_of(q, rng.normal(size=3) * 3.0)


@pytest.mark.parametrize("device_k0", [True, False], ids=["k0_on_device", "k0_on_host"])
def test_first_trigger_when_the_first_vio_pose_is_not_identity(device_k0):
    """Real VINS sessions do not start at identity.  On the very first trigger (solvedUntil == 0) the reference gives keyframe 0 the
    guess w_M_0 (src/PoseGraphSLAM.cpp:1756-1761) and chains the rest from it: w_T_0 * (w_M_0^-1 w_M_u) = w_M_u (:1767-1775)."""
    g = graphgen.config("C1F5")
    G = _generic_se3(3)
    vio = [G @ T_of(g.init_q[i], g.init_t[i]) for i in range(g.n_poses)]
    S = PoseGraphSLAM(max_num_iterations=10, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    S.set_device_graph_construction(device_k0)
    for i in range(g.n_poses):
        S.add_node(0, vio[i].flatten(order="F"))
    for e in range(g.n_loops):
        S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    q0, t0 = S.initial_guess()
    for i in range(g.n_poses):
        assert np.abs(T_of(q0[i], t0[i]) - vio[i]).max() <= 1e-9, i
    node, rw, rT = S.regularizers()
    assert list(node) == [0] and np.abs(rT[0].reshape(4, 4, order="F") - vio[0]).max() <= 1e-9
    # gauge: the same session started at identity converges to the same trajectory moved by G
    S2 = PoseGraphSLAM(max_num_iterations=10, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    S2.set_device_graph_construction(device_k0)
    for i in range(g.n_poses):
        S2.add_node(0, T_of(g.init_q[i], g.init_t[i]).flatten(order="F"))
    for e in range(g.n_loops):
        S2.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    assert S2.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    assert abs(S.summary().final_cost - S2.summary().final_cost) <= 1e-7 * S2.summary().final_cost
    # (the LM damping is not gauge invariant, so the two 10-iteration trajectories agree to the flatness of the minimum, not to rounding)
    err = max(np.abs(S.getNodePose(i) - G @ S2.getNodePose(i)).max() for i in range(g.n_poses))
    assert err <= 1e-4, err
    S.close(); S2.close()


def test_host_and_device_construction_agree_on_not_quite_orthonormal_vio_poses():
    """VIO poses are orthonormal only to ~1e-7.  The reference inverts them with Eigen's general Matrix4d::inverse()
    (src/PoseGraphSLAM.cpp:1599, :1463): the host path, the K0 device path and numpy's general inverse must agree on such input —
    odometry records, weights, initial guesses, and the world-to-world pose at first contact."""
    rng = np.random.default_rng(12)
    g = util.small_graph(90, 0, f=1, seed=9, turn_deg_per_keyframe=2.0)
    G = _generic_se3(4)
    truth = [T_of(g.truth_q[i], g.truth_t[i]) for i in range(90)]
    def rough(T):
        T = T.copy(); T[:3, :3] += rng.normal(size=(3, 3)) * 1e-6
        return T
    vio = [rough(G @ truth[i]) for i in range(50)] + [rough(np.linalg.inv(truth[50]) @ truth[i]) for i in range(50, 90)]
    world = [0] * 50 + [1] * 40
    bTa1 = np.linalg.inv(truth[5]) @ truth[40]
    bTa2 = np.linalg.inv(truth[20]) @ truth[70]
    sessions = []
    for device_k0 in (True, False):
        S = PoseGraphSLAM(max_num_iterations=3, cg_rel_tolerance=1e-12, cg_max_iterations=20000)
        S.set_device_graph_construction(device_k0)
        for i in range(90):
            S.add_node(world[i], vio[i].flatten(order="F"))
        S.add_loop_edge(40, 5, bTa1.flatten(order="F"))
        assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
        first = (S.added_edges(), S.initial_guess())
        S.add_loop_edge(70, 20, bTa2.flatten(order="F"))               # first inter-world edge: worlds merge (:1459-1464)
        assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
        sessions.append((S, first, S.initial_guess()))
    (Sd, fd, gd), (Sh, fh, gh) = sessions
    (c1d, c2d, wd, swd), (q0d, t0d) = fd
    (c1h, c2h, wh, swh), (q0h, t0h) = fh
    assert np.array_equal(c1d, c1h) and np.array_equal(c2d, c2h) and np.array_equal(swd, swh)
    assert np.abs(np.array(wd) - np.array(wh)).max() <= 1e-13
    assert np.abs(q0d - q0h).max() <= 1e-12 and np.abs(t0d - t0h).max() <= 1e-11
    # against the general inverse: weights of the reference policy
    want = expected_odom(vio, 1, 90, world)
    got_w = [w for w, s_ in zip(wh, swh) if s_ < 0]
    assert len(got_w) == len(want) and max(abs(a - b[2]) for a, b in zip(got_w, want)) <= 1e-12
    # after the merge: world-1 keyframes are re-expressed with wb_T_wa = w_M_b * bTa * (w_M_a)^-1 from the ODOMETRY poses
    wb_T_wa = vio[20] @ bTa2 @ np.linalg.inv(vio[70])
    for (q0, t0) in (gd, gh):
        for i in (50, 60, 89):
            want_i = wb_T_wa @ vio[i]
            R = want_i[:3, :3]
            # (the world-1 keyframes were already optimised once in their own frame: equal to their odometry up to that solve's residual)
            assert np.abs(t0[i] - want_i[:3, 3]).max() <= 1e-5
            assert np.abs(T_of(q0[i], t0[i])[:3, :3] - R).max() <= 1e-5
    assert np.abs(gd[0] - gh[0]).max() <= 1e-12 and np.abs(gd[1] - gh[1]).max() <= 1e-10
    Sd.close(); Sh.close()


def test_a_bad_loop_edge_is_dropped_alone_and_a_library_error_does_not_advance_the_trigger():
    g = graphgen.config("C1")
    S = PoseGraphSLAM(max_num_iterations=5)
    for i in range(g.n_poses):
        S.add_node(0, T_of(g.init_q[i], g.init_t[i]).flatten(order="F"))
    S.add_loop_edge(30, 30, np.eye(4).flatten(order="F"))          # a == b: no residual block can hold one parameter block twice
    for e in range(5):
        S.add_loop_edge(int(g.loop_c2[e]), int(g.loop_c1[e]), g.loop_T[e], 1.0)
    assert S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once()
    c1, c2, w, sw = S.added_edges()
    loops = [(a, b, s_) for a, b, s_ in zip(c1, c2, sw) if s_ >= 0]
    assert [s_ for _, _, s_ in loops] == [1, 2, 3, 4, 5]           # switch index = message index; message 0 was dropped alone
    assert S.last_error() == 0 and S.solvedUntil() == g.n_poses - 1
    S.close()
