"""GPU: the two-level preconditioner (block-Jacobi + rigid-body modes of keyframe aggregates; pgo_options.coarse_aggregates).  A
preconditioner does not change what the PCG converges to, so the checks are: same LM trajectory as with plain block-Jacobi and as
the oracle's exact solve, far fewer PCG iterations where it pays, and the once-per-solve comparison dropping it where it does not."""
import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util

pytestmark = pytest.mark.gpu


def run(g, switchable=True, **kw):
    q, t, s = util.initial_state(g, switchable)
    P = util.pgo_problem(g, switchable, **kw)
    out = P.solve(q, t, s)
    P.close()
    return out


@pytest.mark.parametrize("name,switchable", [("C1", True), ("C1F5", True), ("C2", False)])
def test_same_trajectory_far_fewer_iterations_on_the_small_configs(name, switchable):
    g = graphgen.config(name)
    # (mg_min_keyframes = 0 on both sides: since round 6 C2's 10 000 keyframes would take the multigrid, and the comparison would be of a run with itself)
    _, t0, s0, off = run(g, switchable, coarse_aggregates=0, cg_max_iterations=200000, mg_min_keyframes=0)
    _, t1, s1, on = run(g, switchable, mg_min_keyframes=0)
    assert on.num_iterations == off.num_iterations
    for k in range(off.num_logged):
        a, b = off.iterations[k], on.iterations[k]
        assert a.step_is_successful == b.step_is_successful
        assert abs(a.cost - b.cost) <= 1e-7 * max(a.cost, 1e-12), (k, a.cost, b.cost)
    assert np.abs(t1 - t0).max() <= 1e-5
    assert on.cg_iterations * 5 < off.cg_iterations, (on.cg_iterations, off.cg_iterations)


def test_mid_size_graph_matches_the_oracle_with_the_coarse_space_on_at_every_radius():
    """8k keyframes, every step accepted, the trust region grows to 2e8: plain block-Jacobi needs > 10^5 iterations, two levels a few 10^3;
    per-iteration costs against the oracle's exact Cholesky."""
    g = graphgen.generate(8000, 5000, odom_f_max=2, seed=21, outlier_frac=0.1)
    q, t, s = util.initial_state(g, True)
    qo, to, so, sumo = util.oracle_problem(g, True).solve(q, t, s)
    _, tp, sp, sump = run(g, True, mg_min_keyframes=0)      # (the two-level method: since round 6 the multigrid would take this graph)
    assert [sump.iterations[k].step_is_successful for k in range(sump.num_logged)] == [sumo.iterations[k].step_is_successful for k in range(sumo.num_logged)]
    for k in range(sumo.num_logged):
        assert abs(sumo.iterations[k].cost - sump.iterations[k].cost) <= 1e-6 * sumo.iterations[k].cost, k
    assert sump.cg_iterations < 12000
    assert np.abs(sp - so).max() <= 1e-3


def test_large_aggregates_switch_on_only_at_large_radius_and_the_headline_graph_is_untouched(c3=None):
    """100k keyframes / 512 aggregates = 196 keyframes each (> 64): the coarse space waits for radius >= coarse_min_radius, which the
    10-iteration C3 trajectory never reaches — identical iteration counts with and without it."""
    g = graphgen.config("C3")
    # (mg_min_keyframes = 0: at this size the multigrid would replace the two-level method altogether and the comparison would be of a run with itself)
    _, _, _, off = run(g, True, coarse_aggregates=0, mg_min_keyframes=0)
    _, _, _, on = run(g, True, mg_min_keyframes=0)
    assert [on.iterations[k].cg_iterations for k in range(on.num_logged)] == [off.iterations[k].cg_iterations for k in range(off.num_logged)]
    assert on.final_cost == off.final_cost
    # forced on from the start (coarse_min_radius = 0) it still converges to the same trajectory
    _, _, _, forced = run(g, True, coarse_min_radius=0.0, mg_min_keyframes=0)
    assert [forced.iterations[k].step_is_successful for k in range(forced.num_logged)] == [off.iterations[k].step_is_successful for k in range(off.num_logged)]
    assert abs(forced.final_cost - off.final_cost) <= 1e-6 * off.final_cost


def test_dropped_where_it_does_not_pay():
    """A chain that the reference's yaw weights cut into hundreds of loose pieces (15 degrees per keyframe: odometry weights ~1e-17 across
    every turn): the aggregates' rigid-body modes are not the slow modes there and the coarse space costs iterations.  The once-per-solve
    comparison notices at the first step and the rest of the solve runs plain block-Jacobi: same trajectory, bounded overhead."""
    g = graphgen.generate(2000, 400, odom_f_max=5, apply_yaw_weight=1, seed=5, **graphgen._SMALL)
    _, t0, _, off = run(g, True, coarse_aggregates=0)
    _, t1, _, on = run(g, True)
    assert [on.iterations[k].step_is_successful for k in range(on.num_logged)] == [off.iterations[k].step_is_successful for k in range(off.num_logged)]
    assert abs(on.final_cost - off.final_cost) <= 1e-7 * off.final_cost
    its_off = [off.iterations[k].cg_iterations for k in range(1, off.num_logged)]
    its_on = [on.iterations[k].cg_iterations for k in range(1, on.num_logged)]
    # once the comparison has run (at the first step solved to full accuracy with the coarse space) the solve is plain block-Jacobi
    assert all(abs(a - b) <= 0.01 * b + 2 for a, b in zip(its_on[-4:], its_off[-4:])) or on.cg_iterations < off.cg_iterations
    assert on.cg_iterations <= 1.35 * off.cg_iterations, (its_on, its_off)


def test_constant_and_unreferenced_keyframes_with_the_coarse_space():
    g = util.small_graph(700, 90, f=2, seed=37)
    q, t, s = util.initial_state(g, True)
    q = np.vstack([q, [[0.0, 0.0, 0.0, 1.0]] * 3]); t = np.vstack([t, np.arange(9.0).reshape(3, 3)])     # three keyframes nobody refers to
    const = [0, 1, 2, 350, 351, 699]
    from oracle import binding as ob
    O = util.oracle_problem(g, True); O.set_nodes_constant(const)
    qo, to, so, sumo = O.solve(q, t, s, ob.default_options(max_num_iterations=30, function_tolerance=1e-10))
    P = util.pgo_problem(g, True, max_num_iterations=30, function_tolerance=1e-10)
    P.set_nodes_constant(const)
    qp, tp, sp, sump = P.solve(q, t, s)
    P.close()
    assert np.array_equal(tp.reshape(-1, 3)[const], t[const]) and np.array_equal(tp.reshape(-1, 3)[-3:], t[-3:])
    assert abs(sump.final_cost - sumo.final_cost) <= 1e-6 * sumo.final_cost
    assert sump.num_iterations == sumo.num_iterations


def test_two_runs_are_bitwise_identical_with_the_coarse_space():
    """Galerkin assembly in list order, Gauss-Jordan, wavefront reductions: no atomics anywhere, so two solves give the same bits."""
    g = graphgen.generate(6000, 4000, odom_f_max=2, seed=77, outlier_frac=0.1)
    q, t, s = util.initial_state(g, True)
    outs = []
    for _ in range(2):
        P = util.pgo_problem(g, True, mg_min_keyframes=0)      # (the two-level method)
        outs.append(P.solve(q, t, s))
        P.close()
    (qa, ta, sa, suma), (qb, tb, sb, sumb) = outs
    assert suma.cg_iterations == sumb.cg_iterations and suma.cg_iterations < 20000
    assert np.array_equal(qa, qb) and np.array_equal(ta, tb) and np.array_equal(sa, sb) and suma.final_cost == sumb.final_cost


@pytest.mark.gpu
def test_a_handle_that_kept_it_skips_the_comparison_in_its_next_solves():
    """Incremental triggers solve the same kind of graph again and again: after a solve in which the coarse space won its comparison the
    next three solves of the handle use it without paying for the plain block-Jacobi run; the fifth compares again."""
    g = graphgen.generate(6918, 1382, odom_f_max=3, seed=12, outlier_frac=0.3)
    q, t, s = util.initial_state(g, True)
    P = util.pgo_problem(g, True, mg_min_keyframes=0)      # (the two-level method and its once-per-solve comparison: since round 6 the multigrid would take a graph of this size)
    runs = [P.solve(q, t, s)[3] for _ in range(5)]
    P.close()
    cg = [r.cg_iterations for r in runs]
    assert cg[1] == cg[2] == cg[3] and cg[1] < cg[0] and cg[4] == cg[0]
    for r in runs[1:]:
        assert r.num_iterations == runs[0].num_iterations and abs(r.final_cost - runs[0].final_cost) <= 1e-12 * runs[0].final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 96, 200, 1024, 2048, 2304, 3072])
def test_dense_inverse_against_numpy(n):
    """K6's blocked Gauss-Jordan (upper triangle, fp64 MFMA) against numpy on symmetric positive definite matrices whose conditioning
    resembles a coarse operator's (a graph Laplacian-like stiff part plus a small damping on the diagonal).  Sizes up to 2 304 except 2 048 run the
    one-launch-per-block-step form, 2 048 and 3 072 the panels + update pair (launch_coarse_invert)."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, max(8, n // 2)))
    A = B @ B.T + np.diag(rng.uniform(1e-3, 1.0, n))
    A = 0.5 * (A + A.T)
    P = capi.Problem()
    inv, ms = P.dense_spd_inverse(A)
    P.close()
    assert np.array_equal(inv, inv.T)
    ref = np.linalg.inv(A)
    assert np.abs(inv - ref).max() <= 1e-9 * np.abs(ref).max() * max(1.0, np.linalg.cond(A) * 1e-6)
    assert np.abs(inv @ A - np.eye(n)).max() < 1e-7


@pytest.mark.gpu
def test_dense_inverse_reports_an_indefinite_matrix():
    A = np.eye(128); A[70, 70] = -1.0
    P = capi.Problem()
    with pytest.raises(capi.PgoError):
        P.dense_spd_inverse(A)
    P.close()


def test_a_graph_whose_hierarchy_does_not_coarsen_falls_back_to_the_two_level_method():
    """One 9 600-keyframe trajectory whose every 8th link is a SWITCHABLE closure instead of an odometry edge: level 1 groups keyframes along odometry edges only (1 200
    aggregates of 8), and above level 1 a SINGLE switchable closure between two aggregates counts for nothing (mg_loop_discount) — the matching stalls, the hierarchy cannot be
    built, although the graph is large enough for the multigrid (mg_min_keyframes_switchable) and, being one long chain, hard for plain block-Jacobi.  One GPU prepares the
    hierarchy on a worker thread and only learns this where it is first needed (mg_init_finish): from there on the handle must work with the two-level method — what the same
    graph gets when the multigrid is switched off — not with plain block-Jacobi for the rest of its life (round-4 advisor finding).  Checked: steps preconditioned by the
    two-level method appear in the log (never the multigrid), and the trajectory is the oracle's (banded exact Cholesky)."""
    rng = np.random.default_rng(12)
    N, m = 9600, 8
    q = np.zeros((N, 4)); t = np.zeros((N, 3))
    q[0] = [0, 0, 0, 1]
    for k in range(1, N):      # a random walk with gentle turns
        a = rng.normal(size=3) * 0.05
        dq = np.array([a[0] / 2, a[1] / 2, a[2] / 2, 1.0]); dq /= np.linalg.norm(dq)
        x1, y1, z1, w1 = q[k - 1]; x2, y2, z2, w2 = dq
        q[k] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
        q[k] /= np.linalg.norm(q[k])
    M = util.poses_to_matrices(q, np.zeros((N, 3))).reshape(N, 4, 4).transpose(0, 2, 1)
    for k in range(1, N):
        t[k] = t[k - 1] + M[k - 1][:3, :3] @ np.array([1.0, 0.0, 0.0])
    M = util.poses_to_matrices(q, t).reshape(N, 4, 4).transpose(0, 2, 1)          # row-major 4x4 per keyframe

    def rel(a, b, sig):
        T = np.linalg.inv(M[a]) @ M[b]
        T[:, :3, 3] += rng.normal(size=(len(a), 3)) * sig
        return np.ascontiguousarray(T.transpose(0, 2, 1)).reshape(len(a), 16)       # column-major, the reference's layout
    links = np.arange(N - 1, dtype=np.int32)
    is_sw = (links % m) == m - 1
    skip2 = np.arange(N - 2, dtype=np.int32); skip2 = skip2[(skip2 % m) <= m - 3]      # k -> k + 2 inside a run of 8 (redundancy: the minimum is not a zero-cost tree fit)
    oc1 = np.concatenate([links[~is_sw], skip2]); oc2 = np.concatenate([links[~is_sw] + 1, skip2 + 2])
    lc1 = links[is_sw]; lc2 = lc1 + 1
    oT, lT = rel(oc1, oc2, 0.02), rel(lc1, lc2, 0.02)
    S = len(lc1)
    reg_node = np.zeros(1, np.int32); regT = util.poses_to_matrices(q[:1], t[:1])

    def build(cls, **kw):
        P = cls(**kw)
        P.add_relpose_edges(oc1, oc2, oT, np.ones(len(oc1)))
        P.add_switchable_edges(lc1, lc2, lT, np.ones(S), np.arange(S, dtype=np.int32))
        P.set_node_regularizers(reg_node, regT, np.full(1, 4.0))
        return P
    q0 = q + rng.normal(size=q.shape) * 0.01; q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    t0 = t + rng.normal(size=t.shape) * 0.05
    s0 = np.full(S, 0.99)
    P = build(capi.Problem)
    _, t1, s1, first = P.solve(q0, t0, s0)
    _, t2, s2, again = P.solve(q0, t0, s0)
    P.close()
    pre1 = [first.iterations[k].preconditioner & 15 for k in range(1, first.num_logged)]
    pre2 = [again.iterations[k].preconditioner & 15 for k in range(1, again.num_logged)]
    assert capi.PRECOND_MULTIGRID not in pre1 + pre2, (pre1, pre2)
    assert capi.PRECOND_TWO_LEVEL in pre2, (pre1, pre2)              # at the latest from the handle's second solve on (the first may never have needed the hierarchy)
    from oracle import binding as ob
    O = build(ob.OracleProblem)
    _, to, so, sumo = O.solve(q0, t0, s0)
    for got in (first, again):
        assert [got.iterations[k].step_is_successful for k in range(got.num_logged)] == [sumo.iterations[k].step_is_successful for k in range(sumo.num_logged)]
        for k in range(sumo.num_logged):
            assert abs(sumo.iterations[k].cost - got.iterations[k].cost) <= 1e-6 * max(sumo.iterations[k].cost, 1e-12), k
    assert np.abs(t1 - to).max() <= 2e-2 and np.abs(t2 - to).max() <= 2e-2      # (a 9 600-keyframe chain: the valley is flat along its long wavelengths, as on C2)
