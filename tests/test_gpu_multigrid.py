"""GPU: the aggregation multigrid preconditioner (pgo_options.mg_*; csrc/pgo_mg_host.hpp + pgo_mg_kernels.hpp).  A preconditioner does
not change what the PCG converges to, so the checks are: the LM trajectory of the oracle's exact solve and of plain block-Jacobi, far
fewer PCG iterations, every level count (dense only, one and several sparse levels), the hybrid start (block-Jacobi first), constant
keyframes, and bitwise reproducibility."""
import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import graphgen
from tests import util

pytestmark = pytest.mark.gpu


def run(g, switchable=True, state=None, constant=None, **kw):
    q, t, s = state if state is not None else util.initial_state(g, switchable)
    P = util.pgo_problem(g, switchable, **kw)
    if constant is not None:
        P.set_nodes_constant(constant)      # SetParameterBlockConstant (reference src/PoseGraphSLAM.cpp:143-144): keyframes outside the system
    out = P.solve(q, t, s)
    P.close()
    return out


def same_trajectory(a, b, rel):
    assert a.num_iterations == b.num_iterations
    for k in range(a.num_logged):
        x, y = a.iterations[k], b.iterations[k]
        assert x.step_is_successful == y.step_is_successful, k
        assert abs(x.cost - y.cost) <= rel * max(x.cost, 1e-12), (k, x.cost, y.cost)


@pytest.mark.parametrize("dense_max,first,passes", [(512, 3, 2), (96, 3, 2), (24, 2, 2), (40, 1, 3)], ids=["dense_only", "one_sparse_level", "deep", "pairs_then_eights"])
def test_oracle_trajectory_with_every_hierarchy_depth(dense_max, first, passes):
    g = graphgen.generate(2500, 2500, odom_f_max=2, seed=17, outlier_frac=0.1)
    q, t, s = util.initial_state(g, True)
    _, to, so, sumo = util.oracle_problem(g, True).solve(q, t, s)
    _, tp, sp, sump = run(g, True, mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=dense_max, mg_first_passes=first, mg_passes=passes)
    same_trajectory(sumo, sump, 1e-6)
    assert np.abs(tp - to).max() <= 1e-3 and np.abs(sp - so).max() <= 1e-3
    _, _, _, plain = run(g, True, mg_min_keyframes=0, coarse_aggregates=0, cg_max_iterations=200000)
    assert sump.cg_iterations * 3 < plain.cg_iterations, (sump.cg_iterations, plain.cg_iterations)


def test_mid_size_graph_hybrid_start_and_multigrid_from_the_first_iteration():
    """20k keyframes: levels 20000 -> 2500 -> ~430 (dense).  From the first iteration the multigrid needs 3-8x fewer PCG iterations than
    block-Jacobi on the hard (large-radius) systems.  The default start is hybrid: block-Jacobi first, the multigrid takes over a system that
    is not solved after mg_switch_iterations (or whose polled r.z values predict >= 1.75x that many iterations), and a system predicted hard from the
    previous step of the solve starts with it; easy systems never build it.  Same LM trajectory either way."""
    g = graphgen.generate(20000, 20000, odom_f_max=2, seed=3)
    _, t0, s0, plain = run(g, True, mg_min_keyframes=0, coarse_aggregates=0)
    _, t1, s1, mg = run(g, True, mg_min_keyframes=1, mg_switch_iterations=0)
    _, t2, s2, hyb = run(g, True, mg_min_keyframes=1, mg_switch_iterations=400)
    same_trajectory(plain, mg, 1e-6)
    same_trajectory(plain, hyb, 1e-6)
    assert np.abs(t1 - t0).max() <= 1e-4 and np.abs(t2 - t0).max() <= 1e-4
    hard = [k for k in range(1, plain.num_logged) if plain.iterations[k].cg_iterations > 1000]
    assert len(hard) >= 3
    for k in hard:
        assert mg.iterations[k].cg_iterations * 3 < plain.iterations[k].cg_iterations, (k, mg.iterations[k].cg_iterations, plain.iterations[k].cg_iterations)
        assert hyb.iterations[k].cg_iterations < plain.iterations[k].cg_iterations
    # first hard system (no prediction yet): switched in flight — by its measured convergence rate from 96 block-Jacobi iterations on, at the latest after mg_switch_iterations
    assert 96 <= hyb.iterations[hard[0]].cg_iterations < 400 + mg.iterations[hard[0]].cg_iterations + 48
    assert any(hyb.iterations[k].cg_iterations < 400 for k in hard[1:])       # later ones: predicted hard, multigrid from the first iteration
    easy = [k for k in range(1, plain.num_logged) if plain.iterations[k].cg_iterations < 350 and (k == 1 or plain.iterations[k - 1].cg_iterations < 350)]
    # never switched: the same block-Jacobi PCG (its starting point differs from the plain run's within the PCG tolerance of the earlier multigrid steps: +-2 iterations)
    assert easy and all(abs(hyb.iterations[k].cg_iterations - plain.iterations[k].cg_iterations) <= 2 for k in easy)


def test_constant_keyframes_stay_outside_the_hierarchy_and_results_are_reproducible():
    g = graphgen.generate(6000, 4000, odom_f_max=2, seed=23)
    q, t, s = util.initial_state(g, True)
    const = np.r_[0:40, 3000:3010].astype(np.int32)
    outs = []
    for kw in (dict(mg_min_keyframes=0, coarse_aggregates=0, cg_max_iterations=200000), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=128),
               dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=128)):
        P = util.pgo_problem(g, True, **kw)
        P.set_nodes_constant(const)
        outs.append(P.solve(q, t, s))
        P.close()
    (qa, ta, sa, A), (qb, tb, sb, B), (qc, tc, sc, C_) = outs
    same_trajectory(A, B, 1e-6)
    assert np.array_equal(tb.reshape(-1, 3)[const], t[const]) and np.array_equal(qb.reshape(-1, 4)[const], q[const])
    assert np.array_equal(tb, tc) and np.array_equal(qb, qc) and np.array_equal(sb, sc)                              # bitwise reproducible
    assert [B.iterations[k].cg_iterations for k in range(B.num_logged)] == [C_.iterations[k].cg_iterations for k in range(C_.num_logged)]
    assert B.cg_iterations * 3 < A.cg_iterations


def test_a_graph_that_does_not_coarsen_falls_back_to_block_jacobi():
    """loose keyframes (every odometry weight ~ 0, no loops to speak of): the hierarchy builder refuses, the solve runs as without it"""
    g = util.small_graph(400, 3, f=1, seed=2)
    g.odom_w[:] = 1e-6
    _, t0, s0, a = run(g, True, mg_min_keyframes=0, coarse_aggregates=0)
    _, t1, s1, b = run(g, True, mg_min_keyframes=1, coarse_aggregates=0, mg_dense_max_nodes=8)
    assert np.array_equal(t0, t1) and a.cg_iterations == b.cg_iterations


@pytest.mark.parametrize("smoothed", [0, 1, 2])
def test_smoothed_prolongators_keep_the_trajectory_and_cut_the_iterations(smoothed):
    """mg_smoothed_levels: the transitions above level 1 with the smoothed prolongator Ps = (I - w_p D^-1 A) P (the level above = Ps^T A Ps, Ps applied implicitly inside
    the cycle by one more row product before the restriction and after the prolongation).  Same LM trajectory as the oracle's exact solves at every setting; at least
    1.15x fewer PCG iterations than plain aggregation over the whole ten-step solve of this four-level hierarchy (2x on its hard systems)."""
    g = graphgen.generate(20000, 20000, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    _, tp, sp, sump = run(g, True, mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=80, mg_smoothed_levels=smoothed)
    if smoothed == 0:
        test_smoothed_prolongators_keep_the_trajectory_and_cut_the_iterations.base = (sump, tp)
        return
    base, tb = test_smoothed_prolongators_keep_the_trajectory_and_cut_the_iterations.base
    same_trajectory(base, sump, 1e-6)
    assert np.abs(tp - tb).max() <= 1e-4
    assert sump.cg_iterations * 1.15 < base.cg_iterations, (smoothed, sump.cg_iterations, base.cg_iterations)


@pytest.mark.parametrize("shape", ["two_smoothed_deep", "smoothed_below_dense", "default_20k", "multi_world_30k"])
def test_explicit_transfer_operator_is_the_same_cycle(shape):
    """mg_explicit_transfer (round 5): on a level with a smoothed transition above, pre-smoothing + smoothed restriction and smoothed prolongation + post-smoothing are applied
    through R = (Ps - Dinv W)^T (fp32 blocks on W's pattern) — v = x + Dinv (r - A x), r_next = R r, x = v + R^T x_next: two row products on that level instead of four, two
    launches fewer per PCG iteration.  Algebraically the same V(1,1) cycle as the implicit form: same LM trajectory, PCG iteration counts within a few per cent (the operator
    is rounded to fp32 once instead of the level matrix being applied twice in fp32), and the oracle's trajectory where the oracle is affordable.  Shapes: two smoothed
    transitions in a deep hierarchy (a smoothed level below another explicit level: nothing is prolonged into it), a smoothed level directly below the dense one (its restriction
    writes the dense right-hand side, its prolongation reads the dense solution), library defaults on a 20 000-keyframe graph, a multi-world graph with f = 1..5 odometry."""
    oracle = False
    if shape == "two_smoothed_deep":
        g, kw, oracle = graphgen.generate(2500, 2500, odom_f_max=2, seed=17, outlier_frac=0.1), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=24, mg_first_passes=2, mg_smoothed_levels=2), True
    elif shape == "smoothed_below_dense":
        g, kw, oracle = graphgen.generate(2500, 2500, odom_f_max=2, seed=17, outlier_frac=0.1), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=96, mg_first_passes=3, mg_smoothed_levels=1), True
    elif shape == "default_20k":
        g, kw = graphgen.generate(20000, 20000, odom_f_max=2, seed=3), dict(mg_switch_iterations=0)
    else:
        g, kw = graphgen.generate(30000, 6000, odom_f_max=5, apply_yaw_weight=True, n_worlds=3, seed=8), dict(mg_switch_iterations=0)
    _, t0, s0, implicit = run(g, True, mg_explicit_transfer=0, **kw)
    _, t1, s1, explicit = run(g, True, mg_explicit_transfer=1, **kw)
    same_trajectory(implicit, explicit, 1e-7)
    assert np.abs(t1 - t0).max() <= 1e-5 and np.abs(s1 - s0).max() <= 1e-5
    assert explicit.cg_iterations_multigrid > 0 and abs(explicit.cg_iterations - implicit.cg_iterations) <= 0.05 * implicit.cg_iterations + 10, (explicit.cg_iterations, implicit.cg_iterations)
    if oracle:
        q, t, s = util.initial_state(g, True)
        _, to, so, sumo = util.oracle_problem(g, True).solve(q, t, s)
        same_trajectory(sumo, explicit, 1e-6)
        assert np.abs(t1 - to).max() <= 1e-3 and np.abs(s1 - so).max() <= 1e-3


@pytest.mark.parametrize("shape", ["small_with_oracle", "fixed_keyframes", "default_20k", "multi_world_30k"])
def test_smoothed_keyframe_transition_keeps_the_trajectory(shape):
    """mg_smoothed_fine (round 5; round 6: by default where the levels it makes stay below 450 000 blocks, one GPU): the transition keyframes -> level 1 smoothed as well — Ps_0 = (I - w_p D^-1 A) P_0 formed from the keyframe level's own blocks,
    level 1 = Ps_0^T A Ps_0, z = D^-1 r + s Ps_0 V_1(Ps_0^T r) inside the PCG.  Another preconditioner for the same systems: same LM trajectory as the default hierarchy (and as the
    oracle where that is affordable), fewer multigrid iterations."""
    oracle = False
    const = None
    if shape == "small_with_oracle":
        g, kw, oracle = graphgen.generate(2500, 2500, odom_f_max=2, seed=17, outlier_frac=0.1), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=24), True
    elif shape == "fixed_keyframes":
        # keyframes OUTSIDE the system (advisor finding, round 5: this shape never fixed one): the first keyframe, a run in mid-trajectory and single ones — the device paths
        # mg_dinv_kernel's skip_orphans, empty rows of Ps / W, fine_block_value on edges touching fixed keyframes, mg_prolong0s_kernel skipping empty rows; checked against the oracle
        g, kw, oracle = graphgen.generate(6000, 3000, odom_f_max=2, seed=5), dict(mg_min_keyframes=1, mg_switch_iterations=0, mg_dense_max_nodes=64), True
        const = [0, 1, 2999, 3000, 3001, 3002, 4500, 5999]
    elif shape == "default_20k":
        g, kw = graphgen.generate(20000, 20000, odom_f_max=2, seed=3), dict(mg_switch_iterations=0)
    else:
        g, kw = graphgen.generate(30000, 6000, odom_f_max=5, apply_yaw_weight=True, n_worlds=3, seed=8), dict(mg_switch_iterations=0)
    q0, t0, s0, plain = run(g, True, constant=const, mg_smoothed_fine=0, **kw)
    q1, t1, s1, smooth = run(g, True, constant=const, mg_smoothed_fine=1, **kw)
    if const is not None:      # constant keyframes come back bit for bit
        qi, ti, _ = util.initial_state(g, True)
        assert np.array_equal(q1.reshape(-1, 4)[const], qi[const]) and np.array_equal(t1.reshape(-1, 3)[const], ti[const]) and np.array_equal(t0.reshape(-1, 3)[const], ti[const])
    same_trajectory(plain, smooth, 1e-7)
    assert np.abs(t1 - t0).max() <= 1e-5 and np.abs(s1 - s0).max() <= 1e-5
    print("multigrid iterations: default %d, smoothed keyframe transition %d" % (plain.cg_iterations_multigrid, smooth.cg_iterations_multigrid))
    assert 0 < smooth.cg_iterations_multigrid <= 0.8 * plain.cg_iterations_multigrid
    if oracle:
        q, t, s = util.initial_state(g, True)
        O = util.oracle_problem(g, True)
        if const is not None:
            O.set_nodes_constant(const)
        _, to, so, sumo = O.solve(q, t, s)
        same_trajectory(sumo, smooth, 1e-6)
        assert np.abs(t1 - to).max() <= 1e-3 and np.abs(s1 - so).max() <= 1e-3


def test_regroup_follows_the_switches_and_keeps_the_trajectory():
    """mg_regroup_fraction: once the solver has switched the outliers off, the levels above level 1 are matched again along the couplings that are alive (the keyframes'
    level-1 aggregates and level 1's structure are cached).  Same LM trajectory with and without (the preconditioner changes, the steps do not), and a second solve of
    the same handle from the same state is bitwise identical (the hierarchy follows the start again)."""
    g = graphgen.generate(30000, 30000, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    kw = dict(max_num_iterations=16, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, mg_switch_iterations=0)      # (multigrid on every system)
    _, t0, s0, off = run(g, True, mg_regroup_fraction=0.0, **kw)
    P = util.pgo_problem(g, True, mg_regroup_fraction=0.005, **kw)
    _, t1, s1, on = P.solve(q, t, s)
    _, t2, s2, on2 = P.solve(q, t, s)
    P.close()
    same_trajectory(off, on, 1e-6)
    assert np.abs(t1 - t0).max() <= 1e-4 and np.abs(s1 - s0).max() <= 1e-4
    its_on = [on.iterations[k].cg_iterations for k in range(1, on.num_logged)]
    its_off = [off.iterations[k].cg_iterations for k in range(1, off.num_logged)]
    assert its_on[:3] == its_off[:3] and its_on != its_off          # no regroup during the first three LM iterations; afterwards the upper levels are other ones
    assert sum(its_on) <= 1.1 * sum(its_off)                         # (what it buys shows on harder late systems: C3's 305 / 367 / 454 -> 155 / 184 / 246)
    assert on2.final_cost == on.final_cost and np.array_equal(t2, t1) and np.array_equal(s2, s1)
    assert [on2.iterations[k].cg_iterations for k in range(on2.num_logged)] == [on.iterations[k].cg_iterations for k in range(on.num_logged)]


def test_hybrid_schedule_is_reproducible_and_survives_an_abandoned_regroup():
    """The library defaults on a 30k-keyframe graph with outliers (hybrid start, rate-based switch, deferred start, regroup on a worker thread): every decision depends on
    the solve's own history only, so a second handle reproduces iteration counts and state bit for bit.  A solve that is ended right after the step that started the
    regroup's worker (its result is never installed), and a handle destroyed in that state, leave nothing behind: the next solve of the handle is again the same."""
    g = graphgen.generate(30000, 30000, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    kw = dict(max_num_iterations=14, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, mg_regroup_fraction=0.005)
    qa, ta, sa, A = run(g, True, **kw)
    its_a = [A.iterations[k].cg_iterations for k in range(A.num_logged)]
    assert A.cg_iterations_multigrid > 0 and A.cg_iterations_multigrid < A.cg_iterations          # both preconditioners took part
    P = util.pgo_problem(g, True, **kw)
    P.solve_begin(q, t, s)
    for _ in range(4):                     # the worker starts after the third accepted step at the earliest; nothing installs its result in these steps
        P.lm_step(ignore_termination=True)
    P.solve_end()
    qb, tb, sb, B = P.solve(q, t, s)
    assert [B.iterations[k].cg_iterations for k in range(B.num_logged)] == its_a
    assert B.final_cost == A.final_cost and np.array_equal(tb, ta) and np.array_equal(qb, qa) and np.array_equal(sb, sa)
    P.solve_begin(q, t, s)
    for _ in range(4):
        P.lm_step(ignore_termination=True)
    P.close()                              # destroyed with a solve (and possibly a worker) in flight


def test_small_graphs_arm_the_pauses_with_their_first_rejected_step():
    """Below 20 000 keyframes the early-rejection pauses are armed by the solve's first rejected step: a solve whose steps are all accepted is the solve without pauses,
    bit for bit and iteration for iteration."""
    g = graphgen.generate(1500, 300, odom_f_max=5, apply_yaw_weight=1, seed=5, **dict(graphgen._SMALL, turn_deg_per_keyframe=2.0))
    q, t, s = util.initial_state(g, True)
    _, t0, s0, off = run(g, True, max_num_iterations=6, cg_early_tolerance=0.0, cg_mid_tolerance=0.0)
    _, t1, s1, on = run(g, True, max_num_iterations=6)
    assert off.num_unsuccessful_steps == 0
    assert [on.iterations[k].cg_iterations for k in range(on.num_logged)] == [off.iterations[k].cg_iterations for k in range(off.num_logged)]
    assert on.final_cost == off.final_cost and np.array_equal(t1, t0) and np.array_equal(s1, s0)
