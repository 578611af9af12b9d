"""GPU: randomised small graphs (sizes straddling the 64-edge K1 tiles, the 512-side / 85-keyframe matrix-free tiles and the LDS pose
windows; mixed edge classes; regularisers on arbitrary keyframes; constant keyframes; isolated keyframes) — every one is compared with
the oracle: objective + gradient, the damped Schur-reduced operator, and a 3-iteration LM solve with both matvec forms."""
import numpy as np
import pytest

from oracle import binding as ob
from solve_keyframe_pose_graph_amd import capi
from tests import util
from tests.golden.make_functor_goldens import quat_to_R_np

pytestmark = pytest.mark.gpu


def rand_graph(rng, N, n_rel_extra, n_sw, f_max, n_pri, n_const):
    q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    t = rng.normal(size=(N, 3)) * 3

    def T_of(i):
        T = np.eye(4); T[:3, :3] = quat_to_R_np(q[i]); T[:3, 3] = t[i]
        return T

    def meas(a, b, noise):
        M = np.linalg.inv(T_of(a)) @ T_of(b)
        M[:3, 3] += rng.normal(size=3) * noise
        return M.flatten(order="F")
    rel = []
    for u in range(N):
        for f in range(1, f_max + 1):
            if u - f >= 0:
                rel.append((u, u - f, meas(u, u - f, 0.05), 0.9 ** f))
    for _ in range(n_rel_extra):
        a, b = rng.choice(N, 2, replace=False)
        rel.append((int(a), int(b), meas(a, b, 0.1), rng.uniform(0.2, 1.2)))
    sw = []
    for k in range(n_sw):
        a, b = rng.choice(N, 2, replace=False)
        sw.append((int(a), int(b), meas(a, b, 0.1 if rng.uniform() > 0.2 else 3.0), 1.0, k))
    pri_nodes = rng.choice(N, size=min(n_pri, N), replace=False)
    pri = [(int(n), (T_of(n) @ np.diag([1, 1, 1, 1.0])).flatten(order="F"), rng.uniform(1.1, 5.0)) for n in pri_nodes]
    const = list(rng.choice(N, size=min(n_const, max(N - 2, 0)), replace=False)) if n_const else []
    # perturbed starting point
    q0 = q + rng.normal(size=q.shape) * 0.03; q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    t0 = t + rng.normal(size=t.shape) * 0.1
    s0 = rng.uniform(0.3, 1.1, size=n_sw)
    return rel, sw, pri, const, q0, t0, s0


def build(cls_problem, rel, sw, pri, const):
    P = cls_problem()
    if rel:
        P.add_relpose_edges([r[0] for r in rel], [r[1] for r in rel], np.array([r[2] for r in rel]), [r[3] for r in rel])
    if sw:
        P.add_switchable_edges([r[0] for r in sw], [r[1] for r in sw], np.array([r[2] for r in sw]), [r[3] for r in sw], [r[4] for r in sw])
    if pri:
        P.set_node_regularizers([r[0] for r in pri], np.array([r[1] for r in pri]), [r[2] for r in pri])
    if const:
        P.set_nodes_constant(const)
    return P


CASES = [  # N, extra relpose, switchable, f_max, priors, constants
    (3, 0, 1, 1, 1, 0), (9, 3, 0, 2, 1, 0), (64, 10, 20, 1, 1, 0), (65, 0, 64, 1, 2, 1), (86, 30, 40, 5, 1, 2), (128, 64, 65, 2, 3, 0),
    (200, 300, 100, 3, 1, 5), (513, 100, 128, 1, 2, 0), (700, 500, 300, 5, 4, 3), (97, 0, 0, 5, 1, 0), (150, 40, 129, 4, 0, 1),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_graph_matches_oracle(case):
    N, nrel, nsw, fmax, npri, nconst = CASES[case]
    rng = np.random.default_rng(1000 + case)
    rel, sw, pri, const, q0, t0, s0 = rand_graph(rng, N, nrel, nsw, fmax, npri, nconst)
    if case == 5:   # one isolated keyframe appended: must pass through
        q0 = np.vstack([q0, [0, 0, 0, 1.0]]); t0 = np.vstack([t0, [9.0, 9, 9]])
    O = build(ob.OracleProblem, rel, sw, pri, const)
    co, ro, go = O.evaluate(q0, t0, s0)
    for ls in (1, 0):
        P = build(lambda: capi.Problem(linear_solver=ls, cg_rel_tolerance=1e-13, cg_max_iterations=20000, max_num_iterations=3), rel, sw, pri, const)
        cp, rp, gp = P.evaluate(q0, t0, s0)
        assert abs(cp - co) <= 1e-11 * max(1.0, co)
        assert np.abs(rp - ro).max() <= 1e-11 * max(1.0, np.abs(ro).max())
        assert np.abs(gp - go).max() <= 1e-10 * max(1.0, np.abs(go).max())
        if npri > 0 or fmax > 0:
            qo, to, so, sumo = O.solve(q0, t0, s0, ob.default_options(max_num_iterations=3))
            qp, tp, sp, sump = P.solve(q0, t0, s0)
            assert sump.num_iterations == sumo.num_iterations, (sump.num_iterations, sumo.num_iterations)
            assert [sump.iterations[k].step_is_successful for k in range(sump.num_logged)] == [sumo.iterations[k].step_is_successful for k in range(sumo.num_logged)]
            assert abs(sump.final_cost - sumo.final_cost) <= 1e-7 * max(sumo.final_cost, 1e-9), (sump.final_cost, sumo.final_cost)
            assert np.abs(tp - to).max() <= 1e-5 and (sp.size == 0 or np.abs(sp - so).max() <= 1e-5)
            if const:
                assert np.array_equal(tp.reshape(-1, 3)[const], t0[const])
        P.close()


def test_early_rejection_never_changes_the_accept_reject_sequence():
    """Two-stage early rejection (pgo.h: cg_early_tolerance / cg_mid_tolerance) is not a Ceres rule, so it must be invisible: over 240 random
    session-sized graphs with 0-40 % outlier loop closures (the distribution of scripts/gpu_fuzz_soak.py, hundreds of rejected steps) the
    accept/reject sequence and every accepted iterate's cost are those of the run without the stages, with markedly fewer PCG iterations."""
    rng = np.random.default_rng(2025)
    n_graphs, rejected, cg_on, cg_off, fired = 0, 0, 0, 0, 0
    while n_graphs < 240:
        n = int(rng.integers(80, 900)); loops = int(rng.integers(5, max(6, n // 3))); f = int(rng.integers(1, 6))
        out = float(rng.choice([0.0, 0.1, 0.2, 0.4])); seed = int(rng.integers(1, 10 ** 6))
        g = util.small_graph(n, loops, f=f, seed=seed, outlier_frac=out, min_loop_gap=int(rng.integers(5, 30)))
        if g.n_loops == 0:
            continue
        n_graphs += 1
        q, t, s = util.initial_state(g, True)
        res = []
        for kw in (dict(cg_early_tolerance=0.0, cg_mid_tolerance=0.0), dict()):
            P = util.pgo_problem(g, True, **kw)
            res.append(P.solve(q, t, s)[3])
            P.close()
        off, on = res
        seq_off = [off.iterations[i].step_is_successful for i in range(off.num_logged)]
        seq_on = [on.iterations[i].step_is_successful for i in range(on.num_logged)]
        assert seq_on == seq_off, (n, loops, f, out, seed, seq_on, seq_off)
        for i in range(off.num_logged):
            assert abs(on.iterations[i].cost - off.iterations[i].cost) <= 1e-9 * max(off.iterations[i].cost, 1e-12), (n, loops, f, out, seed, i)
        rejected += seq_off.count(0)
        cg_on += on.cg_iterations; cg_off += off.cg_iterations
        fired += sum(1 for i in range(1, on.num_logged) if not seq_on[i] and on.iterations[i].cg_iterations < off.iterations[i].cg_iterations)
    assert rejected >= 100 and fired >= 10, (rejected, fired)
    assert cg_on < cg_off
