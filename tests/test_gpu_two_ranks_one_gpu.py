"""GPU: the multi-GPU LM logic with SEVERAL ranks on ONE MI355X.  Each rank is a libpgo handle in its own thread owning a subset of
the edges (sharding.partition: contiguous / chain / spatial); the collective is supplied through pgo_comm_init_custom by an
in-process harness that sums / maxes the ranks' device buffers.  Everything the 8-GPU run does except RCCL itself is exercised:
rank-local subgraphs (each rank works on the keyframes its edges touch), shared-keyframe discovery and ownership, the exchange of
shared rows of the diagonal blocks / gradient / reduced system / CG matvec output, owner-weighted dot products and norms, the
stopped-PCG-keeps-its-state rule behind early rejection, an idle rank, the owner-wise write-back and the switch ownership merge.
(RCCL itself: test_gpu_fullsize.py::test_rccl_world_size_one_matches_single_gpu.)"""
import ctypes as C
import threading

import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import capi
from solve_keyframe_pose_graph_amd import sharding
from tests import util

pytestmark = pytest.mark.gpu


class InProcessAllReduce:
    """`world` ranks as threads of this process.  transport "local": libpgo's own in-process communicator (pgo_comm_init_local: collectives are kernels reading the peers' device
    buffers); "custom": a caller-supplied all-reduce staged through the host (pgo_comm_init_custom) — the neighbour exchanges are then emulated through it by the library;
    "custom+exchange": the same with a caller-supplied exchange (pgo_comm_set_exchange: MPI_Alltoallv semantics)."""

    def __init__(self, world, transport="local"):
        self.world = world
        self.transport = transport
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.xslots = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0
        self.group = capi.local_group_create(world) if transport == "local" else None
        outer = self

        class _Abort:      # the tests call ar.barrier.abort() when a rank fails: release the ranks waiting inside the library too
            def abort(self_inner):
                if outer.group is not None:
                    capi.local_group_abort(outer.group)
                outer._barrier.abort()

            def wait(self_inner, timeout=None):
                return outer._barrier.wait(timeout=timeout)
        self._barrier = self.barrier
        self.barrier = _Abort()

    def attach(self, P, rank):
        if self.transport == "local":
            P.comm_init_local(rank, self.world, self.group)
        else:
            P.comm_init_custom(rank, self.world, self.make(rank))
            if self.transport == "custom+exchange":
                P.comm_set_exchange(self.make_exchange(rank))

    def close(self):
        if self.group is not None:
            capi.local_group_destroy(self.group)
            self.group = None

    def make(self, rank):
        def fn(buf, count, op, stream):
            assert self.hip.hipStreamSynchronize(stream) == 0
            host = np.empty(count, dtype=np.float64)
            assert self.hip.hipMemcpy(host.ctypes.data, buf, count * 8, 2) == 0          # device -> host
            self.slots[rank] = host
            self.barrier.wait(timeout=120)
            red = np.sum(self.slots, axis=0) if op == 0 else np.max(self.slots, axis=0)
            self.barrier.wait(timeout=120)
            assert self.hip.hipMemcpy(buf, red.ctypes.data, count * 8, 1) == 0           # host -> device
            if rank == 0:
                self.calls += 1
            return 0
        return fn

    def make_exchange(self, rank):
        def fn(sbuf, soff, rbuf, roff, stream):
            assert self.hip.hipStreamSynchronize(stream) == 0
            host = np.empty(max(soff[-1], 1), dtype=np.float64)
            if soff[-1]:
                assert self.hip.hipMemcpy(host.ctypes.data, sbuf, soff[-1] * 8, 2) == 0
            self.xslots[rank] = (host, soff)
            self.barrier.wait(timeout=120)
            got = np.empty(max(roff[-1], 1), dtype=np.float64)
            for q in range(self.world):
                hq, sq = self.xslots[q]
                n = roff[q + 1] - roff[q]
                assert sq[rank + 1] - sq[rank] == n, (rank, q, n, sq[rank + 1] - sq[rank])
                got[roff[q]:roff[q + 1]] = hq[sq[rank]:sq[rank + 1]]
            self.barrier.wait(timeout=120)
            if roff[-1]:
                assert self.hip.hipMemcpy(rbuf, got.ctypes.data, roff[-1] * 8, 1) == 0
            return 0
        return fn


def idle_last_rank(g, world):
    """world-1 working ranks (spatial cells) + one rank without a single residual block"""
    return sharding.partition(g, world - 1, "spatial") + [lambda kind, n: np.arange(0)]


@pytest.mark.parametrize("world,policy,linear_solver,transport", [(2, "contiguous", 1, "custom"), (2, "contiguous", 0, "local"), (3, "spatial", 1, "custom+exchange"), (3, "chain", 0, "local"),
                                                                  (4, "spatial", 1, "local"), (3, "idle", 1, "local")])
def test_ranks_reproduce_the_single_rank_solve(world, policy, linear_solver, transport):
    g = util.small_graph(500, 70, f=2, seed=17)
    q, t, s = util.initial_state(g, True)
    opts = dict(cg_rel_tolerance=1e-12, cg_max_iterations=20000, linear_solver=linear_solver)
    P = util.pgo_problem(g, True, **opts)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    c_ref, r_ref, g_ref = P.evaluate(q, t, s)
    P.close()

    parts = idle_last_rank(g, world) if policy == "idle" else sharding.partition(g, world, policy)
    st = sharding.partition_stats(g, parts)
    assert sum(st["edges_per_rank"]) == g.n_odom + g.n_loops and 0 < st["shared_keyframes"] < g.n_poses
    ar = InProcessAllReduce(world, transport)
    out = [None] * world
    grads = [None] * world
    stats = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **opts)
            ar.attach(Pr, rank)
            c, _, gr = Pr.evaluate(q, t, s)                                  # parity hook through the exchange: cost + full gradient
            grads[rank] = (c, gr)
            out[rank] = Pr.solve(q, t, s)
            stats[rank] = Pr.sharding_stats().as_dict()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:   # make a failing rank release the others
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not err, err
    ar.close()
    for r in range(world):                                       # one neighbour exchange of the shared rows and one 2-scalar all-reduce per CG matvec happened
        assert stats[r]["exchanges"] > 100 and stats[r]["allreduces"] > 100 and stats[r]["pcg_iterations"] > 100, stats[r]
        assert stats[r]["bytes_sent_per_bj_iteration"] <= stats[r]["bytes_round5_per_bj_iteration"] + 16
    if transport != "local":
        assert ar.calls > 100
    # gradient over ALL keyframes on every rank; the switch part holds the rank's own switches
    for r in range(world):
        c, gr = grads[r]
        assert abs(c - c_ref) <= 1e-12 * c_ref
        assert np.abs(gr[:6 * g.n_poses] - g_ref[:6 * g.n_poses]).max() <= 1e-10 * np.abs(g_ref).max()
    gs = np.sum([grads[r][1][6 * g.n_poses:] for r in range(world)], axis=0)
    assert np.abs(gs - g_ref[6 * g.n_poses:]).max() <= 1e-10 * np.abs(g_ref).max()
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert sumr.num_iterations == sum1.num_iterations
        assert [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] == [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
        assert abs(sumr.final_cost - sum1.final_cost) <= 1e-9 * sum1.final_cost
        assert np.abs(tr - t1).max() <= 1e-7 and np.abs(sr - s1).max() <= 1e-7
        for k in range(sumr.num_logged):
            assert abs(sumr.iterations[k].step_norm - sum1.iterations[k].step_norm) <= 1e-7 * max(1.0, sum1.iterations[k].step_norm)   # owner-weighted norms
    # every rank returns the same complete result, including keyframes and switches it never touched
    for r in range(1, world):
        assert np.array_equal(out[0][0], out[r][0]) and np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][2], out[r][2])


def test_default_tolerances_with_early_rejection_across_ranks():
    """Library defaults (two-phase PCG with early rejection: a stopped PCG is evaluated, then resumed) on a graph whose solve rejects
    steps: three ranks follow the single-rank accept/reject sequence."""
    g = util.small_graph(1500, 300, f=2, seed=23, outlier_frac=0.3)
    q, t, s = util.initial_state(g, True)
    P = util.pgo_problem(g, True)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    world = 3
    parts = sharding.partition(g, world, "spatial")
    ar = InProcessAllReduce(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank])
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not err, err
    seq1 = [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
    for r in range(world):
        sumr = out[r][3]
        assert [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] == seq1
        assert abs(sumr.final_cost - sum1.final_cost) <= 1e-6 * sum1.final_cost


def test_keyframes_without_residual_blocks_come_back_as_given_on_every_rank():
    g = util.small_graph(300, 40, f=2, seed=19)
    q, t, s = util.initial_state(g, True)
    rng = np.random.default_rng(1)
    extra_q = rng.normal(size=(4, 4)); extra_q /= np.linalg.norm(extra_q, axis=1, keepdims=True)
    q = np.concatenate([q, extra_q]); t = np.concatenate([t, rng.normal(size=(4, 3))])       # four keyframes no edge refers to
    opts = dict(cg_rel_tolerance=1e-12, cg_max_iterations=20000)
    P = util.pgo_problem(g, True, **opts)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    assert np.array_equal(q1.reshape(-1, 4)[-4:], q[-4:]) and np.array_equal(t1.reshape(-1, 3)[-4:], t[-4:])
    world = 2
    parts = sharding.partition(g, world, "chain")
    ar = InProcessAllReduce(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **opts)
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not err, err
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert np.array_equal(qr.reshape(-1, 4)[-4:], q[-4:]) and np.array_equal(tr.reshape(-1, 3)[-4:], t[-4:])
        assert np.abs(tr - t1).max() <= 1e-7 and abs(sumr.final_cost - sum1.final_cost) <= 1e-9 * sum1.final_cost


def test_c3_four_ranks_on_one_gpu_follow_the_single_rank_trajectory():
    """The headline graph (100k keyframes / 300k edges) dealt out to four ranks by the spatial policy, library defaults: the rank-local
    machinery at full size (tens of thousands of shared keyframes, the Chronopoulos-Gear PCG, two-stage early rejection, the multigrid built from
    the gathered global graph with its in-flight switch) reproduces the single-rank accept/reject sequence, per-iteration costs and PCG iteration counts."""
    from solve_keyframe_pose_graph_amd import graphgen
    g = graphgen.config("C3")
    q, t, s = util.initial_state(g, True)
    P = util.pgo_problem(g, True)     # library defaults on BOTH sides: hybrid block-Jacobi / multigrid PCG (the ranks run it in Chronopoulos-Gear form, the cycle distributed)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    world = 4
    parts = sharding.partition(g, world, "spatial")
    st = sharding.partition_stats(g, parts)
    assert 1000 < st["shared_keyframes"] < 0.2 * g.n_poses
    ar = InProcessAllReduce(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank])
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            stats[rank] = Pr.sharding_stats().as_dict()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    stats = [None] * world
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=1200)
    assert not err, err
    ar.close()
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        # round 6: the multigrid is distributed — a rank's level kernels stream about a quarter of the hierarchy's blocks, and what it sends per multigrid iteration is
        # less than a third of what round 5's union all-reduce carried on the same graph
        print("rank %d: %s" % (r, {k: stats[r][k] for k in ("mg_levels", "mg_levels_distributed", "mg_blocks_own", "mg_blocks_total", "bytes_sent_per_mg_iteration", "bytes_round5_per_mg_iteration", "exchanges_per_mg_iteration")}))
        # (C3 on 4 ranks: level 1 — 13 000 rows — is distributed, the smaller levels are run completely by every rank: at these sizes a level kernel sits at its latency floor
        # whatever its share.  Round 5's all-reduce was dominated by the 6 n_1 level-1 vector here; config 5 on 8 ranks, where the shared keyframes dominate: tests/test_gpu_c5.py)
        assert stats[r]["mg_levels_distributed"] >= 1
        assert stats[r]["mg_blocks_own"] <= 0.80 * stats[r]["mg_blocks_total"]
        assert stats[r]["bytes_sent_per_mg_iteration"] <= 0.65 * stats[r]["bytes_round5_per_mg_iteration"]
        assert sumr.num_iterations == sum1.num_iterations == 10
        for k in range(sum1.num_logged):
            a, b = sum1.iterations[k], sumr.iterations[k]
            assert a.step_is_successful == b.step_is_successful, k
            assert abs(a.cost - b.cost) <= 1e-6 * a.cost, (k, a.cost, b.cost)
        assert abs(sumr.cg_iterations - sum1.cg_iterations) <= 0.05 * sum1.cg_iterations, (sumr.cg_iterations, sum1.cg_iterations)
        assert sumr.cg_iterations_multigrid > 0 and sum1.cg_iterations_multigrid > 0
        assert np.abs(tr - t1).max() <= 1e-4 and np.abs(sr - s1).max() <= 1e-4
    assert np.array_equal(out[0][1], out[3][1])


@pytest.mark.parametrize("switchable", [True, False])
def test_constant_keyframes_and_unused_switches_across_ranks(switchable):
    """load_state semantics (SetParameterBlockConstant, reference src/PoseGraphSLAM.cpp:143-144) on three ranks: the constant keyframes
    are shared between ranks here, stay bit-exact, and the solve equals the single-rank one; unused switch slots pass through."""
    g = util.small_graph(400, 60, f=2, seed=29)
    q, t, s = util.initial_state(g, switchable)
    if switchable:
        s = np.concatenate([s, [0.5, 0.25]])
    const = [0, 1, 150, 151, 399]
    opts = dict(cg_rel_tolerance=1e-12, cg_max_iterations=20000, max_num_iterations=30, function_tolerance=1e-10)
    P = util.pgo_problem(g, switchable, **opts)
    P.set_nodes_constant(const)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    world = 3
    parts = sharding.partition(g, world, "chain")
    ar = InProcessAllReduce(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=switchable, edge_slice=parts[rank], **opts)
            Pr.set_nodes_constant(const)                       # the same list on every rank
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not err, err
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert np.array_equal(qr.reshape(-1, 4)[const], q[const]) and np.array_equal(tr.reshape(-1, 3)[const], t[const])
        assert sumr.num_iterations == sum1.num_iterations
        assert abs(sumr.final_cost - sum1.final_cost) <= 1e-9 * sum1.final_cost
        assert np.abs(tr - t1).max() <= 1e-7
        if switchable:
            assert sr[-2] == 0.5 and sr[-1] == 0.25 and np.abs(sr - s1).max() <= 1e-7


@pytest.mark.parametrize("world,policy,switch_at,linear_solver", [(2, "spatial", 0, 1), (3, "chain", 0, 1), (3, "spatial", 60, 1), (2, "contiguous", 0, 0)])
def test_multigrid_across_ranks_follows_the_single_rank_multigrid(world, policy, switch_at, linear_solver):
    """The aggregation multigrid with several ranks (round 6: DISTRIBUTED): the hierarchy is built from the gathered global graph (identical on every rank) with aggregates that
    never mix owners and owner-major numbering; every rank runs the cycle's kernels on the rows it owns and gets the halo rows its kernels read by neighbour exchanges
    (mg_dist_min_rows = 64 here so that the small test levels ARE distributed; the "chain" / "contiguous" cases keep the default and run every level completely on every rank
    from gathered vectors).  Level 1's Galerkin product is still the all-reduced sum of the ranks' parts.  Same accept/reject sequence, costs to the PCG tolerance.
    switch_at > 0: the hybrid start (block-Jacobi first, multigrid operators built in flight) on every rank at the same iteration."""
    from solve_keyframe_pose_graph_amd import graphgen
    g = graphgen.generate(6000, 3000, odom_f_max=2, seed=7)
    q, t, s = util.initial_state(g, True)
    # (mg_smoothed_fine = 0: the smoothed keyframe transition a single handle takes by default at this size does not exist across ranks — the iteration counts are compared form for form)
    opts = dict(mg_min_keyframes=1000, mg_switch_iterations=switch_at, cg_rel_tolerance=1e-11, linear_solver=linear_solver, max_num_iterations=8, mg_dense_max_nodes=64, mg_smoothed_fine=0)
    P = util.pgo_problem(g, True, **opts)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    assert sum1.cg_iterations_multigrid > 0.5 * sum1.cg_iterations
    B = util.pgo_problem(g, True, mg_min_keyframes=0, coarse_aggregates=0, cg_rel_tolerance=1e-11, max_num_iterations=8)
    _, _, _, sumb = B.solve(q, t, s)
    B.close()
    parts = sharding.partition(g, world, policy)
    ar = InProcessAllReduce(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **dict(opts, mg_dist_min_rows=64 if policy == "spatial" else 8192))
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            stats[rank] = Pr.sharding_stats().as_dict()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    stats = [None] * world
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=900)
    assert not err, err
    ar.close()
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert (stats[r]["mg_levels_distributed"] >= 1) == (policy == "spatial"), stats[r]
        if policy == "spatial":
            assert stats[r]["mg_rows_own"] < 0.8 * stats[r]["mg_rows_total"]
        assert sumr.num_iterations == sum1.num_iterations
        assert [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] == [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
        for k in range(sum1.num_logged):
            assert abs(sumr.iterations[k].cost - sum1.iterations[k].cost) <= 1e-8 * sum1.iterations[k].cost, k
        assert np.abs(tr - t1).max() <= 1e-6 and np.abs(sr - s1).max() <= 1e-6
        assert sumr.cg_iterations_multigrid > 0.5 * sumr.cg_iterations
        # (round 6: aggregates never mix owners.  Dealt out by place — the policy to use — that costs nothing; dealt out by index ranges, loop closures cross ranks and
        # the levels above level 1 can no longer follow them: +27 % iterations measured on this graph with 3 index ranges)
        assert abs(sumr.cg_iterations - sum1.cg_iterations) <= (0.15 if policy == "spatial" else 0.35) * sum1.cg_iterations, (sumr.cg_iterations, sum1.cg_iterations)
        assert sumr.cg_iterations < 0.5 * sumb.cg_iterations          # and it is the multigrid that runs: far fewer iterations than block-Jacobi
    for r in range(1, world):
        assert np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][2], out[r][2])


@pytest.mark.parametrize("world,n,loops,policy,opts", [
    (5, 4000, 2000, "idle", dict(mg_smoothed_levels=1)),                     # an idle rank (no residual blocks): empty row and tile ranges on every level
    (8, 3000, 1500, "spatial", dict(mg_smoothed_levels=0)),                  # many ranks, few rows each: ranks that own nothing on the small levels; plain aggregation on every level
    (3, 5000, 5000, "spatial", dict(mg_smoothed_levels=2)),                  # two stacked smoothed transitions, both distributed: explicit operators on consecutive levels
    (2, 4000, 400, "chain", dict(mg_smoothed_levels=1, mg_dense_max_nodes=16)),   # a deep hierarchy (tiny dense level) on index ranges
    (4, 6000, 3000, "spatial", dict(mg_smoothed_levels=1, mg_dist_min_rows=300)),  # distributed and completely-run levels mixed
])
def test_distributed_multigrid_shapes_follow_the_single_handle(world, n, loops, policy, opts):
    """The distributed cycle on hierarchy shapes the benchmark configs do not produce (round 6): empty ranges, ranks without rows on a level, consecutive explicit operators, deep
    hierarchies, distributed and completely-run levels mixed — every level distributed unless the case says otherwise (mg_dist_min_rows = 1).  Same accept/reject sequence and costs
    as the single handle with the same options, identical results on every rank."""
    from solve_keyframe_pose_graph_amd import graphgen
    g = graphgen.generate(n, loops, odom_f_max=2, seed=n // 100 + world)
    q, t, s = util.initial_state(g, True)
    base = dict(mg_min_keyframes=1000, mg_min_keyframes_switchable=1000, mg_switch_iterations=0, cg_rel_tolerance=1e-11, max_num_iterations=6, mg_dense_max_nodes=48, mg_smoothed_fine=0)
    base.update(opts)
    P = util.pgo_problem(g, True, **base)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()
    assert sum1.cg_iterations_multigrid > 0
    parts = idle_last_rank(g, world) if policy == "idle" else sharding.partition(g, world, policy)
    ar = InProcessAllReduce(world)
    out, err, stats = [None] * world, [], [None] * world

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **dict(dict(mg_dist_min_rows=1), **base))
            ar.attach(Pr, rank)
            out[rank] = Pr.solve(q, t, s)
            stats[rank] = Pr.sharding_stats().as_dict()
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=900)
    assert not err, err
    ar.close()
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert stats[r]["mg_levels_distributed"] >= 1 and sumr.cg_iterations_multigrid > 0, stats[r]
        assert sumr.num_iterations == sum1.num_iterations
        assert [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] == [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
        for k in range(sum1.num_logged):
            assert abs(sumr.iterations[k].cost - sum1.iterations[k].cost) <= 1e-8 * sum1.iterations[k].cost, k
        assert np.abs(tr - t1).max() <= 1e-6 and np.abs(sr - s1).max() <= 1e-6
        assert sumr.pcg_retries == 0                                          # the distributed cycle is a positive definite preconditioner: no breakdown, no retry
        assert sumr.cg_iterations <= 1.6 * sum1.cg_iterations, (sumr.cg_iterations, sum1.cg_iterations)
    for r in range(1, world):
        assert np.array_equal(out[0][1], out[r][1]) and np.array_equal(out[0][2], out[r][2])


def test_cpp_example_runs_four_ranks_from_one_process():
    """examples/ranks_in_process.cpp: the sharded solver through the C-ABI alone — pgo_partition_edges, one handle per rank on its own thread, pgo_comm_init_local — against the
    single handle (the program exits non-zero when decisions or the final cost differ)."""
    import subprocess
    from solve_keyframe_pose_graph_amd import _build
    exe = _build.build_example_ranks()
    r = subprocess.run([exe, "12000", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "decisions equal" in r.stdout and "4 ranks" in r.stdout, r.stdout
    print(r.stdout)
