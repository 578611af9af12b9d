"""GPU: the multigrid's DISTRIBUTED SET-UP (pgo_options.mg_dist_setup = 1, pgo_solver.hip: build_mg_ranks) against the replicated one (= 0: level 1's blocks all-reduced, every
level above formed by every rank) with in-process ranks on one GPU.  After ONE LM iteration from the same state with the multigrid from the first PCG iteration — one set-up on
identical inputs — what every rank's cycle kernels read of every level (its rows' fp64 blocks, their fp32 copy, block-Jacobi inverses, R^T, R, the dense inverse:
pgo_mg_level_norms) agrees to the order of the sums; full solves take the same decisions with the same costs.  (Host side of the same thing, replayed on the CPU:
tests/test_mg_distributed.py::test_distributed_setup_replayed_with_scalar_blocks.)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,world,dist_min,smoothed,policy,f", [
    (6000, 3, 300, 1, "spatial", 2),      # one distributed level with a smoothed transition above it, the level above gathered
    (12000, 5, 64, 2, "spatial", 2),      # two distributed levels, both transitions smoothed
    (6000, 4, 1, 0, "spatial", 2),        # plain transitions only (what graphs beyond 500 000 keyframes get: BASELINE config 5)
    (12000, 5, 64, 1, "spatial", 5),      # f = 1..5 + yaw weights: the smoother's safety rescaling triggers (the distributed power method: the iterate's halo exchanged before every step)
    (6000, 3, 300, 1, "chain", 2),        # a partition by index ranges: loop closures cross ranks
])
def test_distributed_setup_forms_the_same_operators(n, world, dist_min, smoothed, policy, f):
    import gpu_dist_setup_check as chk
    assert chk.check(n, world, dist_min, smoothed, policy, f=f, verbose=False)


def test_setup_counters_on_a_sharded_graph():
    """what the set-up sends and forms, from the library's own counters: a rank forms its share of the blocks, its block exchanges carry less than the all-reduce of level 1 did"""
    import threading

    from solve_keyframe_pose_graph_amd import capi, graphgen, sharding
    from tests import util
    g = graphgen.generate(30000, 30000, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    world = 4
    parts = sharding.partition(g, world, "spatial")
    group = capi.local_group_create(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            P = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], mg_dist_min_rows=500, max_num_iterations=3, mg_switch_iterations=0)
            P.comm_init_local(rank, world, group)
            P.solve(q, t, s)
            out[rank] = P.sharding_stats().as_dict()
            P.comm_destroy(); P.close()
        except Exception as e:   # noqa: BLE001
            err.append(repr(e)); capi.local_group_abort(group)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    capi.local_group_destroy(group)
    assert not err, err
    for st in out:
        assert st["mg_setup_levels_own_rows"] >= 1 and st["mg_setup_levels_own_rows"] == st["mg_levels_distributed"]
        assert st["mg_setup_blocks_own"] < 0.6 * st["mg_setup_blocks_total"]
        assert 0 < st["bytes_sent_per_mg_setup"] < st["bytes_allreduce_replicated_setup"]
        assert st["mg_setup_exchanges"] >= 3


def test_setup_kernels_can_be_timed_alone_on_one_handle_and_on_ranks():
    """pgo_time_kernel(8): the kernels of one multigrid set-up — on ranks every rank's own, without the exchanges (the ranks take turns on the GPU), after which the operators are
    formed again properly: the solve that follows takes the decisions of the single handle"""
    import threading

    from solve_keyframe_pose_graph_amd import capi, graphgen, sharding
    from tests import util
    g = graphgen.generate(12000, 12000, odom_f_max=2, seed=3)
    q, t, s = util.initial_state(g, True)
    opts = dict(mg_min_keyframes=1000, mg_min_keyframes_switchable=1000, mg_switch_iterations=0, mg_smoothed_fine=0, max_num_iterations=4)
    P = util.pgo_problem(g, True, **opts)
    P.solve_begin(q, t, s)
    ms1, _ = P.time_kernel(8, 5)
    P.solve_end()
    _, _, _, sum1 = P.solve(q, t, s)
    with pytest.raises(capi.PgoError):
        P.mg_level_norms(99)
    P.close()
    assert 0.05 < ms1 < 50.0
    world = 3
    parts = sharding.partition(g, world, "spatial")
    group = capi.local_group_create(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], mg_dist_min_rows=200, **opts)
            Pr.comm_init_local(rank, world, group)
            Pr.solve_begin(q, t, s)
            ms, _ = Pr.time_kernel(8, 5)
            Pr.solve_end()
            res = Pr.solve(q, t, s)
            out[rank] = (ms, res[3], Pr.sharding_stats().as_dict())
            Pr.comm_destroy(); Pr.close()
        except Exception as e:   # noqa: BLE001
            err.append(repr(e)); capi.local_group_abort(group)
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [x.start() for x in th]; [x.join() for x in th]
    capi.local_group_destroy(group)
    assert not err, err
    for ms, summ, st in out:
        assert 0.05 < ms < 50.0 and st["mg_setup_levels_own_rows"] >= 1
        assert [summ.iterations[k].step_is_successful for k in range(summ.num_logged)] == [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
        assert abs(summ.final_cost - sum1.final_cost) <= 1e-7 * sum1.final_cost


@pytest.mark.parametrize("transport", ["custom", "custom+exchange"])
def test_distributed_setup_through_a_caller_supplied_collective(transport):
    """the set-up's block exchanges through pgo_comm_init_custom: with an exchange callback (all-to-all-v semantics), and without one — the library then emulates every exchange by an
    all-reduce of a zero-padded buffer (also the fp32 blocks of R, which travel as raw bytes: adding zeros to a finite double changes nothing).  Same bits as the in-process
    communicator's solve: the parts of a block are added in rank order on the device whatever carried them."""
    import threading

    import numpy as np

    from solve_keyframe_pose_graph_amd import capi, graphgen, sharding
    from tests import util
    from tests.test_gpu_two_ranks_one_gpu import InProcessAllReduce
    g = graphgen.generate(6000, 3000, odom_f_max=2, seed=5)
    q, t, s = util.initial_state(g, True)
    world = 3
    parts = sharding.partition(g, world, "spatial")
    opts = dict(mg_min_keyframes=1000, mg_min_keyframes_switchable=1000, mg_switch_iterations=0, mg_smoothed_fine=0, mg_smoothed_levels=1, mg_dist_min_rows=100, max_num_iterations=5)

    def solve(tr):
        ar = InProcessAllReduce(world, tr)
        out, err = [None] * world, []

        def run(rank):
            try:
                P = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], **opts)
                ar.attach(P, rank)
                res = P.solve(q, t, s)
                out[rank] = (res, P.sharding_stats().as_dict())
                P.comm_destroy(); P.close()
            except Exception as e:   # noqa: BLE001
                err.append(repr(e)); ar.barrier.abort()
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [x.start() for x in th]; [x.join() for x in th]
        ar.close()
        assert not err, err
        return out
    A, B = solve(transport), solve("local")
    assert A[0][1]["mg_setup_levels_own_rows"] >= 1 and A[0][1]["mg_setup_exchanges"] >= 10
    for r in range(world):
        assert np.array_equal(A[r][0][1], B[0][0][1]) and np.array_equal(A[r][0][2], B[0][0][2]) and np.array_equal(A[r][0][0], B[0][0][0])
        assert A[r][0][3].cg_iterations == B[0][0][3].cg_iterations and A[r][0][3].pcg_retries == 0
