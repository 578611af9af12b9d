"""GPU: the multi-GPU LM logic with TWO ranks on ONE MI355X.  Each rank is a libpgo handle in its own thread owning a contiguous
slice of the edges (sharding.edge_slice); the collective is supplied through pgo_comm_init_custom by an in-process harness that
sums / maxes the two ranks' device buffers.  Everything the 8-GPU run does except RCCL itself is exercised: keyframe-participation
union, all-reduced diagonal blocks + gradient, the lead rank owning damping / regularisers, the all-reduce per CG matvec, scalar
reductions, switch ownership merge.  (RCCL itself: test_gpu_fullsize.py::test_rccl_world_size_one_matches_single_gpu.)"""
import ctypes as C
import threading

import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import capi
from solve_keyframe_pose_graph_amd.sharding import edge_slice
from tests import util

pytestmark = pytest.mark.gpu


class InProcessAllReduce:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.calls = 0

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


@pytest.mark.parametrize("linear_solver", [1, 0])
def test_two_ranks_reproduce_the_single_rank_solve(linear_solver):
    g = util.small_graph(500, 70, f=2, seed=17)
    q, t, s = util.initial_state(g, True)
    opts = dict(cg_rel_tolerance=1e-12, cg_max_iterations=20000, linear_solver=linear_solver)
    P = util.pgo_problem(g, True, **opts)
    q1, t1, s1, sum1 = P.solve(q, t, s)
    P.close()

    world = 2
    ar = InProcessAllReduce(world)
    out = [None] * world
    err = []

    def run(rank):
        try:
            Pr = capi.problem_from_graph(g, switchable=True, edge_slice=edge_slice(rank, world), **opts)
            Pr.comm_init_custom(rank, world, ar.make(rank))
            out[rank] = Pr.solve(q, t, s)
            Pr.comm_destroy()
            Pr.close()
        except Exception as e:   # make a failing rank release the other one
            err.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not err, err
    assert ar.calls > 100                                        # one exchange per CG matvec happened
    for r in range(world):
        qr, tr, sr, sumr = out[r]
        assert sumr.num_iterations == sum1.num_iterations
        assert [sumr.iterations[k].step_is_successful for k in range(sumr.num_logged)] == [sum1.iterations[k].step_is_successful for k in range(sum1.num_logged)]
        assert abs(sumr.final_cost - sum1.final_cost) <= 1e-9 * sum1.final_cost
        assert np.abs(tr - t1).max() <= 1e-7 and np.abs(sr - s1).max() <= 1e-7
    # both ranks hold the same replicated result, including the switches owned by the other rank
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
