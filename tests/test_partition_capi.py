"""CPU: the edge-sharding policies behind the C-ABI (pgo_partition_edges, host only) against solve_keyframe_pose_graph_amd/sharding.py — the same
rank for every edge and the same part for every keyframe, for all three policies and several world sizes."""
import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import capi, graphgen, sharding
from tests import util


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("policy", ["contiguous", "chain", "spatial"])
def test_c_abi_partition_equals_the_python_policies(policy, world):
    g = graphgen.generate(5000, 2200, odom_f_max=2, seed=9)
    part, rr, sr = capi.partition_edges(policy, world, g.init_t, g.odom_c1, g.odom_c2, g.loop_c1, g.loop_c2)
    sels = sharding.partition(g, world, policy)
    rel_ref = np.full(g.n_odom, -1); sw_ref = np.full(g.n_loops, -1)
    for r, sel in enumerate(sels):
        rel_ref[sel("odom", g.n_odom)] = r
        sw_ref[sel("loop", g.n_loops)] = r
    assert np.array_equal(rr, rel_ref) and np.array_equal(sr, sw_ref)
    if policy != "contiguous":
        assert np.array_equal(part, sharding.keyframe_parts(g, world, policy))
        counts = np.bincount(np.concatenate([rr, sr]), minlength=world)
        assert counts.max() <= 1.05 * counts.mean() + 8            # balanced by edge load


def test_c_abi_partition_rejects_bad_input():
    g = util.small_graph(100, 10)
    with pytest.raises(capi.PgoError):
        capi.partition_edges("spatial", 0, g.init_t, g.odom_c1, g.odom_c2, g.loop_c1, g.loop_c2)
    bad = g.odom_c1.copy(); bad[0] = 100
    with pytest.raises(capi.PgoError):
        capi.partition_edges("chain", 2, g.init_t, bad, g.odom_c2, g.loop_c1, g.loop_c2)
