#!/usr/bin/env python3
"""The kernels of the multigrid's set-up on in-process ranks, for a kernel trace: `world` ranks of config `name` on one GPU, one LM system, then pgo_time_kernel(8) on every rank (its own
set-up kernels without the exchanges, the ranks taking turns).  Under `rocprofv3 --kernel-trace --stats` the per-kernel averages are per rank and set-up (30 timed + warm-up set-ups per
rank; the exchanges' gather / scatter / copy kernels of the set-ups that ran WITH exchanges are in the same table).
  python scripts/dev/setup_kernels_ranks.py [C5] [8] [mg_dist_setup = 1]"""
import sys
import threading

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solve_keyframe_pose_graph_amd import capi, graphgen, sharding  # noqa: E402
from tests import util  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dist = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
parts = sharding.partition(g, world, "spatial")
group = capi.local_group_create(world)


def run(rank):
    P = capi.problem_from_graph(g, switchable=True, edge_slice=parts[rank], max_num_iterations=1, mg_dist_setup=dist)
    P.comm_init_local(rank, world, group)
    P.solve_begin(q, t, s)
    ms = P.time_kernel(8, 10)[0]
    P.solve_end()
    print("rank %d: set-up kernels %.4f ms" % (rank, ms), flush=True)
    P.comm_destroy()
    P.close()


th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[x.start() for x in th]
[x.join() for x in th]
capi.local_group_destroy(group)
