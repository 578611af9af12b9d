"""Edge sharding for multi-GPU solves.  Residual blocks are the independent units: every rank adds a subset of the
edges to its own libpgo handle.  Inside libpgo a rank then works on the keyframes ITS edges touch (a rank-local subgraph); keyframes
touched by two or more ranks are "shared" and only their rows travel: ONE all-reduce of 6 x n_shared + 2 doubles per CG iteration
(both dot products ride along), 42 x n_shared twice per LM iteration.  How the edges are dealt out therefore decides the exchange
volume — the policies here differ only in that:

  contiguous   a contiguous index range of every edge class per rank (what a caller gets without thinking; loop closures land on
               arbitrary ranks, so most keyframes end up shared)
  chain        keyframes split into `world` consecutive index ranges, an edge goes to the range of its LATER endpoint:
               odometry stays local, a loop closure drags in its earlier endpoint
  spatial      keyframes split by recursive coordinate bisection of their initial positions into `world` equally sized cells, an edge
               goes to the cell of its later endpoint: loop closures connect places that are close in space, so only keyframes near
               cell boundaries are shared

Regularisers go to the rank whose part holds their keyframe (contiguous: rank 0)."""
import numpy as np


def edge_slice(rank, world):
    """The `contiguous` policy as a selector `sel(kind, n) -> index array` for capi.problem_from_graph."""
    def sel(kind, n):
        if kind == "reg":
            return np.arange(n) if rank == 0 else np.arange(0)
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        return np.arange(lo, hi)
    return sel


def keyframe_parts(g, world, policy):
    """-> int array [n_poses]: the part (rank) of every keyframe under `chain` or `spatial`.  Parts are balanced by the number of
    edges they will receive (edges whose later endpoint lies in the part), not by keyframes."""
    n = g.n_poses
    load = np.bincount(np.maximum(g.odom_c1, g.odom_c2), minlength=n).astype(np.float64)
    if g.n_loops:
        load += np.bincount(np.maximum(g.loop_c1, g.loop_c2), minlength=n)
    load += 1e-3                                   # keyframes without edges still spread evenly
    if policy == "chain":
        c = np.cumsum(load)
        return np.minimum((c - load) * world / c[-1], world - 1).astype(np.int32)
    if policy != "spatial":
        raise ValueError("unknown partition policy %r" % policy)
    part = np.zeros(n, np.int32)
    pos = np.asarray(g.init_t, dtype=np.float64)

    def split(idx, lo, hi):          # cells [lo, hi) get the keyframes `idx`
        if hi - lo <= 1:
            part[idx] = lo
            return
        mid = (lo + hi) // 2
        p = pos[idx]
        axis = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        order = idx[np.argsort(p[:, axis], kind="stable")]
        c = np.cumsum(load[order])
        k = int(np.searchsorted(c, c[-1] * (mid - lo) / (hi - lo)))
        k = min(max(k, 1), len(order) - 1) if len(order) > 1 else len(order)
        split(order[:k], lo, mid)
        split(order[k:], mid, hi)
    split(np.arange(n), 0, world)
    return part


def partition(g, world, policy="spatial"):
    """-> list of selectors, one per rank (see edge_slice)."""
    if policy == "contiguous":
        return [edge_slice(r, world) for r in range(world)]
    part = keyframe_parts(g, world, policy)
    own = {"odom": part[np.maximum(g.odom_c1, g.odom_c2)], "loop": part[np.maximum(g.loop_c1, g.loop_c2)] if g.n_loops else np.zeros(0, np.int32),
           "reg": part[g.reg_node] if len(g.reg_node) else np.zeros(0, np.int32)}

    def make(rank):
        def sel(kind, n):
            assert n == len(own[kind])
            return np.nonzero(own[kind] == rank)[0]
        return sel
    return [make(r) for r in range(world)]


def partition_stats(g, selectors):
    """What the partition costs: edges per rank and the number of keyframes shared between ranks (the rows every exchange carries)."""
    touched = np.zeros(g.n_poses, np.int32)
    edges, kf = [], []
    for sel in selectors:
        t = np.zeros(g.n_poses, bool)
        io, il, ir = sel("odom", g.n_odom), sel("loop", g.n_loops), sel("reg", len(g.reg_node))
        t[g.odom_c1[io]] = True; t[g.odom_c2[io]] = True
        if len(il):
            t[g.loop_c1[il]] = True; t[g.loop_c2[il]] = True
        if len(ir):
            t[g.reg_node[ir]] = True
        touched += t
        edges.append(len(io) + len(il))
        kf.append(int(t.sum()))
    return {"edges_per_rank": edges, "keyframes_per_rank": kf, "shared_keyframes": int((touched >= 2).sum()), "keyframes": int(g.n_poses)}
