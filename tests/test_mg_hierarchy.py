"""CPU: the host-side hierarchy of the aggregation multigrid (csrc/pgo_mg_host.hpp: graph-following aggregates by heavy-edge matching,
contiguous renumbering, block structures, Galerkin contribution lists) checked for its invariants through tests/native/mg_host.cpp.
Host logic coverage; the cycle runs in HIP kernels (tests/test_gpu_multigrid.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import graphgen

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def shim():
    so = os.path.join(HERE, "native", "libmg_host.so")
    src = os.path.join(HERE, "native", "mg_host.cpp")
    hdr = os.path.join(ROOT, "solve_keyframe_pose_graph_amd", "csrc", "pgo_mg_host.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.dirname(hdr), "-o", so, src])
    lib = C.CDLL(so)
    lib.mgh_build.restype = C.c_void_p
    return lib


def I32(x): return np.ascontiguousarray(x, dtype=np.int32)
def ptr(a, t): return a.ctypes.data_as(C.POINTER(t))


def build(lib, g, free=None, passes0=3, passes=2, dense_max=64, tile_rows=32, max_levels=12, level0_loops=True, smoothed=0):
    N = g.n_poses
    nf = np.ones(N, np.uint8) if free is None else np.ascontiguousarray(free, dtype=np.uint8)
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    h = lib.mgh_build(C.c_longlong(N), ptr(nf, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)), ptr(sc1, C.c_int),
                      ptr(sc2, C.c_int), passes0, passes, dense_max, tile_rows, max_levels, 1 if level0_loops else 0, smoothed)
    if not h:
        return None
    h = C.c_void_p(h)
    levels = []
    for l in range(lib.mgh_levels(h)):
        sz = np.zeros(6, np.int64)
        lib.mgh_sizes(h, l, ptr(sz, C.c_longlong))
        n, nnzb, nent, npar, nagg, ntile = [int(x) for x in sz]
        L = dict(n=n, rowptr=np.zeros(n + 1, np.int64), col=np.zeros(nnzb, np.int32), g_ptr=np.zeros(nnzb + 1, np.int64), g_ent=np.zeros(nent, np.int64),
                 parent=np.zeros(npar, np.int32), agg_ptr=np.zeros(nagg, np.int32), tile_agg0=np.zeros(ntile, np.int32))
        lib.mgh_level(h, l, ptr(L["rowptr"], C.c_longlong), ptr(L["col"], C.c_int), ptr(L["g_ptr"], C.c_longlong), ptr(L["g_ent"], C.c_longlong), ptr(L["parent"], C.c_int),
                      ptr(L["agg_ptr"], C.c_int), ptr(L["tile_agg0"], C.c_int))
        ss = np.zeros(3, np.int64)
        lib.mgh_smoothed_sizes(h, l, ptr(ss, C.c_longlong))
        if ss[0] >= 0:
            S = dict(ps_rowptr=np.zeros(n + 1, np.int32), ps_col=np.zeros(int(ss[0]), np.int32), w_rowptr=np.zeros(n + 1, np.int32), w_col=np.zeros(int(ss[1]), np.int32),
                     psT_ptr=np.zeros(nagg, np.int64), psT_ent=np.zeros(int(ss[2]), np.int64))
            lib.mgh_smoothed(h, l, ptr(S["ps_rowptr"], C.c_int), ptr(S["ps_col"], C.c_int), ptr(S["w_rowptr"], C.c_int), ptr(S["w_col"], C.c_int), ptr(S["psT_ptr"], C.c_longlong), ptr(S["psT_ent"], C.c_longlong))
            S.update(ps_of_w=np.zeros(int(ss[1]), np.int32), rT_rowptr=np.zeros(nagg, np.int32), rT_col=np.zeros(int(ss[1]), np.int32), rT_of_w=np.zeros(int(ss[1]), np.int32))
            seg = C.c_int(0)
            lib.mgh_explicit(h, l, ptr(S["ps_of_w"], C.c_int), ptr(S["rT_rowptr"], C.c_int), ptr(S["rT_col"], C.c_int), ptr(S["rT_of_w"], C.c_int), C.byref(seg))
            S["rT_seg"] = seg.value
            L["smoothed"] = S
        levels.append(L)
    n1 = levels[0]["n"]
    agg0 = np.zeros(N, np.int32); mem0_ptr = np.zeros(n1 + 1, np.int32)
    nmem = int((nf != 0).sum())
    mem0 = np.zeros(nmem, np.int32)
    lib.mgh_level0(h, ptr(agg0, C.c_int), ptr(mem0_ptr, C.c_int), ptr(mem0, C.c_int))
    lib.mgh_free(h)
    return dict(levels=levels, agg0=agg0, mem0_ptr=mem0_ptr, mem0=mem0)


def check(g, H, free, passes0, passes, dense_max, level0_loops=True):
    N = g.n_poses
    L = H["levels"]
    agg0 = H["agg0"]
    n1 = L[0]["n"]
    # every free keyframe in exactly one level-1 aggregate of <= 2^passes0 keyframes; fixed ones in none
    assert np.array_equal(agg0 >= 0, free.astype(bool))
    assert set(agg0[agg0 >= 0]) == set(range(n1))
    sizes = np.bincount(agg0[agg0 >= 0], minlength=n1)
    assert sizes.min() >= 1 and sizes.max() <= 2 ** passes0
    for a in range(n1):
        mem = H["mem0"][H["mem0_ptr"][a]:H["mem0_ptr"][a + 1]]
        assert len(mem) == sizes[a] and np.all(agg0[mem] == a)
    # aggregates are connected through the graph (matching only ever merges across an edge)
    nbr = [set() for _ in range(N)]
    for c1, c2 in list(zip(g.odom_c1, g.odom_c2)) + (list(zip(g.loop_c1, g.loop_c2)) if level0_loops else []):   # the library's level 1 follows odometry edges only
        nbr[c1].add(c2); nbr[c2].add(c1)
    for a in range(0, n1, max(1, n1 // 200)):
        mem = set(H["mem0"][H["mem0_ptr"][a]:H["mem0_ptr"][a + 1]].tolist())
        seen, todo = set(), [next(iter(mem))]
        while todo:
            x = todo.pop()
            if x in seen: continue
            seen.add(x); todo += [y for y in nbr[x] if y in mem and y not in seen]
        assert seen == mem
    # level 1 blocks: exactly the aggregate pairs the free-free edges connect, diagonal first, both triangles; every contribution listed once
    fe = [(e, c1, c2, 1) for e, (c1, c2) in enumerate(zip(g.odom_c1, g.odom_c2))] + [(e, c1, c2, 3) for e, (c1, c2) in enumerate(zip(g.loop_c1, g.loop_c2))]
    want = {(a, a) for a in range(n1)}
    want_ent = {}
    for i in range(N):
        if agg0[i] >= 0: want_ent.setdefault((agg0[i], agg0[i]), []).append((i << 3) | 0)
    for e, c1, c2, kind in fe:
        a, b = agg0[c1], agg0[c2]
        if a < 0 or b < 0: continue
        want.add((a, b)); want.add((b, a))
        want_ent.setdefault((a, b), []).append((e << 3) | kind); want_ent.setdefault((b, a), []).append((e << 3) | (kind + 1))
    A = L[0]
    got = set()
    for r in range(n1):
        cols = A["col"][A["rowptr"][r]:A["rowptr"][r + 1]]
        assert cols[0] == r and len(set(cols.tolist())) == len(cols)
        for k in range(A["rowptr"][r], A["rowptr"][r + 1]):
            got.add((r, int(A["col"][k])))
            ent = A["g_ent"][A["g_ptr"][k]:A["g_ptr"][k + 1]].tolist()
            assert sorted(ent) == sorted(want_ent[(r, int(A["col"][k]))])
    assert got == want
    # coarser levels
    for l in range(len(L) - 1):
        A, B = L[l], L[l + 1]
        par, ap = A["parent"], A["agg_ptr"]
        assert len(par) == A["n"] and len(ap) == B["n"] + 1 and ap[0] == 0 and ap[-1] == A["n"]
        assert np.all(np.diff(par) >= 0) and par[0] == 0 and par[-1] == B["n"] - 1            # members contiguous, parents ascending
        assert np.all(np.diff(ap) >= 1) and np.all(np.diff(ap) <= 2 ** passes)
        for a in range(B["n"]):
            assert np.all(par[ap[a]:ap[a + 1]] == a)
        t = A["tile_agg0"]
        assert t[0] == 0 and t[-1] == B["n"] and np.all(np.diff(t) >= 1)
        assert all(ap[t[k + 1]] - ap[t[k]] <= 32 for k in range(len(t) - 1))
        rows = np.repeat(np.arange(A["n"]), np.diff(A["rowptr"]))
        wantB = {}
        for k in range(len(A["col"])):
            wantB.setdefault((int(par[rows[k]]), int(par[A["col"][k]])), []).append((int(rows[k]) << 32) | k)
        gotB = {}
        for r in range(B["n"]):
            cols = B["col"][B["rowptr"][r]:B["rowptr"][r + 1]]
            assert cols[0] == r
            for k in range(B["rowptr"][r], B["rowptr"][r + 1]):
                gotB[(r, int(B["col"][k]))] = B["g_ent"][B["g_ptr"][k]:B["g_ptr"][k + 1]].tolist()
        assert set(gotB) == set(wantB)
        for key in wantB:
            assert sorted(gotB[key]) == sorted(wantB[key])
        assert all((b, a) in gotB for (a, b) in gotB)                                             # structurally symmetric
    assert L[-1]["n"] <= dense_max and len(L[-1]["parent"]) == 0
    assert all(L[l + 1]["n"] < L[l]["n"] for l in range(len(L) - 1))


@pytest.mark.parametrize("level0_loops", [False, True], ids=["level1_along_odometry", "level1_along_all_edges"])
@pytest.mark.parametrize("n,loops,f,passes0,passes,dense_max", [(1500, 1500, 2, 3, 2, 64), (4000, 2500, 2, 2, 2, 100), (900, 100, 1, 3, 3, 16), (2500, 2500, 5, 1, 1, 200)])
def test_hierarchy_invariants(shim, n, loops, f, passes0, passes, dense_max, level0_loops):
    g = graphgen.generate(n, loops, odom_f_max=f, seed=n)
    free = np.ones(n, np.uint8)
    H = build(shim, g, free, passes0, passes, dense_max, level0_loops=level0_loops)
    assert H is not None and len(H["levels"]) >= 2
    check(g, H, free, passes0, passes, dense_max, level0_loops)


def test_fixed_keyframes_stay_outside_and_isolated_graphs_are_refused(shim):
    g = graphgen.generate(1200, 600, odom_f_max=2, seed=4)
    free = np.ones(1200, np.uint8); free[:300] = 0; free[700] = 0
    H = build(shim, g, free, 3, 2, 64, level0_loops=False)
    assert H is not None
    check(g, H, free, 3, 2, 64, False)
    # a "graph" without edges cannot coarsen: the builder says so instead of returning a useless hierarchy
    class E: pass
    e = E(); e.n_poses = 1000
    e.odom_c1 = e.odom_c2 = e.loop_c1 = e.loop_c2 = np.zeros(0, np.int32); e.odom_w = np.zeros(0)
    assert build(shim, e, np.ones(1000, np.uint8), 3, 2, 64) is None


def test_sharded_build_has_the_global_structure_and_every_contribution_exactly_once(shim):
    """Edge sharding (several ranks): every rank builds the hierarchy from the GLOBAL graph — identical structure on all of them — and lists only its own
    contributions to level 1 (its edges, the diagonal blocks of the keyframes it owns), in its rank-local numbering.  The union of the ranks' lists must be
    the single-handle list: every contribution exactly once."""
    lib = shim
    lib.mgh_build_sharded.restype = C.c_void_p
    g = graphgen.generate(3000, 900, odom_f_max=2, seed=5)
    N, world = g.n_poses, 3
    ref = build(lib, g, passes0=3, passes=2, dense_max=64, level0_loops=False)
    rng = np.random.default_rng(3)
    rank_rel = I32(rng.integers(0, world, size=g.n_odom)); rank_sw = I32(rng.integers(0, world, size=g.n_loops))
    touch = np.full((world, N), False)
    for r in range(world):
        for c, rk in ((g.odom_c1, rank_rel), (g.odom_c2, rank_rel), (g.loop_c1, rank_sw), (g.loop_c2, rank_sw)):
            touch[r, np.asarray(c)[rk == r]] = True
    owner = I32(np.argmax(touch, axis=0))                     # lowest touching rank (every keyframe is touched: the chain is complete)
    assert touch.any(axis=0).all()
    nf = np.ones(N, np.uint8)
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    seen = {}
    for rank in range(world):
        l2g = np.zeros(N, np.int32); rl2g = np.zeros(g.n_odom, np.int32); sl2g = np.zeros(g.n_loops, np.int32)
        nl, nr, ns = C.c_longlong(), C.c_longlong(), C.c_longlong()
        h = lib.mgh_build_sharded(C.c_longlong(N), ptr(nf, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)),
                                  ptr(sc1, C.c_int), ptr(sc2, C.c_int), 3, 2, 64, 32, 12, ptr(rank_rel, C.c_int), ptr(rank_sw, C.c_int), ptr(owner, C.c_int), rank,
                                  ptr(l2g, C.c_int), C.byref(nl), ptr(rl2g, C.c_int), C.byref(nr), ptr(sl2g, C.c_int), C.byref(ns))
        assert h
        h = C.c_void_p(h)
        assert lib.mgh_levels(h) == len(ref["levels"])
        for l, R in enumerate(ref["levels"]):
            sz = np.zeros(6, np.int64); lib.mgh_sizes(h, l, ptr(sz, C.c_longlong))
            n, nnzb, nent, npar, nagg, ntile = [int(x) for x in sz]
            L = dict(rowptr=np.zeros(n + 1, np.int64), col=np.zeros(nnzb, np.int32), g_ptr=np.zeros(nnzb + 1, np.int64), g_ent=np.zeros(nent, np.int64),
                     parent=np.zeros(npar, np.int32), agg_ptr=np.zeros(nagg, np.int32), tile_agg0=np.zeros(ntile, np.int32))
            lib.mgh_level(h, l, ptr(L["rowptr"], C.c_longlong), ptr(L["col"], C.c_int), ptr(L["g_ptr"], C.c_longlong), ptr(L["g_ent"], C.c_longlong), ptr(L["parent"], C.c_int),
                          ptr(L["agg_ptr"], C.c_int), ptr(L["tile_agg0"], C.c_int))
            # the structure is the global one on every rank
            assert n == R["n"] and np.array_equal(L["rowptr"], R["rowptr"]) and np.array_equal(L["col"], R["col"]) and np.array_equal(L["parent"], R["parent"])
            if l >= 1:
                assert np.array_equal(L["g_ent"], R["g_ent"])             # levels above 1 are replicated: the full lists
                continue
            for blk in range(nnzb):
                for ent in L["g_ent"][L["g_ptr"][blk]:L["g_ptr"][blk + 1]]:
                    kind, idx = int(ent & 7), int(ent >> 3)
                    gidx = int(l2g[idx]) if kind == 0 else int(rl2g[idx]) if kind <= 2 else int(sl2g[idx])      # rank-local -> global
                    key = (blk, kind, gidx)
                    assert key not in seen, key
                    seen[key] = rank
        lib.mgh_free(h)
    R = ref["levels"][0]
    want = set()
    for blk in range(len(R["col"])):
        for ent in R["g_ent"][R["g_ptr"][blk]:R["g_ptr"][blk + 1]]:
            want.add((blk, int(ent & 7), int(ent >> 3)))
    assert set(seen) == want
    # diagonal blocks come from the owner, edges from the rank that holds them
    for (blk, kind, gidx), rank in seen.items():
        assert rank == (owner[gidx] if kind == 0 else rank_rel[gidx] if kind <= 2 else rank_sw[gidx])


def test_smoothed_transition_structures_are_the_sparse_products(shim):
    """Smoothed aggregation on the transitions above level 1: Ps = (I - w D^-1 A) P has the pattern of A P, W = A Ps that of A A P, and the level above that of
    Ps^T W — checked against scipy's boolean products; the column-wise view of Ps lists every block exactly once; the diagonal block stays first in every row."""
    import scipy.sparse as sp
    g = graphgen.generate(4000, 1500, odom_f_max=2, seed=11)
    H = build(shim, g, passes0=3, passes=2, dense_max=40, level0_loops=False, smoothed=2)
    Href = build(shim, g, passes0=3, passes=2, dense_max=40, level0_loops=False, smoothed=0)
    assert len(H["levels"]) == len(Href["levels"]) >= 3
    for l, L in enumerate(H["levels"][:-1]):
        assert ("smoothed" in L) == (l < 2)
        if "smoothed" not in L:
            continue
        n, S = L["n"], L["smoothed"]
        nb = H["levels"][l + 1]["n"]
        rows = np.repeat(np.arange(n), np.diff(L["rowptr"]))
        A = sp.csr_matrix((np.ones(len(L["col"])), (rows, L["col"])), shape=(n, n))
        P = sp.csr_matrix((np.ones(n), (np.arange(n), L["parent"])), shape=(n, nb))
        def pattern(rowptr, col, shape):
            r = np.repeat(np.arange(shape[0]), np.diff(rowptr))
            return sp.csr_matrix((np.ones(len(col)), (r, col)), shape=shape)
        Ps = pattern(S["ps_rowptr"], S["ps_col"], (n, nb)); W = pattern(S["w_rowptr"], S["w_col"], (n, nb))
        assert ((A @ P) != 0).astype(int).toarray().tolist() == (Ps != 0).astype(int).toarray().tolist() if n <= 600 else (abs((A @ P).astype(bool).astype(int) - Ps.astype(bool).astype(int))).nnz == 0
        assert (abs((A @ Ps).astype(bool).astype(int) - W.astype(bool).astype(int))).nnz == 0
        B = H["levels"][l + 1]
        Bp = pattern(B["rowptr"], B["col"], (nb, nb))
        assert (abs((Ps.T @ W).astype(bool).astype(int) - Bp.astype(bool).astype(int))).nnz == 0
        assert all(B["col"][B["rowptr"][a]] == a for a in range(nb))
        # every row of Ps and W ascending (the numeric kernels search them), Ps by column complete
        for a in range(n):
            assert np.all(np.diff(S["ps_col"][S["ps_rowptr"][a]:S["ps_rowptr"][a + 1]]) > 0) and np.all(np.diff(S["w_col"][S["w_rowptr"][a]:S["w_rowptr"][a + 1]]) > 0)
        seen = np.zeros(len(S["ps_col"]), int)
        for a in range(nb):
            for ent in S["psT_ent"][S["psT_ptr"][a]:S["psT_ptr"][a + 1]]:
                i, sl = int(ent >> 32), int(ent & 0xffffffff)
                assert S["ps_rowptr"][i] <= sl < S["ps_rowptr"][i + 1] and S["ps_col"][sl] == a
                seen[sl] += 1
        assert np.all(seen == 1)
        # explicit transfer operator R^T = Ps - Dinv W on W's pattern: every W block knows its Ps block (same row, same column) or -1 — and every Ps block is found;
        # W's pattern by coarse row is a permutation of it (rT_of_w), rows of R ascend by fine row; the lane groups of the restriction kernel follow the row lengths
        w_rows = np.repeat(np.arange(n), np.diff(S["w_rowptr"]))
        hit = S["ps_of_w"] >= 0
        assert hit.sum() == len(S["ps_col"]) and len(set(S["ps_of_w"][hit].tolist())) == hit.sum()
        ps_rows = np.repeat(np.arange(n), np.diff(S["ps_rowptr"]))
        assert np.array_equal(ps_rows[S["ps_of_w"][hit]], w_rows[hit]) and np.array_equal(S["ps_col"][S["ps_of_w"][hit]], S["w_col"][hit])
        assert sorted(S["rT_of_w"].tolist()) == list(range(len(S["w_col"])))
        assert S["rT_rowptr"][0] == 0 and S["rT_rowptr"][-1] == len(S["w_col"]) and len(S["rT_rowptr"]) == nb + 1
        r_rows = np.repeat(np.arange(nb), np.diff(S["rT_rowptr"]))
        assert np.array_equal(r_rows[S["rT_of_w"]], S["w_col"]) and np.array_equal(S["rT_col"][S["rT_of_w"]], w_rows)
        for a in range(nb):
            assert np.all(np.diff(S["rT_col"][S["rT_rowptr"][a]:S["rT_rowptr"][a + 1]]) > 0)
        assert S["rT_seg"] in (1, 2, 4, 8) and (S["rT_seg"] == 8 or len(S["w_col"]) / nb <= 5.0 * S["rT_seg"])
        # the level above is denser than with the tentative prolongator, the levels' sizes are the same
        assert len(B["col"]) > len(Href["levels"][l + 1]["col"]) and B["n"] == Href["levels"][l + 1]["n"]


def test_smoothed_keyframe_transition_structures_are_the_sparse_products(shim):
    """Smoothed transition keyframes -> level 1: on the keyframe level's own block pattern (parallel edges repeat a column, fixed keyframes are outside the system) Ps has the
    pattern of A P, W that of A Ps, level 1 that of Ps^T W — without contribution lists, one hop wider than the aggregated edges —, the explicit operator's tables are consistent,
    and everything above level 1 is built from that wider pattern."""
    import scipy.sparse as sp
    g = graphgen.generate(3000, 1200, odom_f_max=2, seed=13)
    N = g.n_poses
    free = np.ones(N, np.uint8); free[0] = 0; free[1700] = 0
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    shim.mgh_build_fine.restype = C.c_void_p
    shim.mgh_fine_nnzb.restype = C.c_longlong
    h = C.c_void_p(shim.mgh_build_fine(C.c_longlong(N), ptr(free, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)),
                                       ptr(sc1, C.c_int), ptr(sc2, C.c_int), 3, 2, 40, 32, 12, 1, 64))
    assert h.value
    levels, agg0, mem0_ptr, mem0 = _read_hierarchy(shim, h, N)
    nnzb = shim.mgh_fine_nnzb(h)
    assert nnzb == N + 2 * (len(rc1) + len(sc1))
    rowptr = np.zeros(N + 1, np.int64); col = np.zeros(nnzb, np.int32)
    shim.mgh_fine_pattern(h, ptr(rowptr, C.c_longlong), ptr(col, C.c_int))
    ss = np.zeros(3, np.int64)
    shim.mgh_smoothed_sizes(h, -1, ptr(ss, C.c_longlong))
    n1 = levels[0]["n"]
    S = dict(ps_rowptr=np.zeros(N + 1, np.int32), ps_col=np.zeros(int(ss[0]), np.int32), w_rowptr=np.zeros(N + 1, np.int32), w_col=np.zeros(int(ss[1]), np.int32),
             psT_ptr=np.zeros(n1 + 1, np.int64), psT_ent=np.zeros(int(ss[2]), np.int64), ps_of_w=np.zeros(int(ss[1]), np.int32), rT_rowptr=np.zeros(n1 + 1, np.int32),
             rT_col=np.zeros(int(ss[1]), np.int32), rT_of_w=np.zeros(int(ss[1]), np.int32))
    shim.mgh_smoothed(h, -1, ptr(S["ps_rowptr"], C.c_int), ptr(S["ps_col"], C.c_int), ptr(S["w_rowptr"], C.c_int), ptr(S["w_col"], C.c_int), ptr(S["psT_ptr"], C.c_longlong), ptr(S["psT_ent"], C.c_longlong))
    seg = C.c_int(0)
    shim.mgh_explicit(h, -1, ptr(S["ps_of_w"], C.c_int), ptr(S["rT_rowptr"], C.c_int), ptr(S["rT_col"], C.c_int), ptr(S["rT_of_w"], C.c_int), C.byref(seg))
    shim.mgh_free(h)
    assert np.array_equal(agg0 >= 0, free.astype(bool))
    fr = np.nonzero(free)[0]
    rows = np.repeat(np.arange(N), np.diff(rowptr))
    keep = (free[rows] != 0) & (free[col] != 0)                      # the system: rows and columns of the free keyframes
    A = sp.csr_matrix((np.ones(int(keep.sum())), (rows[keep], col[keep])), shape=(N, N))
    P = sp.csr_matrix((np.ones(len(fr)), (fr, agg0[fr])), shape=(N, n1))
    def pattern(rp, cl, shape):
        r = np.repeat(np.arange(shape[0]), np.diff(rp))
        return sp.csr_matrix((np.ones(len(cl)), (r, cl)), shape=shape)
    def same(X, Y): return (abs(X.astype(bool).astype(int) - Y.astype(bool).astype(int))).nnz == 0
    Ps = pattern(S["ps_rowptr"], S["ps_col"], (N, n1)); W = pattern(S["w_rowptr"], S["w_col"], (N, n1))
    assert same(A @ P, Ps) and same(A @ Ps, W)
    for i in np.nonzero(free == 0)[0]:
        assert S["ps_rowptr"][i] == S["ps_rowptr"][i + 1] and S["w_rowptr"][i] == S["w_rowptr"][i + 1]
    B = levels[0]
    assert same(Ps.T @ W, pattern(B["rowptr"], B["col"], (n1, n1))) and all(B["col"][B["rowptr"][a]] == a for a in range(n1)) and len(B["g_ent"]) == 0
    for i in range(N):
        assert np.all(np.diff(S["ps_col"][S["ps_rowptr"][i]:S["ps_rowptr"][i + 1]]) > 0) and np.all(np.diff(S["w_col"][S["w_rowptr"][i]:S["w_rowptr"][i + 1]]) > 0)
    seen = np.zeros(len(S["ps_col"]), int)
    for a in range(n1):
        for ent in S["psT_ent"][S["psT_ptr"][a]:S["psT_ptr"][a + 1]]:
            i, sl = int(ent >> 32), int(ent & 0xffffffff)
            assert S["ps_rowptr"][i] <= sl < S["ps_rowptr"][i + 1] and S["ps_col"][sl] == a
            seen[sl] += 1
    assert np.all(seen == 1)
    w_rows = np.repeat(np.arange(N), np.diff(S["w_rowptr"])); ps_rows = np.repeat(np.arange(N), np.diff(S["ps_rowptr"]))
    hit = S["ps_of_w"] >= 0
    assert hit.sum() == len(S["ps_col"]) and np.array_equal(ps_rows[S["ps_of_w"][hit]], w_rows[hit]) and np.array_equal(S["ps_col"][S["ps_of_w"][hit]], S["w_col"][hit])
    assert sorted(S["rT_of_w"].tolist()) == list(range(len(S["w_col"])))
    r_rows = np.repeat(np.arange(n1), np.diff(S["rT_rowptr"]))
    assert np.array_equal(r_rows[S["rT_of_w"]], S["w_col"]) and np.array_equal(S["rT_col"][S["rT_of_w"]], w_rows)
    assert seg.value in (1, 2, 4, 8)
    # level 2 is the smoothed product of the WIDER level 1 (smoothed_levels = 1)
    L1 = levels[0]
    rows1 = np.repeat(np.arange(n1), np.diff(L1["rowptr"]))
    A1 = sp.csr_matrix((np.ones(len(L1["col"])), (rows1, L1["col"])), shape=(n1, n1))
    n2 = levels[1]["n"]
    P1 = sp.csr_matrix((np.ones(n1), (np.arange(n1), L1["parent"])), shape=(n1, n2))
    assert same(P1.T @ A1 @ A1 @ A1 @ P1, pattern(levels[1]["rowptr"], levels[1]["col"], (n2, n2)))
    print("blocks per keyframe: Ps %.2f  W %.2f ; level 1: %d nodes, %.1f blocks per row" % (len(S["ps_col"]) / N, len(S["w_col"]) / N, n1, len(B["col"]) / n1))


def _read_hierarchy(lib, h, N):
    levels = []
    for l in range(lib.mgh_levels(h)):
        sz = np.zeros(6, np.int64)
        lib.mgh_sizes(h, l, ptr(sz, C.c_longlong))
        n, nnzb, nent, npar, nagg, ntile = [int(x) for x in sz]
        L = dict(n=n, rowptr=np.zeros(n + 1, np.int64), col=np.zeros(nnzb, np.int32), g_ptr=np.zeros(nnzb + 1, np.int64), g_ent=np.zeros(nent, np.int64),
                 parent=np.zeros(npar, np.int32), agg_ptr=np.zeros(nagg, np.int32), tile_agg0=np.zeros(ntile, np.int32))
        lib.mgh_level(h, l, ptr(L["rowptr"], C.c_longlong), ptr(L["col"], C.c_int), ptr(L["g_ptr"], C.c_longlong), ptr(L["g_ent"], C.c_longlong), ptr(L["parent"], C.c_int),
                      ptr(L["agg_ptr"], C.c_int), ptr(L["tile_agg0"], C.c_int))
        levels.append(L)
    n1 = levels[0]["n"]
    agg0 = np.zeros(N, np.int32); mem0_ptr = np.zeros(n1 + 1, np.int32); mem0 = np.zeros(N, np.int32)
    lib.mgh_level0(h, ptr(agg0, C.c_int), ptr(mem0_ptr, C.c_int), ptr(mem0, C.c_int))
    return levels, agg0, mem0_ptr, mem0


@pytest.mark.parametrize("smoothed,discount,block", [(0, 0.0, 0), (1, 3.0, 64)])
def test_regroup_from_the_cache_equals_a_fresh_build(shim, smoothed, discount, block):
    """A regroup inside a solve rebuilds the hierarchy from the CURRENT switch values with the cached level-1 aggregation and level-1 structure (BuildCache): it must
    be exactly the hierarchy a fresh build with those switch values gives — and differ from the one built before the switches moved."""
    lib = shim
    lib.mgh_build_regroup.restype = C.c_void_p
    g = graphgen.generate(5000, 3000, odom_f_max=2, seed=13)
    N = g.n_poses
    nf = np.ones(N, np.uint8)
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    rng = np.random.default_rng(2)
    w_first = np.full(g.n_loops, 0.99 ** 2)
    w_now = np.where(rng.random(g.n_loops) < 0.2, 1e-6, rng.uniform(0.5, 1.0, g.n_loops))        # a fifth of the loop closures switched off
    def call(use_cache, wa, wb):
        h = lib.mgh_build_regroup(C.c_longlong(N), ptr(nf, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)), ptr(sc1, C.c_int),
                                  ptr(sc2, C.c_int), ptr(wa, C.c_double), ptr(wb, C.c_double), use_cache, 3, 2, 64, 32, 12, smoothed, C.c_double(discount), block)
        assert h
        h = C.c_void_p(h)
        out = _read_hierarchy(lib, h, N)
        lib.mgh_free(h)
        return out
    fresh = call(0, w_now, w_now)
    regrouped = call(1, w_first, w_now)
    before = call(0, w_first, w_first)
    (La, a0, mpa, ma), (Lb, b0, mpb, mb) = fresh, regrouped
    assert len(La) == len(Lb) and np.array_equal(a0, b0) and np.array_equal(mpa, mpb) and np.array_equal(ma, mb)
    for A, B in zip(La, Lb):
        for k in ("rowptr", "col", "g_ptr", "g_ent", "parent", "agg_ptr", "tile_agg0"):
            assert np.array_equal(A[k], B[k]), k
    # the switches mattered: the aggregates above level 1 changed
    assert not np.array_equal(before[0][0]["parent"], La[0]["parent"])
