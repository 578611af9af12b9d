"""CPU: the DISTRIBUTED multigrid's host side (csrc/pgo_mg_host.hpp, round 6) through tests/native/mg_host.cpp — aggregates that never mix owners, owner-major numbering,
tiles that never straddle ranks, and the neighbour-exchange plans: replayed here with numpy for every rank, a rank ends up holding exactly the rows its kernels read.
(The kernels themselves: tests/test_gpu_two_ranks_one_gpu.py, tests/test_gpu_c5.py; the same protocol over gloo: tests/test_sharding_gloo.py.)"""
import ctypes as C

import numpy as np
import pytest

from solve_keyframe_pose_graph_amd import graphgen, sharding
from tests.test_mg_hierarchy import I32, build, ptr, shim  # noqa: F401  (shim: the module-scoped fixture)


def touch_masks(g, world, policy, with_owner=False):
    """bit r of mask[k]: rank r holds a residual block on keyframe k (what libpgo's graph build all-reduces as sum of 2^rank); owner[k]: the rank holding most of k's residual
    blocks, the lowest such rank on a tie (its second all-reduce: max of (blocks + 1) * 64 + 63 - rank)"""
    parts = sharding.partition(g, world, policy)
    m = np.zeros(g.n_poses, np.uint64)
    best = np.zeros(g.n_poses, np.int64)
    for r, sel in enumerate(parts):
        io, il, ir = sel("odom", g.n_odom), sel("loop", g.n_loops), sel("reg", len(g.reg_node))
        deg = np.zeros(g.n_poses, np.int64)
        for a in (g.odom_c1[io], g.odom_c2[io], g.loop_c1[il], g.loop_c2[il], g.reg_node[ir]):
            np.add.at(deg, a, 1)
        t = deg > 0
        m[t] |= np.uint64(1 << r)
        best = np.maximum(best, np.where(t, (deg + 1) * 64 + 63 - r, 0))
    owner = np.where(best > 0, 63 - best % 64, -1).astype(np.int32)
    return (m, owner) if with_owner else m


def owner_of(g, world, policy):
    return touch_masks(g, world, policy, with_owner=True)[1]


def build_owned(lib, g, masks, world, dist_min_rows=64, dense_max=24, smoothed=1, passes0=3, passes=2, owner=None, policy="spatial"):
    N = g.n_poses
    owner = np.ascontiguousarray(owner_of(g, world, policy) if owner is None else owner, dtype=np.int32)
    nf = (masks != 0).astype(np.uint8)
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    mk = np.ascontiguousarray(masks, dtype=np.uint64)
    lib.mgh_build_owned.restype = C.c_void_p
    h = lib.mgh_build_owned(C.c_longlong(N), ptr(nf, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)), ptr(sc1, C.c_int),
                            ptr(sc2, C.c_int), passes0, passes, dense_max, 32, 12, smoothed, ptr(mk, C.c_ulonglong), ptr(owner, C.c_int), world, dist_min_rows)
    assert h, "the owner-constrained hierarchy did not coarsen"
    return C.c_void_p(h)


def levels_of(lib, h, world):
    out = []
    for l in range(lib.mgh_levels(h)):
        sz = np.zeros(6, np.int64)
        lib.mgh_sizes(h, l, ptr(sz, C.c_longlong))
        n, nnzb, nent, npar, nagg, ntile = [int(x) for x in sz]
        L = dict(n=n, rowptr=np.zeros(n + 1, np.int64), col=np.zeros(nnzb, np.int32), g_ptr=np.zeros(nnzb + 1, np.int64), g_ent=np.zeros(nent, np.int64),
                 parent=np.zeros(npar, np.int32), agg_ptr=np.zeros(nagg, np.int32), tile_agg0=np.zeros(ntile, np.int32))
        lib.mgh_level(h, l, ptr(L["rowptr"], C.c_longlong), ptr(L["col"], C.c_int), ptr(L["g_ptr"], C.c_longlong), ptr(L["g_ent"], C.c_longlong), ptr(L["parent"], C.c_int),
                      ptr(L["agg_ptr"], C.c_int), ptr(L["tile_agg0"], C.c_int))
        own_ptr = np.zeros(world + 1, np.int32); tile_ptr = np.zeros(world + 1, np.int32); d = C.c_int(0)
        lib.mgh_ownership(h, l, ptr(own_ptr, C.c_int), ptr(tile_ptr, C.c_int), C.byref(d))
        L.update(own_ptr=own_ptr, tile_ptr=tile_ptr, distributed=bool(d.value))
        ss = np.zeros(3, np.int64)
        lib.mgh_smoothed_sizes(h, l, ptr(ss, C.c_longlong))
        if ss[0] >= 0:
            S = dict(ps_rowptr=np.zeros(n + 1, np.int32), ps_col=np.zeros(int(ss[0]), np.int32), w_rowptr=np.zeros(n + 1, np.int32), w_col=np.zeros(int(ss[1]), np.int32),
                     psT_ptr=np.zeros(nagg, np.int64), psT_ent=np.zeros(int(ss[2]), np.int64))
            lib.mgh_smoothed(h, l, ptr(S["ps_rowptr"], C.c_int), ptr(S["ps_col"], C.c_int), ptr(S["w_rowptr"], C.c_int), ptr(S["w_col"], C.c_int), ptr(S["psT_ptr"], C.c_longlong), ptr(S["psT_ent"], C.c_longlong))
            S.update(ps_of_w=np.zeros(int(ss[1]), np.int32), rT_rowptr=np.zeros(nagg, np.int32), rT_col=np.zeros(int(ss[1]), np.int32), rT_of_w=np.zeros(int(ss[1]), np.int32))
            seg = C.c_int(0)
            lib.mgh_explicit(h, l, ptr(S["ps_of_w"], C.c_int), ptr(S["rT_rowptr"], C.c_int), ptr(S["rT_col"], C.c_int), ptr(S["rT_of_w"], C.c_int), C.byref(seg))
            L["smoothed"] = S
        out.append(L)
    return out


def plans_of(lib, h, masks, world, dist_min_rows, rank, n_levels):
    mk = np.ascontiguousarray(masks, dtype=np.uint64)
    lib.mgh_plans.restype = C.c_void_p
    ps = C.c_void_p(lib.mgh_plans(h, ptr(mk, C.c_ulonglong), C.c_longlong(len(mk)), world, dist_min_rows, rank))
    out = {}
    for l in [-1] + list(range(n_levels + 1)):      # -1: the keyframes' plan; n_levels: the prolongation's (level-1 rows)
        ns, nr = C.c_longlong(0), C.c_longlong(0)
        lib.mgh_plan_sizes(ps, l, C.byref(ns), C.byref(nr))
        P = dict(send_idx=np.zeros(ns.value, np.int32), recv_idx=np.zeros(nr.value if l >= 0 else 0, np.int32), send_off=np.zeros(world + 1, np.int64), recv_off=np.zeros(world + 1, np.int64),
                 pair_cnt=np.zeros(world * world, np.int64))
        lib.mgh_plan_get(ps, l, ptr(P["send_idx"], C.c_int), ptr(P["send_off"], C.c_longlong), ptr(P["recv_idx"], C.c_int), ptr(P["recv_off"], C.c_longlong), ptr(P["pair_cnt"], C.c_longlong))
        out[l] = P
    nsh, nsrc = C.c_longlong(0), C.c_longlong(0)
    lib.mgh_fine_sizes(ps, C.byref(nsh), C.byref(nsrc))
    F = dict(sh_loc=np.zeros(nsh.value, np.int32), sum_ptr=np.zeros(nsh.value + 1, np.int32), sum_src=np.zeros(nsrc.value, np.int32))
    lib.mgh_fine_get(ps, ptr(F["sh_loc"], C.c_int), ptr(F["sum_ptr"], C.c_int), ptr(F["sum_src"], C.c_int))
    out["fine"] = F
    lib.mgh_plans_free(ps)
    return out


def owner_of_rows(L, world):
    o = np.zeros(L["n"], np.int32)
    for r in range(world):
        o[L["own_ptr"][r]:L["own_ptr"][r + 1]] = r
    return o


@pytest.mark.parametrize("world,policy,smoothed", [(2, "chain", 1), (4, "spatial", 1), (3, "spatial", 0), (8, "spatial", 0)])
def test_aggregates_never_mix_owners_and_every_level_is_owner_major(shim, world, policy, smoothed):
    g = graphgen.generate(6000, 3000, odom_f_max=2, seed=7)
    masks, kf_owner = touch_masks(g, world, policy, with_owner=True)
    h = build_owned(shim, g, masks, world, smoothed=smoothed, owner=kf_owner)
    assert shim.mgh_world(h) == world
    L = levels_of(shim, h, world)
    agg0 = np.zeros(g.n_poses, np.int32); mem0_ptr = np.zeros(L[0]["n"] + 1, np.int32); mem0 = np.zeros(int((masks != 0).sum()), np.int32)
    shim.mgh_level0(h, ptr(agg0, C.c_int), ptr(mem0_ptr, C.c_int), ptr(mem0, C.c_int))
    assert np.all((masks[kf_owner >= 0] >> kf_owner[kf_owner >= 0].astype(np.uint64)) & np.uint64(1))      # the owner is one of the touching ranks
    own = [owner_of_rows(A, world) for A in L]
    # level 1: every aggregate holds keyframes of ONE owner, and that rank owns the aggregate
    assert np.array_equal(own[0][agg0[agg0 >= 0]], kf_owner[agg0 >= 0])
    for l, A in enumerate(L):
        assert A["own_ptr"][0] == 0 and A["own_ptr"][-1] == A["n"] and np.all(np.diff(A["own_ptr"]) >= 0)
        if l + 1 < len(L):
            assert np.array_equal(own[l + 1][A["parent"]], own[l])                   # parents inherit the owner: the numbering is owner-major on every level
            ta = A["tile_agg0"]
            assert A["tile_ptr"][0] == 0 and A["tile_ptr"][-1] == len(ta) - 1
            up = L[l + 1]["own_ptr"]
            for r in range(world):                                                     # a rank's tiles hold its own aggregates, all of them, nothing else
                t0, t1 = A["tile_ptr"][r], A["tile_ptr"][r + 1]
                assert (ta[t0] == up[r] and ta[t1] == up[r + 1]) or up[r] == up[r + 1]
            rows_per_tile = A["agg_ptr"][ta[1:]] - A["agg_ptr"][ta[:-1]]
            assert rows_per_tile.max() <= 32
    assert any(A["distributed"] for A in L[:-1]) and not L[-1]["distributed"]
    shim.mgh_free(h)


@pytest.mark.parametrize("world,policy,smoothed,dist_min", [(2, "chain", 1, 64), (4, "spatial", 1, 64), (3, "spatial", 0, 64), (4, "spatial", 1, 400), (8, "spatial", 0, 16)])
def test_exchange_plans_bring_every_row_a_rank_reads(shim, world, policy, smoothed, dist_min):
    """The cycle's exchanges replayed with numpy: every rank holds level vectors that are valid on its own rows only (NaN elsewhere); after the exchange of a level's plan the
    entries its kernels read are there — the columns of its rows, the rows of R its coarse rows restrict from, the columns of R^T one level up, the aggregates of the keyframes it
    touches; on a level every rank runs completely: everything.  Sender and receiver agree on every segment without a handshake."""
    g = graphgen.generate(6000, 3000, odom_f_max=2, seed=7)
    masks = touch_masks(g, world, policy)
    h = build_owned(shim, g, masks, world, dist_min_rows=dist_min, smoothed=smoothed, policy=policy)
    L = levels_of(shim, h, world)
    nl = len(L)
    agg0 = np.zeros(g.n_poses, np.int32); mem0_ptr = np.zeros(L[0]["n"] + 1, np.int32); mem0 = np.zeros(int((masks != 0).sum()), np.int32)
    shim.mgh_level0(h, ptr(agg0, C.c_int), ptr(mem0_ptr, C.c_int), ptr(mem0, C.c_int))
    P = [plans_of(shim, h, masks, world, dist_min, r, nl) for r in range(world)]
    own = [owner_of_rows(A, world) for A in L]
    for l, A in enumerate(L):
        truth = np.arange(A["n"], dtype=np.float64) * 1.5 + 7.0 + 100.0 * l
        vec = [np.where(own[l] == r, truth, np.nan) for r in range(world)]
        for r in range(world):                                                    # the same pair counts on every rank; a segment's two ends agree
            assert np.array_equal(P[r][l]["pair_cnt"], P[0][l]["pair_cnt"])
            for q in range(world):
                ns = P[r][l]["send_off"][q + 1] - P[r][l]["send_off"][q]
                assert ns == P[q][l]["recv_off"][r + 1] - P[q][l]["recv_off"][r] == P[0][l]["pair_cnt"][r * world + q]
                seg = P[r][l]["send_idx"][P[r][l]["send_off"][q]:P[r][l]["send_off"][q + 1]]
                assert np.all(own[l][seg] == r) and np.all(np.diff(seg) > 0)          # a rank sends rows it owns, ascending
                dst = P[q][l]["recv_idx"][P[q][l]["recv_off"][r]:P[q][l]["recv_off"][r + 1]]
                assert np.array_equal(seg, dst)
                vec[q][dst] = vec[r][seg]
        for r in range(world):
            rows = np.nonzero(own[l] == r)[0]
            if not A["distributed"]:
                assert np.array_equal(vec[r], truth)                              # a level every rank runs completely: gathered
                continue
            need = [A["col"][A["rowptr"][i]:A["rowptr"][i + 1]] for i in rows]
            if "smoothed" in A:
                S = A["smoothed"]
                crow = np.nonzero(own[l + 1] == r)[0]
                need += [S["rT_col"][S["rT_rowptr"][c]:S["rT_rowptr"][c + 1]] for c in crow]
            if l > 0 and "smoothed" in L[l - 1] and L[l - 1]["distributed"]:
                S = L[l - 1]["smoothed"]
                below = np.nonzero(own[l - 1] == r)[0]
                need += [S["w_col"][S["w_rowptr"][i]:S["w_rowptr"][i + 1]] for i in below]
            need = np.unique(np.concatenate(need)) if need else np.zeros(0, np.int64)
            assert np.array_equal(vec[r][need], truth[need]), (l, r)
            got = np.nonzero(~np.isnan(vec[r]))[0]
            assert np.array_equal(got, np.union1d(rows, need))                   # ... and nothing else travelled
    # the prolongation's own plan: x of level 1 at the aggregates of every keyframe a rank touches
    if L[0]["distributed"]:
        A = L[0]
        truth = np.arange(A["n"], dtype=np.float64) + 0.25
        vec = [np.where(own[0] == r, truth, np.nan) for r in range(world)]
        for r in range(world):
            for q in range(world):
                seg = P[r][nl]["send_idx"][P[r][nl]["send_off"][q]:P[r][nl]["send_off"][q + 1]]
                dst = P[q][nl]["recv_idx"][P[q][nl]["recv_off"][r]:P[q][nl]["recv_off"][r + 1]]
                assert np.array_equal(seg, dst) and np.all(own[0][seg] == r)
                vec[q][dst] = vec[r][seg]
        for r in range(world):
            touched = np.nonzero((masks >> np.uint64(r)) & np.uint64(1))[0]
            need = np.unique(agg0[touched][agg0[touched] >= 0])
            assert np.array_equal(vec[r][need], truth[need])
            assert np.array_equal(np.nonzero(~np.isnan(vec[r]))[0], np.union1d(np.nonzero(own[0] == r)[0], need))
            assert len(P[r][nl]["send_idx"]) <= len(P[r][0]["send_idx"]) or "smoothed" not in A
    # distributed levels send far less than a gather would
    l1 = 0
    if L[l1]["distributed"]:
        assert P[0][l1]["pair_cnt"].sum() < 0.5 * (world - 1) * L[l1]["n"]
    shim.mgh_free(h)


@pytest.mark.parametrize("world,policy", [(2, "contiguous"), (3, "spatial"), (4, "chain")])
def test_keyframe_plan_sums_the_partial_rows_in_rank_order(shim, world, policy):
    """The keyframes' own exchange: every rank holds a PARTIAL row of each keyframe it touches; after the exchange + the ordered sums every touching rank holds the total, bit for
    bit the same number on all of them (the parts are added in ascending rank order everywhere)."""
    g = graphgen.generate(3000, 1500, odom_f_max=2, seed=5)
    masks = touch_masks(g, world, policy)
    h = build_owned(shim, g, masks, world, smoothed=0, policy=policy)
    nl = shim.mgh_levels(h)
    P = [plans_of(shim, h, masks, world, 64, r, nl) for r in range(world)]
    rng = np.random.default_rng(0)
    part = rng.normal(size=(world, g.n_poses))                 # rank r's partial value for keyframe k (used where r touches k)
    l2g = [np.nonzero((masks >> np.uint64(r)) & np.uint64(1))[0] for r in range(world)]
    local = [part[r][l2g[r]].copy() for r in range(world)]
    recv = []
    for r in range(world):
        X = P[r][-1]
        buf = np.full(int(X["recv_off"][-1]), np.nan)
        for q in range(world):
            Xq = P[q][-1]
            seg = Xq["send_idx"][Xq["send_off"][r]:Xq["send_off"][r + 1]]            # LOCAL ids on rank q
            assert len(seg) == X["recv_off"][q + 1] - X["recv_off"][q]
            buf[X["recv_off"][q]:X["recv_off"][q + 1]] = local[q][seg]
            mine = X["send_idx"][X["send_off"][q]:X["send_off"][q + 1]]
            assert np.array_equal(l2g[q][seg], l2g[r][mine])                          # both ends list the same keyframes in the same order
        recv.append(buf)
    total = {}
    for r in range(world):
        F = P[r]["fine"]
        cnt = np.array([bin(int(m)).count("1") for m in masks[l2g[r]]])
        assert np.array_equal(F["sh_loc"], np.nonzero(cnt >= 2)[0])
        for j, l in enumerate(F["sh_loc"]):
            s = 0.0
            for e in range(F["sum_ptr"][j], F["sum_ptr"][j + 1]):
                src = F["sum_src"][e]
                s += local[r][l] if src < 0 else recv[r][src]
            k = int(l2g[r][l])
            ranks = [q for q in range(world) if (int(masks[k]) >> q) & 1]
            ref = 0.0
            for q in ranks:
                ref += part[q][k]
            assert s == ref                                                            # the same order of additions: the same bits
            total.setdefault(k, s)
            assert total[k] == s
    shim.mgh_free(h)


def setup_plans_of(lib, h, g, world, policy, rank, L):
    """pgo_mg_host.hpp: build_setup_plans for one rank, from the edge lists gathered rank by rank (as libpgo's graph build gathers them)"""
    parts = sharding.partition(g, world, policy)
    io = [sel("odom", g.n_odom) for sel in parts]; il = [sel("loop", g.n_loops) for sel in parts]
    rc1, rc2 = I32(np.concatenate([g.odom_c1[i] for i in io])), I32(np.concatenate([g.odom_c2[i] for i in io]))
    sc1, sc2 = I32(np.concatenate([g.loop_c1[i] for i in il])), I32(np.concatenate([g.loop_c2[i] for i in il]))
    ro = np.concatenate([[0], np.cumsum([len(i) for i in io])]).astype(np.int64); so = np.concatenate([[0], np.cumsum([len(i) for i in il])]).astype(np.int64)
    lib.mgh_setup_plans.restype = C.c_void_p
    sp = C.c_void_p(lib.mgh_setup_plans(h, rank, world, C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(ro, C.c_longlong), C.c_longlong(len(sc1)), ptr(sc1, C.c_int), ptr(sc2, C.c_int),
                                        ptr(so, C.c_longlong)))
    fw = lib.mgh_setup_first_whole(sp)
    out = dict(first_whole=fw, val=[], ps=[], rv=[], prod=[])
    lib.mgh_setup_prod_size.restype = C.c_longlong

    def get(kind, l):
        ns, nr, nd, nsrc = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        lib.mgh_setup_sizes(sp, kind, l, C.byref(ns), C.byref(nr), C.byref(nd), C.byref(nsrc))
        P = dict(send_idx=np.zeros(ns.value, np.int32), recv_idx=np.zeros(nr.value if kind else 0, np.int32), send_off=np.zeros(world + 1, np.int64), recv_off=np.zeros(world + 1, np.int64),
                 pair_cnt=np.zeros(world * world, np.int64), dst=np.zeros(nd.value, np.int32), sum_ptr=np.zeros(nd.value + 1, np.int32), sum_src=np.zeros(nsrc.value, np.int32))
        lib.mgh_setup_get(sp, kind, l, ptr(P["send_idx"], C.c_int), ptr(P["send_off"], C.c_longlong), ptr(P["recv_idx"], C.c_int), ptr(P["recv_off"], C.c_longlong), ptr(P["pair_cnt"], C.c_longlong),
                          ptr(P["dst"], C.c_int), ptr(P["sum_ptr"], C.c_int), ptr(P["sum_src"], C.c_int))
        return P
    for l in range(fw + 1 if fw > 0 else 0):
        out["val"].append(get(0, l))
    for l in range(fw):
        sm = "smoothed" in L[l]
        out["ps"].append(get(1, l) if sm else None); out["rv"].append(get(2, l) if sm else None)
        n = lib.mgh_setup_prod_size(sp, l)
        pr = np.zeros(int(n), np.int32)
        lib.mgh_setup_prod_get(sp, l, ptr(pr, C.c_int))
        out["prod"].append(pr)
    lib.mgh_setup_free(sp)
    return out, (rc1, rc2, ro, sc1, sc2, so)


def exchange_copy(world, plans, arrays):
    """neighbour exchange with ONE producer per entry: plans[r] / arrays[r] of every rank; segment (r -> q) must be the same list on both ends"""
    for r in range(world):
        assert np.array_equal(plans[r]["pair_cnt"], plans[0]["pair_cnt"])
    staged = [[arrays[r][plans[r]["send_idx"][plans[r]["send_off"][q]:plans[r]["send_off"][q + 1]]].copy() for q in range(world)] for r in range(world)]
    for r in range(world):
        for q in range(world):
            seg = plans[r]["send_idx"][plans[r]["send_off"][q]:plans[r]["send_off"][q + 1]]
            dst = plans[q]["recv_idx"][plans[q]["recv_off"][r]:plans[q]["recv_off"][r + 1]]
            assert np.array_equal(seg, dst) and len(seg) == plans[0]["pair_cnt"][r * world + q]
            arrays[q][dst] = staged[r][q]


def exchange_sum(world, plans, arrays):
    """... with several contributors per entry: every needer adds the parts in ascending rank order (its own where its rank comes)"""
    for r in range(world):
        assert np.array_equal(plans[r]["pair_cnt"], plans[0]["pair_cnt"])
    recv = []
    for q in range(world):
        buf = np.full(int(plans[q]["recv_off"][-1]), np.nan)
        for r in range(world):
            seg = plans[r]["send_idx"][plans[r]["send_off"][q]:plans[r]["send_off"][q + 1]]
            assert len(seg) == plans[q]["recv_off"][r + 1] - plans[q]["recv_off"][r] == plans[0]["pair_cnt"][r * world + q] and np.all(np.diff(seg) > 0)
            buf[plans[q]["recv_off"][r]:plans[q]["recv_off"][r + 1]] = arrays[r][seg]
        recv.append(buf)
    for q in range(world):
        P = plans[q]
        for j, k in enumerate(P["dst"]):
            src = P["sum_src"][P["sum_ptr"][j]:P["sum_ptr"][j + 1]]
            tot = 0.0
            for s_ in src:
                tot += arrays[q][k] if s_ < 0 else recv[q][s_]
            arrays[q][k] = tot


@pytest.mark.parametrize("world,policy,smoothed,dist_min", [(2, "chain", 1, 64), (4, "spatial", 1, 64), (3, "spatial", 0, 64), (4, "spatial", 2, 32), (8, "spatial", 0, 16), (3, "spatial", 1, 100000)])
def test_distributed_setup_replayed_with_scalar_blocks(shim, world, policy, smoothed, dist_min):
    """The distributed SET-UP (pgo_solver.hip: build_mg_ranks) replayed with numpy, every 6x6 block a scalar: each rank holds arrays that are NaN wherever it has not formed or
    received a number, forms what libpgo's kernels form on its own rows — level 1 from its own edges and owned keyframes, Dinv, Ps, W, R, the Galerkin products — from exactly the
    entries those kernels read, and exchanges blocks by the plans of pgo_mg_host.hpp: build_setup_plans.  Every entry its cycle kernels read must then equal the single-process
    result: its rows' blocks (and the transposed upper blocks of other owners: the fp32 copy is symmetrised), Dinv on its rows and their halo, R^T on its rows, R on its coarse
    rows, everything of the first level every rank runs completely."""
    g = graphgen.generate(2500, 1500, odom_f_max=2, seed=11)
    masks, kf_owner = touch_masks(g, world, policy, with_owner=True)
    h = build_owned(shim, g, masks, world, dist_min_rows=dist_min, smoothed=smoothed, owner=kf_owner, policy=policy)
    L = levels_of(shim, h, world)
    nl = len(L)
    agg0 = np.zeros(g.n_poses, np.int32); mem0_ptr = np.zeros(L[0]["n"] + 1, np.int32); mem0 = np.zeros(int((masks != 0).sum()), np.int32)
    shim.mgh_level0(h, ptr(agg0, C.c_int), ptr(mem0_ptr, C.c_int), ptr(mem0, C.c_int))
    SP, edges = zip(*[setup_plans_of(shim, h, g, world, policy, r, L) for r in range(world)])
    rc1, rc2, ro, sc1, sc2, so = edges[0]
    fw = SP[0]["first_whole"]
    assert all(S["first_whole"] == fw for S in SP)
    if not L[0]["distributed"]:
        assert fw == 0      # level 1 not distributed: the set-up stays replicated
        shim.mgh_free(h)
        return
    assert fw >= 1 and all(L[l]["distributed"] for l in range(fw)) and not L[fw]["distributed"]
    LP = [plans_of(shim, h, masks, world, dist_min, r, nl) for r in range(world)]
    own = [owner_of_rows(A, world) for A in L]
    rng = np.random.default_rng(5)

    def slot_of(A, a, b):
        if a == b:
            return int(A["rowptr"][a])
        lo, hi = int(A["rowptr"][a]) + 1, int(A["rowptr"][a + 1])
        k = lo + int(np.searchsorted(A["col"][lo:hi], b))
        assert k < hi and A["col"][k] == b
        return k
    row_of = [np.repeat(np.arange(A["n"]), np.diff(A["rowptr"])) for A in L]
    # ---- level 1: the truth and every rank's part (a keyframe's diagonal block by its owner, an edge's two blocks by the rank holding the edge)
    A0 = L[0]
    truth = [np.zeros(len(A["col"])) for A in L]
    part = [np.full(len(A0["col"]), np.nan) for _ in range(world)]

    def add(r, k, v):
        truth[0][k] += v
        part[r][k] = v if np.isnan(part[r][k]) else part[r][k] + v
    dk = 5.0 + rng.random(g.n_poses)
    for i in range(g.n_poses):
        if agg0[i] >= 0:
            add(int(kf_owner[i]), slot_of(A0, agg0[i], agg0[i]), dk[i])
    for (c1, c2, off) in ((rc1, rc2, ro), (sc1, sc2, so)):
        he = -0.3 * rng.random(len(c1))
        for r in range(world):
            for e in range(int(off[r]), int(off[r + 1])):
                a, b = agg0[c1[e]], agg0[c2[e]]
                if a < 0 or b < 0:
                    continue
                add(r, slot_of(A0, a, b), he[e]); add(r, slot_of(A0, b, a), he[e])
    val = [part]      # val[l][r]: rank r's array of level l
    exchange_sum(world, [S["val"][0] for S in SP], val[0])
    cs = 0.6
    Dinv_t, Ps_t, W_t, R_t = {}, {}, {}, {}
    for l in range(fw + 1):
        A = L[l]
        # what a rank's kernels read of this level's blocks: its rows; the blocks above the diagonal whose column it owns (transposed into its fp32 copy); the first whole level: all
        for r in range(world):
            blocks = np.arange(len(A["col"]))
            need = blocks if l == fw else blocks[(own[l][row_of[l]] == r) | ((A["col"] > row_of[l]) & (own[l][A["col"]] == r))]
            assert np.allclose(val[l][r][need], truth[l][need], rtol=1e-13, atol=0), (l, r)
        if l == fw:
            break
        B = L[l + 1]
        diag = A["rowptr"][:-1]
        Dinv_t[l] = 1.0 / truth[l][diag]
        Dinv = [np.where(own[l] == r, 1.0 / val[l][r][diag], np.nan) for r in range(world)]
        nxt = [np.full(len(B["col"]), np.nan) for _ in range(world)]
        if "smoothed" in A:
            S = A["smoothed"]
            exchange_copy(world, [LP[r][l] for r in range(world)], Dinv)       # the halo's Dinv (the cycle forms x = Dinv r on receipt): the level's own plan
            ps_row = np.repeat(np.arange(A["n"]), np.diff(S["ps_rowptr"])); w_row = np.repeat(np.arange(A["n"]), np.diff(S["w_rowptr"]))

            def form_ps(valA, Dv, rows):
                out = {}
                for i in rows:
                    for pk in range(S["ps_rowptr"][i], S["ps_rowptr"][i + 1]):
                        a = S["ps_col"][pk]
                        acc = sum(valA[k] for k in range(A["rowptr"][i], A["rowptr"][i + 1]) if A["parent"][A["col"][k]] == a)
                        out[pk] = (1.0 if A["parent"][i] == a else 0.0) - cs * Dv[i] * acc
                return out

            def form_w(valA, Ps, rows):
                out = {}
                for i in rows:
                    for wk in range(S["w_rowptr"][i], S["w_rowptr"][i + 1]):
                        b = S["w_col"][wk]
                        acc = 0.0
                        for k in range(A["rowptr"][i], A["rowptr"][i + 1]):
                            j = A["col"][k]
                            for pk in range(S["ps_rowptr"][j], S["ps_rowptr"][j + 1]):
                                if S["ps_col"][pk] == b:
                                    acc += valA[k] * Ps[pk]
                        out[wk] = acc
                return out
            allrows = range(A["n"])
            pst = form_ps(truth[l], Dinv_t[l], allrows); Ps_t[l] = np.array([pst[k] for k in range(len(S["ps_col"]))])
            wt = form_w(truth[l], Ps_t[l], allrows); W_t[l] = np.array([wt[k] for k in range(len(S["w_col"]))])
            Ps = [np.full(len(S["ps_col"]), np.nan) for _ in range(world)]
            for r in range(world):
                for k, v in form_ps(val[l][r], Dinv[r], np.nonzero(own[l] == r)[0]).items():
                    Ps[r][k] = v
            exchange_copy(world, [SP[r]["ps"][l] for r in range(world)], Ps)
            W = [np.full(len(S["w_col"]), np.nan) for _ in range(world)]
            Rv = [np.full(len(S["w_col"]), np.nan) for _ in range(world)]      # R by coarse row (slot rT_of_w[k] for block k of W)
            R_t[l] = np.zeros(len(S["w_col"]))
            for k in range(len(S["w_col"])):
                ps = S["ps_of_w"][k]
                R_t[l][S["rT_of_w"][k]] = (Ps_t[l][ps] if ps >= 0 else 0.0) - Dinv_t[l][w_row[k]] * W_t[l][k]
            for r in range(world):
                for k, v in form_w(val[l][r], Ps[r], np.nonzero(own[l] == r)[0]).items():
                    W[r][k] = v
                    ps = S["ps_of_w"][k]
                    Rv[r][S["rT_of_w"][k]] = (Ps[r][ps] if ps >= 0 else 0.0) - Dinv[r][w_row[k]] * v
                assert not np.isnan(W[r][own[l][w_row] == r]).any()                                   # every entry the W kernel read was there
            exchange_copy(world, [SP[r]["rv"][l] for r in range(world)], Rv)
            for r in range(world):
                c0, c1_ = L[l + 1]["own_ptr"][r], L[l + 1]["own_ptr"][r + 1]
                mine = np.arange(S["rT_rowptr"][c0], S["rT_rowptr"][c1_])
                assert np.allclose(Rv[r][mine], R_t[l][mine], rtol=1e-12, atol=1e-15), (l, r)          # R on the rank's coarse rows
                halo = np.unique(np.concatenate([A["col"][A["rowptr"][i]:A["rowptr"][i + 1]] for i in np.nonzero(own[l] == r)[0]] + [S["rT_col"][mine]]))
                assert np.allclose(Dinv[r][halo], Dinv_t[l][halo], rtol=1e-13), (l, r)                 # Dinv wherever the cycle forms x = Dinv r on receipt
            # the product Ps^T W: the truth, and every rank's rows' part on the blocks it contributes to
            for i in range(A["n"]):
                for pk in range(S["ps_rowptr"][i], S["ps_rowptr"][i + 1]):
                    for wk in range(S["w_rowptr"][i], S["w_rowptr"][i + 1]):
                        kb = slot_of(B, S["ps_col"][pk], S["w_col"][wk])
                        truth[l + 1][kb] += Ps_t[l][pk] * W_t[l][wk]
                        r = own[l][i]
                        v = Ps[r][pk] * W[r][wk]
                        nxt[r][kb] = v if np.isnan(nxt[r][kb]) else nxt[r][kb] + v
            for r in range(world):
                assert np.array_equal(np.nonzero(~np.isnan(nxt[r]))[0], SP[r]["prod"][l])              # exactly the blocks the library launches the product on
        else:
            for kb in range(len(B["col"])):
                ent = B["g_ent"][B["g_ptr"][kb]:B["g_ptr"][kb + 1]]
                truth[l + 1][kb] = sum(truth[l][int(e) & 0xffffffff] for e in ent)
                r = own[l + 1][row_of[l + 1][kb]]
                rows = (ent >> 32).astype(np.int64)
                assert np.all(own[l][rows] == r)                                                       # a coarse row's contributions are blocks of its own rank's rows
                nxt[r][kb] = sum(val[l][r][int(e) & 0xffffffff] for e in ent)
        val.append(nxt)
        exchange_sum(world, [S_["val"][l + 1] for S_ in SP], val[l + 1])
    # what travels: far less than level 1's all-reduce (every block, from every rank)
    sent = sum(int(S["val"][0]["pair_cnt"].sum()) for S in SP[:1])
    assert sent < 0.5 * len(A0["col"]) * world
    shim.mgh_free(h)
