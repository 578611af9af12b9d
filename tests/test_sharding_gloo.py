"""CPU, world_size 2 and 3 over gloo: the construction of the N > 1 path.  Each rank owns the edges a sharding.partition policy deals
it and works on the keyframes those edges touch (rank-local subgraph); keyframes touched by >= 2 ranks are shared and ONLY their rows
travel.  With the oracle standing in for the kernels (no GPU in this container) the test replays the exchanges libpgo issues (round 6:
NEIGHBOUR exchanges) — the touch masks at graph build (ONE all-reduce of sum 2^rank: who touches a keyframe, the lowest of them owns it), the
partial rows of diagonal + gradient per linearisation and of the CG matvec output sent between the ranks that share a keyframe (send / recv)
and summed in ascending rank order, BOTH dot products of the iteration in one 2-double all-reduce (the rank-local partial of u.Au and the
owner-weighted partial of r.u: the Chronopoulos-Gear form needs no other collective), owner-wise write-back — and checks every rank ends up
with the single-rank numbers on ITS keyframes.  test_level_halo_protocol_over_gloo replays the distributed multigrid's level exchanges with the
plans csrc/pgo_mg_host.hpp builds (through tests/native/mg_host.cpp): after the sends / receives a rank holds every row its level kernels read."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from solve_keyframe_pose_graph_amd import sharding
from solve_keyframe_pose_graph_amd.sharding import edge_slice
from tests import util


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _local_problem(g, sel):
    from oracle import binding as ob
    P = ob.OracleProblem()
    io, il, ir = sel("odom", g.n_odom), sel("loop", g.n_loops), sel("reg", len(g.reg_node))
    if len(io):
        P.add_relpose_edges(g.odom_c1[io], g.odom_c2[io], g.odom_T[io], g.odom_w[io])
    if len(il):
        P.add_switchable_edges(g.loop_c1[il], g.loop_c2[il], g.loop_T[il], g.loop_w[il], il)
    if len(ir):
        P.set_node_regularizers(g.reg_node[ir], g.reg_T[ir], g.reg_w[ir])
    touched = np.zeros(g.n_poses, bool)
    for a in (g.odom_c1[io], g.odom_c2[io], g.loop_c1[il], g.loop_c2[il], g.reg_node[ir]):
        touched[a] = True
    return P, il, touched


def _reduced_operator(H, N, owned_sw, radius, x):
    """(H_pp_local - sum_{owned switches} c c^T / a) x : the rank's part A_r x of the Schur-reduced operator, without the damping"""
    Hpp = H[:6 * N, :6 * N]
    y = Hpp @ x
    for e in owned_sw:
        c = H[:6 * N, 6 * N + e]
        hss = H[6 * N + e, 6 * N + e]
        sc = 1.0 / (1.0 + np.sqrt(hss))
        lam_s = min(max(sc * sc * hss, 1e-6), 1e32) / (radius * sc * sc)
        y -= c * (c @ x) / (hss + lam_s)
    return y


def _exchange_rows(rows_mine, mine_shared, masks, rank, world, extra=None):
    """libpgo's exchange_rows (round 6): this rank's partial rows of the keyframes it shares with peer q go to q (the keyframes both touch, ascending — both ends derive the same
    list from the touch masks), the parts are summed in ascending rank order; the iteration's scalars by one small all-reduce"""
    rows_mine = np.ascontiguousarray(rows_mine)
    segs, reqs, bufs = {}, [], {}
    for q in range(world):
        if q == rank:
            continue
        sel = np.nonzero((masks[mine_shared] >> np.uint64(q)) & np.uint64(1))[0]
        segs[q] = sel
        if len(sel) == 0:
            continue
        bufs[q] = torch.zeros(len(sel), rows_mine.shape[1], dtype=torch.float64)
        reqs.append(dist.isend(torch.from_numpy(rows_mine[sel].copy()), dst=q))
        reqs.append(dist.irecv(bufs[q], src=q))
    for r in reqs:
        r.wait()
    total = np.zeros_like(rows_mine)
    for q in range(world):                                          # ascending rank order, the rank's own part where its rank comes: the same bits on every rank
        if q == rank:
            total += rows_mine
        elif len(segs[q]):
            total[segs[q]] += bufs[q].numpy()
    if extra is None:
        return total, None
    ex = torch.from_numpy(np.asarray(extra, dtype=np.float64).copy()); dist.all_reduce(ex)
    return total, ex.numpy()


def _worker(rank, world, port, policy, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = util.small_graph(90, 18, f=2, seed=6)
    q, t, s = util.initial_state(g, True, perturb=0.02, seed=3)
    N, S = g.n_poses, g.n_loops
    P, owned_sw, touched = _local_problem(g, sharding.partition(g, world, policy)[rank])
    cost, _, grad = P.evaluate(q, t, s, want_residuals=False)
    H = P.dense_normal_matrix(q, t, s)
    # ---- graph build: ONE all-reduce of sum 2^rank over the touching ranks: who they are, how many, the lowest of them (the owner)
    mk = torch.from_numpy(np.where(touched, float(1 << rank), 0.0)); dist.all_reduce(mk)
    masks = mk.numpy().astype(np.uint64)
    cnt = np.array([bin(int(m)).count("1") for m in masks])
    owner = np.array([(int(m) & -int(m)).bit_length() - 1 for m in masks])
    assert cnt.min() >= 1 and not touched.all()                  # every keyframe is somebody's, nobody holds them all
    shared = np.nonzero(cnt >= 2)[0]                              # the same ordered list on every rank
    pos_of = -np.ones(N, int); pos_of[shared] = np.arange(len(shared))
    mine = np.nonzero(touched)[0]                                 # rank-local keyframes (ascending global order = local numbering)
    mine_shared = mine[cnt[mine] >= 2]
    own_w = (owner[mine] == rank).astype(np.float64)              # owner weights over the local keyframes
    rows = lambda v, k: v.reshape(N, k)                           # noqa: E731
    # ---- linearisation: cost (scalar sum), diagonal + gradient rows of the SHARED keyframes only
    tc = torch.tensor([cost], dtype=torch.float64); dist.all_reduce(tc)
    dg = np.concatenate([rows(np.diag(H)[:6 * N].copy(), 6), rows(grad[:6 * N].copy(), 6)], axis=1)    # [N][12]
    got, _ = _exchange_rows(dg[mine_shared], mine_shared, masks, rank, world)
    dg[mine_shared] = got
    diag_l, grad_l = dg[mine, :6], dg[mine, 6:]                   # complete on this rank's keyframes
    # ---- damped operator: the OWNER adds the damping once; shared rows of y summed, x.(A_r x) rides along
    radius = 1e4
    full_diag = np.zeros((N, 6)); full_diag[mine] = diag_l
    sc = 1.0 / (1.0 + np.sqrt(full_diag[mine]))
    lam = np.clip(sc ** 2 * full_diag[mine], 1e-6, 1e32) / (radius * sc ** 2)
    x = np.random.default_rng(0).normal(size=6 * N)
    x_l = np.zeros(6 * N); rows(x_l, 6)[mine] = rows(x, 6)[mine]  # a rank only ever holds x on its keyframes
    y = rows(_reduced_operator(H, N, owned_sw, radius, x_l), 6)
    y[mine] += own_w[:, None] * lam * rows(x, 6)[mine]
    assert np.abs(np.delete(y, mine, axis=0)).max(initial=0.0) == 0.0     # A_r only reaches the rank's own keyframes
    pAp_local = float((rows(x, 6)[mine] * y[mine]).sum())                       # rank-local partial: sum over ranks = x.Ax
    xx_local = float((own_w[:, None] * rows(x, 6)[mine] ** 2).sum())           # owner-weighted partial: every keyframe counted once
    got, extra = _exchange_rows(y[mine_shared], mine_shared, masks, rank, world, [pAp_local, xx_local])   # neighbour sends / receives + ONE 2-double all-reduce
    y[mine_shared] = got
    # ---- write-back: owners scatter their keyframes into a zeroed global array, one all-reduce replicates it
    wb = torch.zeros(N, 6, dtype=torch.float64)
    wb[mine[own_w > 0]] = torch.from_numpy(y[mine[own_w > 0]])
    dist.all_reduce(wb)
    np.savez(out % rank, cost=tc.numpy(), mine=mine, grad=grad_l, diag=diag_l, y=y[mine], pAp=extra[:1], xx=extra[1:], y_all=wb.numpy(), n_shared=len(shared))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,policy", [(2, "contiguous"), (3, "spatial"), (2, "chain")])
def test_rank_local_subgraphs_reproduce_the_full_system(tmp_path, world, policy):
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), policy, out), nprocs=world, join=True)
    g = util.small_graph(90, 18, f=2, seed=6)
    q, t, s = util.initial_state(g, True, perturb=0.02, seed=3)
    N, S = g.n_poses, g.n_loops
    O = util.oracle_problem(g, True)
    cost, _, grad = O.evaluate(q, t, s, want_residuals=False)
    H = O.dense_normal_matrix(q, t, s)
    radius = 1e4
    scale = 1.0 / (1.0 + np.sqrt(np.diag(H)))
    lam = np.clip(scale ** 2 * np.diag(H), 1e-6, 1e32) / (radius * scale ** 2)
    Hd = H + np.diag(lam)
    A = Hd[:6 * N, :6 * N] - Hd[:6 * N, 6 * N:] @ np.linalg.solve(Hd[6 * N:, 6 * N:], Hd[6 * N:, :6 * N])
    x = np.random.default_rng(0).normal(size=6 * N)
    Ax = (A @ x).reshape(N, 6)
    for rank in range(world):
        r = np.load(out % rank)
        mine = r["mine"]
        assert 0 < r["n_shared"] < N and len(mine) < N
        assert abs(r["cost"][0] - cost) <= 1e-12 * cost
        assert np.abs(r["grad"] - grad[:6 * N].reshape(N, 6)[mine]).max() <= 1e-11 * np.abs(grad).max()
        assert np.abs(r["diag"] - np.diag(H)[:6 * N].reshape(N, 6)[mine]).max() <= 1e-11 * np.diag(H).max()
        assert np.abs(r["y"] - Ax[mine]).max() <= 1e-10 * np.abs(Ax).max()          # complete on the rank's keyframes after ONE exchange
        assert abs(r["pAp"][0] - x @ (A @ x)) <= 1e-10 * abs(x @ (A @ x))           # sum over ranks of the rank-local partials
        assert abs(r["xx"][0] - x @ x) <= 1e-12 * (x @ x)                             # every keyframe counted once
        assert np.abs(r["y_all"] - Ax).max() <= 1e-10 * np.abs(Ax).max()             # owner-wise write-back


def _halo_worker(rank, world, port, policy, out):
    import ctypes as C
    from solve_keyframe_pose_graph_amd import graphgen
    from tests import test_mg_distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libmg_host.so"))
    lib.mgh_build.restype = C.c_void_p
    g = graphgen.generate(4000, 2000, odom_f_max=2, seed=9)
    sel = sharding.partition(g, world, policy)[rank]
    touched = np.zeros(g.n_poses, bool)
    for a in (g.odom_c1[sel("odom", g.n_odom)], g.odom_c2[sel("odom", g.n_odom)], g.loop_c1[sel("loop", g.n_loops)], g.loop_c2[sel("loop", g.n_loops)], g.reg_node[sel("reg", len(g.reg_node))]):
        touched[a] = True
    mk = torch.from_numpy(np.where(touched, float(1 << rank), 0.0)); dist.all_reduce(mk)          # the graph build's one all-reduce
    masks = mk.numpy().astype(np.uint64)
    # every rank builds the SAME hierarchy from the global graph + the masks, and its own plans
    h = D.build_owned(lib, g, masks, world, dist_min_rows=64, smoothed=1, policy=policy)      # (the owners: libpgo's second all-reduce, computed here from the partition)
    L = D.levels_of(lib, h, world)
    P = D.plans_of(lib, h, masks, world, 64, rank, len(L))
    ok = []
    for l, A in enumerate(L):
        own = D.owner_of_rows(A, world)
        truth = np.stack([np.arange(A["n"]) * 1.5 + 7.0 + 100.0 * l + c for c in range(6)], axis=1)       # 6 doubles per node, like a level vector
        vec = np.where((own == rank)[:, None], truth, np.nan)
        X = P[l]
        reqs, rb = [], {}
        for q in range(world):
            if q == rank:
                continue
            s_idx = X["send_idx"][X["send_off"][q]:X["send_off"][q + 1]]
            n_r = int(X["recv_off"][q + 1] - X["recv_off"][q])
            if len(s_idx):
                reqs.append(dist.isend(torch.from_numpy(vec[s_idx].copy()), dst=q))
            if n_r:
                rb[q] = torch.zeros(n_r, 6, dtype=torch.float64)
                reqs.append(dist.irecv(rb[q], src=q))
        for r in reqs:
            r.wait()
        for q, b in rb.items():
            vec[X["recv_idx"][X["recv_off"][q]:X["recv_off"][q + 1]]] = b.numpy()
        rows = np.nonzero(own == rank)[0]
        if A["distributed"]:
            cols = np.unique(np.concatenate([A["col"][A["rowptr"][i]:A["rowptr"][i + 1]] for i in rows])) if len(rows) else np.zeros(0, int)
            ok.append(bool(np.array_equal(vec[cols], truth[cols])) and bool(np.isnan(vec).any()))        # the halo is there, and it is a halo, not a gather
        else:
            ok.append(bool(np.array_equal(vec, truth)))
    np.savez(out % rank, ok=np.array(ok), n_levels=len(L), distributed=np.array([A["distributed"] for A in L]))
    lib.mgh_free(h)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,policy", [(2, "spatial"), (3, "spatial")])
def test_level_halo_protocol_over_gloo(tmp_path, world, policy):
    from tests.test_mg_hierarchy import shim as _shim_fixture  # noqa: F401  (the native library is built by that module's fixture; build it here when missing)
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "native", "libmg_host.so"), os.path.join(here, "native", "mg_host.cpp")
    hdr = os.path.join(os.path.dirname(here), "solve_keyframe_pose_graph_amd", "csrc", "pgo_mg_host.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.dirname(hdr), "-o", so, src])
    out = str(tmp_path / "halo%d.npz")
    mp.spawn(_halo_worker, args=(world, _free_port(), policy, out), nprocs=world, join=True)
    for rank in range(world):
        r = np.load(out % rank)
        assert r["ok"].all(), (rank, r["ok"])
        assert r["distributed"][:-1].any() and not r["distributed"][-1]


def test_policies_deal_out_every_edge_exactly_once():
    g = util.small_graph(300, 40, f=2, seed=11)
    for world in (1, 2, 3, 8):
        for policy in ("contiguous", "chain", "spatial"):
            parts = sharding.partition(g, world, policy)
            for kind, n in (("odom", g.n_odom), ("loop", g.n_loops), ("reg", len(g.reg_node))):
                got = np.sort(np.concatenate([parts[r](kind, n) for r in range(world)]))
                assert np.array_equal(got, np.arange(n)), (world, policy, kind)
            st = sharding.partition_stats(g, parts)
            assert sum(st["edges_per_rank"]) == g.n_odom + g.n_loops
            if policy != "contiguous" and world > 1:
                assert max(st["edges_per_rank"]) <= 1.5 * (g.n_odom + g.n_loops) / world + 8      # balanced by edge load
    # locality on the benchmark's graph family (C3 structure: loop closures tie places that are close in space, far apart in time):
    # spatial cells share fewer keyframes than index ranges, which share fewer than dealing out edge-index ranges
    from solve_keyframe_pose_graph_amd import graphgen
    g = graphgen.generate(40000, 40000, odom_f_max=2, seed=3)
    sh = {pol: sharding.partition_stats(g, sharding.partition(g, 4, pol))["shared_keyframes"] for pol in ("contiguous", "chain", "spatial")}
    assert sh["spatial"] < 0.7 * sh["chain"] and sh["chain"] < sh["contiguous"], sh


def test_slices_partition_every_edge_class():
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 300000):
            got = np.concatenate([edge_slice(r, world)("odom", n) for r in range(world)])
            assert np.array_equal(got, np.arange(n))
        assert sum(len(edge_slice(r, world)("reg", 3)) for r in range(world)) == 3


def _setup_worker(rank, world, port, policy, out):
    """the distributed set-up's first exchange between independent processes: every rank forms ITS part of level 1's blocks (scalar blocks: its own edges, its owned keyframes), sends
    the parts of the blocks it shares by the BlockPlan it derived on its own, and sums what it gets in ascending rank order — no handshake, the same bits on every needer"""
    import ctypes as C
    from solve_keyframe_pose_graph_amd import graphgen
    from tests import test_mg_distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libmg_host.so"))
    lib.mgh_build.restype = C.c_void_p
    g = graphgen.generate(4000, 2000, odom_f_max=2, seed=9)
    masks, kf_owner = D.touch_masks(g, world, policy, with_owner=True)
    h = D.build_owned(lib, g, masks, world, dist_min_rows=64, smoothed=1, owner=kf_owner, policy=policy)
    L = D.levels_of(lib, h, world)
    SP, (rc1, rc2, ro, sc1, sc2, so) = D.setup_plans_of(lib, h, g, world, policy, rank, L)
    A0 = L[0]
    agg0 = np.zeros(g.n_poses, np.int32); mem0_ptr = np.zeros(A0["n"] + 1, np.int32); mem0 = np.zeros(int((masks != 0).sum()), np.int32)
    lib.mgh_level0(h, D.ptr(agg0, C.c_int), D.ptr(mem0_ptr, C.c_int), D.ptr(mem0, C.c_int))

    def slot(a, b):
        if a == b:
            return int(A0["rowptr"][a])
        lo, hi = int(A0["rowptr"][a]) + 1, int(A0["rowptr"][a + 1])
        return lo + int(np.searchsorted(A0["col"][lo:hi], b))
    rng = np.random.default_rng(5)      # (the same stream on every rank: the truth is computable everywhere)
    dk = 5.0 + rng.random(g.n_poses)
    he = [-0.3 * rng.random(len(rc1)), -0.3 * rng.random(len(sc1))]
    truth = np.zeros(len(A0["col"])); mine = np.full(len(A0["col"]), np.nan)

    def add(r, k, v):
        truth[k] += v
        if r == rank:
            mine[k] = v if np.isnan(mine[k]) else mine[k] + v
    for i in range(g.n_poses):
        if agg0[i] >= 0:
            add(int(kf_owner[i]), slot(agg0[i], agg0[i]), dk[i])
    for cls, (c1, c2, off) in enumerate(((rc1, rc2, ro), (sc1, sc2, so))):
        for r in range(world):
            for e in range(int(off[r]), int(off[r + 1])):
                a, b = agg0[c1[e]], agg0[c2[e]]
                if a >= 0 and b >= 0:
                    add(r, slot(a, b), he[cls][e]); add(r, slot(b, a), he[cls][e])
    X = SP["val"][0]
    reqs, rb = [], {}
    recv = np.full(int(X["recv_off"][-1]), np.nan)
    for q in range(world):
        if q == rank:
            continue
        s_idx = X["send_idx"][X["send_off"][q]:X["send_off"][q + 1]]
        n_r = int(X["recv_off"][q + 1] - X["recv_off"][q])
        if len(s_idx):
            reqs.append(dist.isend(torch.from_numpy(mine[s_idx].copy()), dst=q))
        if n_r:
            rb[q] = torch.zeros(n_r, dtype=torch.float64)
            reqs.append(dist.irecv(rb[q], src=q))
    for r in reqs:
        r.wait()
    for q, b in rb.items():
        recv[X["recv_off"][q]:X["recv_off"][q + 1]] = b.numpy()
    for j, k in enumerate(X["dst"]):
        tot = 0.0
        for s_ in X["sum_src"][X["sum_ptr"][j]:X["sum_ptr"][j + 1]]:
            tot += mine[k] if s_ < 0 else recv[s_]
        mine[k] = tot
    own = D.owner_of_rows(A0, world)
    row_of = np.repeat(np.arange(A0["n"]), np.diff(A0["rowptr"]))
    need = np.nonzero((own[row_of] == rank) | ((A0["col"] > row_of) & (own[A0["col"]] == rank)))[0]
    # the blocks above the diagonal that two owners share: both needers must hold the SAME bits — gathered and compared on rank 0 by the test
    np.savez(out % rank, ok=bool(np.allclose(mine[need], truth[need], rtol=1e-13, atol=0)), first_whole=SP["first_whole"], sent=len(X["send_idx"]), blocks=len(A0["col"]),
             need=need, got=mine[need])
    lib.mgh_free(h)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,policy", [(2, "spatial"), (3, "spatial")])
def test_setup_block_sums_over_gloo(tmp_path, world, policy):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "native", "libmg_host.so"), os.path.join(here, "native", "mg_host.cpp")
    hdr = os.path.join(os.path.dirname(here), "solve_keyframe_pose_graph_amd", "csrc", "pgo_mg_host.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.dirname(hdr), "-o", so, src])
    out = str(tmp_path / "setup%d.npz")
    mp.spawn(_setup_worker, args=(world, _free_port(), policy, out), nprocs=world, join=True)
    R = [np.load(out % rank) for rank in range(world)]
    for rank, r in enumerate(R):
        assert bool(r["ok"]), rank
        assert int(r["first_whole"]) >= 1 and 0 < int(r["sent"]) < 0.2 * int(r["blocks"])      # only the blocks two ranks share travel
    for a in range(world):                                                                        # a block two ranks need holds the same bits on both
        for b in range(a + 1, world):
            common, ia, ib = np.intersect1d(R[a]["need"], R[b]["need"], return_indices=True)
            assert len(common) > 0 and np.array_equal(R[a]["got"][ia], R[b]["got"][ib])
