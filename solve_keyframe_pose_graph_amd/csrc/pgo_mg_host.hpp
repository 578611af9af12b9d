// pgo_mg_host.hpp — host side of the aggregation-multigrid preconditioner: builds, once per graph, the level hierarchy the device
// kernels of pgo_mg_kernels.hpp work on.  Included by pgo_solver.hip only.
//
// What it replaces in the reference: nothing one-to-one — Ceres factorises the normal equations exactly
// (SPARSE_NORMAL_CHOLESKY, reference src/PoseGraphSLAM.cpp:1270); here the PCG that stands in for that factorisation is
// preconditioned by  z = D^-1 r + P V(P^T r):  block-Jacobi on the keyframes plus one V(1,1) cycle over a hierarchy of ever coarser
// "keyframes", each the rigid-body motion of an aggregate of the level below (dtheta_i = dtheta_a, dt_i = dt_a - 2 [d_i]x dtheta_a,
// d_i = position_i - centroid_a), the coarsest level (<= dense_max nodes) solved densely.  Aggregates follow the GRAPH by heavy-edge pairwise
// matching (`passes` rounds per level -> up to 2^passes nodes each).  Level 1 matches keyframes along relative-pose (odometry) edges only:
// a switchable loop closure can be an outlier that the solver switches off a few LM steps later, and an aggregate held together by nothing
// else then stops being a rigid piece (measured on C3: 2-3x the PCG iterations from then on).  From level 1 up whole groups are matched
// along ALL their summed couplings — on revisited places loop closures tie groups as strongly as odometry does, and one dead edge among
// several no longer decides anything; chain-only aggregates on every level need ~2x the iterations (scripts/research/amg_probe.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <utility>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <atomic>
#include <new>
#include <vector>
#include <chrono>
#include <cstdio>
// phase times of build_hierarchy on stderr when pgo_mg::timing() is set (pgo_options.verbosity > 1)
#define PGO_MG_T0() auto t_mg_ = std::chrono::steady_clock::now()
#define PGO_MG_T(what) do { if (pgo_mg::timing()) { auto n_ = std::chrono::steady_clock::now(); std::fprintf(stderr, "[pgo] hierarchy (host): %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n_ - t_mg_).count()); t_mg_ = n_; } } while (0)

namespace pgo_mg { inline bool& timing() { static thread_local bool on = false; return on; } }      // per thread: every handle's hierarchy build (worker thread or caller) sets its own

namespace pgo_mg {

struct HostLevel {                       // level l >= 1
    int32_t n = 0;                       // nodes
    std::vector<int64_t> rowptr;         // [n+1] block rows, the diagonal block first
    std::vector<int32_t> col;            // [nnzb]
    std::vector<int64_t> g_ptr;          // [nnzb+1] contributions of the level below to each block
    std::vector<int64_t> g_ent;          // level 1: (index << 3) | kind — 0 keyframe diagonal block, 1/2 relative-pose edge forward/transposed, 3/4 switchable edge;
                                         // level >= 2: (row << 32) | block slot of the level below
    std::vector<int32_t> parent;         // [n] node of level l+1 (empty on the coarsest level); members of a parent are CONTIGUOUS
    std::vector<int32_t> agg_ptr;        // [n_next+1] first member of each parent
    std::vector<int32_t> tile_agg0;      // [tiles+1] workgroup tiles of whole aggregates, <= tile_rows / seg rows
    int seg = 1;                         // lanes-of-a-row groups that share one block row in the level kernels (1, 2, 4 or 8): long rows (smoothed Galerkin products) are split
    // SMOOTHED transition to the level above (smoothed aggregation): the prolongator is Ps = (I - w_p D^-1 A) P instead of the tentative P (rigid motion of the
    // parent), the level above is the Galerkin product Ps^T A Ps.  Only its STRUCTURE is fixed here; the numbers follow the LM system (pgo_mg_kernels.hpp):
    bool smoothed = false;
    std::vector<int32_t> ps_rowptr, ps_col;     // Ps: row i holds the parents of the columns of A's row i, ascending
    std::vector<int32_t> w_rowptr, w_col;       // W = A Ps: row i holds the union of the Ps rows of the columns of A's row i, ascending
    std::vector<int64_t> psT_ptr, psT_ent;      // Ps by coarse column: [n_next+1], entries (row << 32) | block of Ps
    // EXPLICIT transfer operator of the smoothed transition (round 5): inside the cycle the pre-smoothing step, the smoothed restriction, the smoothed prolongation and the
    // post-smoothing step of this level collapse into  v = x_pre + Dinv (r - A x_pre),  r_next = R r,  x = v + R^T x_next  with  R^T = Ps - Dinv W  on the pattern of W
    // (pattern(Ps) is a subset of it) — two row products per cycle on this level instead of four.  R^T lives on W's block-CSR (w_rowptr / w_col); R = its transpose by coarse row:
    std::vector<int32_t> ps_of_w;               // [n_w] block of Ps at the same (row, column) as block k of W, or -1
    std::vector<int32_t> rT_rowptr, rT_col;     // R by coarse row: [n_next+1]; fine rows ascending
    std::vector<int32_t> rT_of_w;               // [n_w] slot of block (column, row) in R's arrays for block k = (row, column) of W
    int rT_seg = 1;                             // lane groups sharing a row of R in the level kernel (tiles of tile_rows / rT_seg consecutive coarse rows)
    // SEVERAL RANKS, distributed cycle (round 6; build_hierarchy with an Owners argument): every aggregate of every level holds nodes of ONE rank, the levels are numbered
    // owner-major (rank 0's nodes first), tiles never mix ranks — so the rows a rank works on are one contiguous range of rows and of tiles on every level:
    std::vector<int32_t> own_ptr;               // [world+1] rows [own_ptr[r], own_ptr[r+1]) belong to rank r; empty on one GPU
    std::vector<int32_t> tile_ptr;              // [world+1] the rank's tiles (empty on the coarsest level)
    bool distributed = false;                   // the cycle's kernels of this level run on the owner's rows only (else: every rank runs all of it from gathered vectors)
};

struct Hierarchy {
    std::vector<HostLevel> L;            // L[0] = level 1
    // SMOOTHED transition keyframes -> level 1 (round 5, pgo_options::mg_smoothed_fine): F is the keyframe level as a block-CSR level of its own — rowptr / col = the solver's block pattern (diagonal block first,
    // then one block per incident edge; parallel edges repeat a column), parent = agg0 (-1: fixed keyframe, outside the system: its rows of Ps and W are empty) — with the
    // structures of Ps, W = A Ps and the explicit operator R^T; L[0]'s pattern is then that of Ps^T W and carries no contribution lists.
    bool fine_smoothed = false;
    HostLevel F;
    std::vector<int32_t> agg0;           // [N] level-1 node of each keyframe, -1 for keyframes outside the system (fixed)
    std::vector<int32_t> mem0_ptr, mem0; // level-1 node -> its keyframes
    int world = 1;                       // > 1: built with Owners (owner-pure aggregates, owner-major numbering, own_ptr / tile_ptr on every level)
};

// Several ranks (edge sharding), distributed cycle: which ranks hold a residual block on each keyframe (bit r of touch_mask[g]) and which of them OWNS it — the one holding most
// of its residual blocks (the lowest such rank): under a partition by place that is the rank of the keyframe's own cell, so that owners follow the partition and aggregates of
// one owner are whole pieces of trajectory (with "the lowest touching rank" every loop closure into a lower cell moved its far keyframe there: measured on config 5 with 8
// ranks, the owner-pure matching then stalled at 7 360 nodes where the single handle reaches 518).  Both are known on every rank after two all-reduces at graph build
// (the sum of 2^rank over the touching ranks is exact in a double up to 52 ranks; the maximum of (blocks + 1) * 64 + 63 - rank picks the owner).
struct Owners {
    const std::vector<uint64_t>* touch_mask = nullptr;   // [N] in the numbering the hierarchy is built in (the caller's global one)
    const std::vector<int32_t>* owner = nullptr;         // [N] owning rank, -1: no rank touches the keyframe
    int world = 1;
    int32_t dist_min_rows = 8192;                        // levels with fewer rows than this are not distributed: every rank runs all their rows from gathered vectors
};

struct WEdge { int32_t u, v; double w; };

// rows [0, n) in `parts` contiguous ranges, one thread each (the last one on the caller): fn(part, lo, hi).  The hierarchy build itself runs on a worker thread beside
// build_graph's other host work; inside it the independent pieces — level 1's block structure beside the matching of the levels above, the rows of the smoothed
// transition's sparsity patterns — are spread over a few more.  The results do not depend on the thread count: every part writes its own output, joined in order.
template <class Fn>
inline void parallel_ranges(int32_t n, int parts, Fn fn) {
    if (parts < 1) parts = 1;
    if (parts == 1 || n < 2048) { fn(0, 0, n); return; }
    // exception safety (the C-ABI above never throws; an allocation failure inside fn must not reach std::terminate): the helper threads are joined on EVERY exit of this
    // function — also while an exception thrown by fn on the calling thread unwinds — and an exception inside a helper thread is caught there and re-thrown here as bad_alloc
    std::atomic<bool> helper_failed{false};      // (declared BEFORE the joiner: destroyed after it, i.e. after the helper threads that may still store to it have been joined)
    struct Joiner { std::vector<std::thread> th; ~Joiner() { for (std::thread& t : th) if (t.joinable()) t.join(); } } j;
    const int32_t step = (n + parts - 1) / parts;
    for (int k = 0; k + 1 < parts; ++k) {
        const int32_t lo = std::min<int64_t>((int64_t)k * step, n), hi = std::min<int64_t>((int64_t)(k + 1) * step, n);
        bool started = false;
        try { j.th.emplace_back([=, &fn, &helper_failed]() { try { fn(k, lo, hi); } catch (...) { helper_failed.store(true); } }); started = true; } catch (...) {}
        if (!started) fn(k, lo, hi);      // no thread to be had: the same range on this one
    }
    fn(parts - 1, std::min<int64_t>((int64_t)(parts - 1) * step, n), n);
    for (std::thread& t : j.th) t.join();
    if (helper_failed.load()) throw std::bad_alloc();
}
inline int host_threads() { const unsigned hc = std::thread::hardware_concurrency(); return hc >= 32 ? 8 : hc >= 8 ? 4 : hc >= 4 ? 2 : 1; }      // (row ranges are joined in row order: the result does not depend on the count)

// Stable bucket sort: elements of `v` ordered by bucket(v[i]) in [0, nb), the order inside a bucket kept; `start` [nb+1] receives the bucket bounds.  The big sorts of
// the hierarchy build (700 000 block triples of C3's level 1, 200 000 couplings per matching pass) have keys "row, then column" with short rows: one linear pass by
// row and a small sort per row instead of a comparison sort of everything (level-1 block structure 31 -> 9 ms on C3).
template <class T, class BucketFn>
inline void bucket_sort(std::vector<T>& v, size_t nb, BucketFn bucket, std::vector<size_t>& start) {
    start.assign(nb + 1, 0);
    for (const T& x : v) start[(size_t)bucket(x) + 1]++;
    for (size_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    std::vector<T> out(v.size());
    std::vector<size_t> fill(start.begin(), start.end() - 1);
    for (const T& x : v) out[fill[(size_t)bucket(x)]++] = x;
    v.swap(out);
}
// indices 0 .. n-1 ordered by DESCENDING key (positive finite doubles), ties in index order: LSD radix sort on the bit patterns (monotone for positive doubles) —
// the same order as std::stable_sort with `key[a] > key[b]`
inline std::vector<uint32_t> order_descending(const std::vector<double>& key) {
    const size_t n = key.size();
    std::vector<uint64_t> k(n), k2(n);
    std::vector<uint32_t> idx(n), idx2(n);
    for (size_t i = 0; i < n; ++i) { uint64_t b; std::memcpy(&b, &key[i], 8); k[i] = ~b; idx[i] = (uint32_t)i; }
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 16 * pass;
        std::vector<uint32_t> cnt(65537, 0);
        for (size_t i = 0; i < n; ++i) cnt[((k[i] >> sh) & 0xffff) + 1]++;
        for (int b = 0; b < 65536; ++b) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; ++i) { const uint32_t o = cnt[(k[i] >> sh) & 0xffff]++; k2[o] = k[i]; idx2[o] = idx[i]; }
        k.swap(k2); idx.swap(idx2);
    }
    return idx;
}

// `passes` rounds of greedy heavy-edge matching.  edges: undirected, u != v, duplicates allowed (their weights add up).  Returns the
// aggregate of every node (ids in order of first appearance), n_agg through the reference.  `skip[i]` nodes get -1.
inline std::vector<int32_t> match_passes(int32_t n, std::vector<WEdge> edges, int passes, const std::vector<uint8_t>* skip, int32_t& n_agg) {
    std::vector<int32_t> agg(n);
    std::iota(agg.begin(), agg.end(), 0);
    int32_t cur_n = n;
    for (int p = 0; p < passes; ++p) {
        // merge parallel edges
        for (WEdge& e : edges) if (e.u > e.v) std::swap(e.u, e.v);
        {   // by (u, v): buckets by u, a small stable sort by v inside each
            std::vector<size_t> st;
            bucket_sort(edges, (size_t)cur_n, [](const WEdge& e) { return e.u; }, st);
            for (int32_t u = 0; u < cur_n; ++u) if (st[(size_t)u + 1] - st[u] > 1) std::stable_sort(edges.begin() + st[u], edges.begin() + st[(size_t)u + 1], [](const WEdge& a, const WEdge& b) { return a.v < b.v; });
        }
        size_t m = 0;
        for (size_t k = 0; k < edges.size(); ++k) {
            if (m > 0 && edges[m - 1].u == edges[k].u && edges[m - 1].v == edges[k].v) edges[m - 1].w += edges[k].w;
            else edges[m++] = edges[k];
        }
        edges.resize(m);
        // strength of a coupling relative to what else its endpoints are tied to: w_ij / sqrt(W_i W_j), W = weighted degree (the analogue of
        // |a_ij| / sqrt(a_ii a_jj)).  Raw summed weights would let the aggregates that have already merged the most keep pairing up with each
        // other while light nodes stay single for ever, and the coarse graphs degenerate into stars.
        std::vector<double> W(cur_n, 0.0);
        for (const WEdge& e : edges) { W[e.u] += e.w; W[e.v] += e.w; }
        std::vector<double> strength(edges.size());
        for (size_t k = 0; k < edges.size(); ++k) strength[k] = edges[k].w / std::sqrt(W[edges[k].u] * W[edges[k].v]);
        const std::vector<uint32_t> order = order_descending(strength);
        std::vector<int32_t> mate(cur_n, -1);
        for (uint32_t k : order) {
            const WEdge& e = edges[k];
            if (mate[e.u] >= 0 || mate[e.v] >= 0) continue;
            mate[e.u] = e.v; mate[e.v] = e.u;
        }
        std::vector<int32_t> a2(cur_n, -1);
        int32_t na = 0;
        for (int32_t i = 0; i < cur_n; ++i) {
            if (a2[i] >= 0) continue;
            a2[i] = na;
            if (mate[i] >= 0) a2[mate[i]] = na;
            ++na;
        }
        for (int32_t i = 0; i < n; ++i) agg[i] = a2[agg[i]];
        size_t m2 = 0;
        for (size_t k = 0; k < edges.size(); ++k) {
            const int32_t u = a2[edges[k].u], v = a2[edges[k].v];
            if (u != v) edges[m2++] = WEdge{u, v, edges[k].w};
        }
        edges.resize(m2);
        cur_n = na;
    }
    // skipped nodes are not part of any aggregate: renumber the rest in order of first appearance
    std::vector<int32_t> remap(cur_n, -1);
    int32_t na = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (skip && (*skip)[i]) { agg[i] = -1; continue; }
        if (remap[agg[i]] < 0) remap[agg[i]] = na++;
        agg[i] = remap[agg[i]];
    }
    n_agg = na;
    return agg;
}

// block structure of a level from (row node, column node, entry) triples: rows sorted, the diagonal block first in each row
inline void build_blocks(int32_t n, std::vector<std::pair<int64_t, int64_t>>& trip /* (row * n + col, entry) */, HostLevel& out) {
    // diagonal first: key' = row * (n+1) + (col == row ? 0 : col + 1)
    for (auto& t : trip) { const int64_t r = t.first / n, c = t.first % n; t.first = r * ((int64_t)n + 1) + (c == r ? 0 : c + 1); }
    {   // stable by key: buckets by row, a small stable sort inside each row
        std::vector<size_t> st;
        const int64_t n1 = (int64_t)n + 1;
        bucket_sort(trip, (size_t)n, [n1](const std::pair<int64_t, int64_t>& t) { return t.first / n1; }, st);
        for (int32_t r = 0; r < n; ++r) if (st[(size_t)r + 1] - st[r] > 1)
            std::stable_sort(trip.begin() + st[r], trip.begin() + st[(size_t)r + 1], [](const std::pair<int64_t, int64_t>& a, const std::pair<int64_t, int64_t>& b) { return a.first < b.first; });
    }
    out.rowptr.assign((size_t)n + 1, 0);
    out.col.clear(); out.g_ptr.clear(); out.g_ent.clear();
    int64_t prev = -1;
    for (const auto& t : trip) {
        if (t.first != prev) {
            const int64_t r = t.first / ((int64_t)n + 1), cc = t.first % ((int64_t)n + 1);
            out.col.push_back((int32_t)(cc == 0 ? r : cc - 1));
            out.g_ptr.push_back((int64_t)out.g_ent.size());
            out.rowptr[(size_t)r + 1]++;
            prev = t.first;
        }
        if (t.second >= 0) out.g_ent.push_back(t.second);
    }
    out.g_ptr.push_back((int64_t)out.g_ent.size());
    for (int32_t r = 0; r < n; ++r) out.rowptr[(size_t)r + 1] += out.rowptr[r];
}

// Several ranks (edge sharding): the hierarchy is built from the GLOBAL graph — every rank gathers the endpoints and weights of all edges and builds the
// same levels — while the contributions a rank adds to level 1's Galerkin product are its OWN: the diagonal blocks of the keyframes it owns and its own
// edges, both in the handle's rank-local numbering (the device kernel indexes the rank-local arrays).  The ranks' parts of a level-1 block are summed where the
// block is needed (build_setup_plans below: the distributed set-up; with pgo_options.mg_dist_setup = 0 by an all-reduce of all blocks, everything above level 1 then formed by every rank).
struct LocalContrib {
    const std::vector<int32_t>* l2g;          // local keyframe -> global keyframe
    const std::vector<double>* own;           // [N_local] 1.0 where this rank contributes the keyframe's (already summed) diagonal block
    const std::vector<int32_t>* rc1; const std::vector<int32_t>* rc2;     // the rank's relative-pose edges (GLOBAL endpoints), in its own edge order
    const std::vector<int32_t>* sc1; const std::vector<int32_t>* sc2;     // the rank's switchable edges
};

// What build_hierarchy keeps between two builds of the SAME graph (same keyframes, edges, fixed keyframes, relative-pose weights, options) when only the SWITCH values
// have changed — a regroup of the levels above level 1 inside a solve (pgo_solver.hip: regroup): the keyframes' level-1 aggregates (matched along relative-pose edges
// only: they do not depend on the switches), the structure of level 1 in their provisional numbering, and the level-1 couplings with the switchable part kept per pair.
// With a valid cache a rebuild skips the two expensive sorts (level-0 matching, level-1 block structure) and costs what the small upper levels cost.
struct BuildCache {
    bool valid = false;
    std::vector<int32_t> agg0_prov; int32_t n1 = 0;     // level-1 node of each keyframe, provisional numbering (after the run cuts); -1: outside the system
    HostLevel L1prov;                                     // rowptr / col / g_ptr / g_ent of level 1 in that numbering
    std::vector<WEdge> rel1;                              // level-1 couplings through relative-pose edges (collapsed; match_passes merges parallel ones)
    std::vector<int32_t> sw_u, sw_v;                      // level-1 pairs (u < v) coupled by switchable edges ...
    std::vector<int64_t> sw_ptr; std::vector<int32_t> sw_edge;   // ... and the switchable edges of each pair
    std::vector<int32_t> own1_prov;                       // several ranks: owner of each provisional level-1 node (its keyframes all have that owner)
};

// level 1 in its FINAL numbering from the cached provisional structure: rows permuted, columns relabelled, every row again "diagonal block first, then ascending column",
// every block with its contribution list as it was (the same lists, in the same order, as a direct build in the final numbering gives)
inline void permute_level1(const HostLevel& P, int32_t n, const std::vector<int32_t>& newid, HostLevel& out) {
    std::vector<int32_t> old_of((size_t)n);
    for (int32_t i = 0; i < n; ++i) old_of[newid[i]] = i;
    out.rowptr.assign((size_t)n + 1, 0); out.col.resize(P.col.size()); out.g_ptr.assign(P.col.size() + 1, 0); out.g_ent.resize(P.g_ent.size());
    int64_t kb = 0, ge = 0;
    std::vector<std::pair<int32_t, int64_t>> row;
    for (int32_t r = 0; r < n; ++r) {
        const int32_t o = old_of[r];
        row.clear();
        for (int64_t k = P.rowptr[o]; k < P.rowptr[(size_t)o + 1]; ++k) { const int32_t c = newid[P.col[k]]; row.push_back({c == r ? -1 : c, k}); }
        std::sort(row.begin(), row.end());
        for (const auto& ck : row) {
            out.col[(size_t)kb] = ck.first < 0 ? r : ck.first;
            out.g_ptr[(size_t)kb] = ge;
            for (int64_t g = P.g_ptr[ck.second]; g < P.g_ptr[(size_t)ck.second + 1]; ++g) out.g_ent[(size_t)ge++] = P.g_ent[g];
            ++kb;
        }
        out.rowptr[(size_t)r + 1] = kb;
    }
    out.g_ptr[(size_t)kb] = ge;
}

// Structure of a SMOOTHED transition from level A to the level above, B (n and the parents of A's nodes known): Ps, W = A Ps, Ps by coarse column, B = Ps^T W (diagonal block
// first), and the explicit operator's tables.  Nodes of A with parent -1 (fixed keyframes on the keyframe level) are outside the system: their rows of Ps and W are empty and
// they appear in no row as a column.
// ps_keep (keyframe level only, round 6): Ps is formed with a FILTERED matrix — blocks k of A with ps_keep[k] == 0 (switchable loop closures) do not enter (I - w D_f^-1 A_f) P, so a row
// of Ps holds the parents of the row's ODOMETRY neighbours only and level 1 does not take every loop closure of a neighbouring keyframe along; W = A Ps and the level above
// = Ps^T W are formed with the whole matrix (a Galerkin product of the true operator, whatever Ps is).
inline void smoothed_structure(HostLevel& A, HostLevel& B, const std::vector<uint8_t>* ps_keep = nullptr) {
    // structure of Ps, W = A Ps and B = Ps^T W (all by sorted unions; the numeric kernels search these short rows)
    A.smoothed = true;
    const int32_t n = A.n, nb = B.n;
    // rows are independent: contiguous row ranges on a few threads, each with its own output and marker array, joined in row order
    const int nth = host_threads();
    auto rows_in_parallel = [&](int32_t rows, std::vector<int32_t>& rowptr_out, std::vector<int32_t>& col_out, auto row_fn /* (row, tmp, stamp) -> fills tmp, sorted */) {
        std::vector<std::vector<int32_t>> part_cols((size_t)nth), part_len((size_t)nth);
        parallel_ranges(rows, nth, [&](int part, int32_t lo, int32_t hi) {
            std::vector<int32_t> tmp, stamp((size_t)nb, -1);
            std::vector<int32_t>& pc = part_cols[(size_t)part]; std::vector<int32_t>& pl = part_len[(size_t)part];
            pl.reserve((size_t)(hi - lo));
            for (int32_t i = lo; i < hi; ++i) { tmp.clear(); row_fn(i, tmp, stamp); pc.insert(pc.end(), tmp.begin(), tmp.end()); pl.push_back((int32_t)tmp.size()); }
        });
        rowptr_out.assign((size_t)rows + 1, 0); col_out.clear();
        int32_t r = 0;
        for (int k = 0; k < nth; ++k) {
            for (int32_t len : part_len[(size_t)k]) { rowptr_out[(size_t)r + 1] = rowptr_out[r] + len; ++r; }
            col_out.insert(col_out.end(), part_cols[(size_t)k].begin(), part_cols[(size_t)k].end());
        }
    };
    rows_in_parallel(n, A.ps_rowptr, A.ps_col, [&](int32_t i, std::vector<int32_t>& tmp, std::vector<int32_t>&) {
        if (A.parent[i] < 0) return;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[(size_t)i + 1]; ++k) { if (ps_keep && !(*ps_keep)[(size_t)k]) continue; const int32_t a = A.parent[A.col[k]]; if (a >= 0) tmp.push_back(a); }
        std::sort(tmp.begin(), tmp.end()); tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    });
    rows_in_parallel(n, A.w_rowptr, A.w_col, [&](int32_t i, std::vector<int32_t>& tmp, std::vector<int32_t>& stamp) {      // unions by marking: each coarse column enters a row's list once
        if (A.parent[i] < 0) return;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[(size_t)i + 1]; ++k) {
            const int32_t j = A.col[k];
            for (int32_t sl = A.ps_rowptr[j]; sl < A.ps_rowptr[(size_t)j + 1]; ++sl) { const int32_t c = A.ps_col[sl]; if (stamp[c] != i) { stamp[c] = i; tmp.push_back(c); } }
        }
        std::sort(tmp.begin(), tmp.end());
    });
    A.psT_ptr.assign((size_t)nb + 1, 0);
    for (int32_t c : A.ps_col) A.psT_ptr[(size_t)c + 1]++;
    for (int32_t a = 0; a < nb; ++a) A.psT_ptr[(size_t)a + 1] += A.psT_ptr[a];
    A.psT_ent.resize(A.ps_col.size());
    { std::vector<int64_t> fill(A.psT_ptr.begin(), A.psT_ptr.end() - 1);
      for (int32_t i = 0; i < n; ++i) for (int32_t sl = A.ps_rowptr[i]; sl < A.ps_rowptr[(size_t)i + 1]; ++sl) A.psT_ent[(size_t)fill[A.ps_col[sl]]++] = ((int64_t)i << 32) | (int64_t)sl; }
    {
        std::vector<int32_t> brow, bcol;
        rows_in_parallel(nb, brow, bcol, [&](int32_t a, std::vector<int32_t>& tmp, std::vector<int32_t>& stamp) {
            std::vector<int32_t> un;
            for (int64_t e = A.psT_ptr[a]; e < A.psT_ptr[(size_t)a + 1]; ++e) {
                const int32_t i = (int32_t)(A.psT_ent[e] >> 32);
                for (int32_t sl = A.w_rowptr[i]; sl < A.w_rowptr[(size_t)i + 1]; ++sl) { const int32_t c = A.w_col[sl]; if (stamp[c] != a) { stamp[c] = a; un.push_back(c); } }
            }
            std::sort(un.begin(), un.end());
            tmp.push_back(a);                                         // the diagonal block first, as everywhere
            for (int32_t c : un) if (c != a) tmp.push_back(c);
        });
        B.rowptr.assign(brow.begin(), brow.end()); B.col.swap(bcol);
    }
    B.g_ptr.assign(B.col.size() + 1, 0); B.g_ent.clear();         // (no contribution lists: the product is formed from Ps and W)
    // explicit transfer operator: Ps slot of every W block (both rows ascend by column: one merge per row), and W's pattern by coarse column
    A.ps_of_w.assign(A.w_col.size(), -1);
    for (int32_t i = 0; i < n; ++i) {
        int32_t ps = A.ps_rowptr[i]; const int32_t pe = A.ps_rowptr[(size_t)i + 1];
        for (int32_t k = A.w_rowptr[i]; k < A.w_rowptr[(size_t)i + 1] && ps < pe; ++k) if (A.w_col[k] == A.ps_col[ps]) A.ps_of_w[(size_t)k] = ps++;
    }
    A.rT_rowptr.assign((size_t)nb + 1, 0);
    for (int32_t c : A.w_col) A.rT_rowptr[(size_t)c + 1]++;
    for (int32_t a = 0; a < nb; ++a) A.rT_rowptr[(size_t)a + 1] += A.rT_rowptr[a];
    A.rT_col.resize(A.w_col.size()); A.rT_of_w.resize(A.w_col.size());
    { std::vector<int32_t> fill(A.rT_rowptr.begin(), A.rT_rowptr.end() - 1);
      for (int32_t i = 0; i < n; ++i) for (int32_t k = A.w_rowptr[i]; k < A.w_rowptr[(size_t)i + 1]; ++k) { const int32_t sl = fill[A.w_col[k]]++; A.rT_col[(size_t)sl] = i; A.rT_of_w[(size_t)k] = sl; } }
    {   // lane groups per coarse row, by the rule of the level kernels below (<= ~5 blocks per group, up to 8 groups)
        const double mean_row = (double)A.w_col.size() / (double)std::max(1, nb);
        static const double rt_blocks_per_group = []() {      // (debug override for scans, as below: PGO_ENABLE_DEBUG_HOOKS=1 and a value in [1, 64])
            const char* m = std::getenv("PGO_ENABLE_DEBUG_HOOKS"); const char* e = std::getenv("PGO_DEBUG_RT_SEG_BLOCKS");
            const double v = (m && m[0] == '1' && m[1] == 0 && e) ? std::atof(e) : 0.0; return v >= 1.0 && v <= 64.0 ? v : 5.0; }();
        A.rT_seg = 1;
        while (A.rT_seg < 8 && mean_row > rt_blocks_per_group * A.rT_seg) A.rT_seg *= 2;
    }
}

// N keyframes, node_free[N]; edge lists of both classes (endpoints in the handle's local numbering — the global one with several ranks —, weights of the
// relative-pose class at rel_w[rel_w_stride * e]).
// passes0: matching rounds keyframes -> level 1, passes: for the levels above.  Returns false when the graph does not coarsen down to
// dense_max nodes (e.g. mostly isolated keyframes): the caller then runs without the multigrid.
inline bool build_hierarchy(int64_t N, const std::vector<uint8_t>& node_free, const std::vector<int32_t>& rc1, const std::vector<int32_t>& rc2, const double* rel_w, int rel_w_stride,
                            const std::vector<int32_t>& sc1, const std::vector<int32_t>& sc2, const double* sw_weight /* per switchable edge: s^2 of its switch at graph build, or nullptr = 1 */,
                            int passes0, int passes, int dense_max, int tile_rows, int max_levels, Hierarchy& H, bool level0_follows_switchable = true, int level0_block = 0,
                            const LocalContrib* local = nullptr, int smoothed_levels = 0 /* transitions level l -> l+1, l = 1 .. smoothed_levels, use the smoothed prolongator */,
                            double loop_discount = 0.0 /* loop closures of a pair of level-1 nodes that do not count in the matching above level 1 */,
                            BuildCache* cache = nullptr /* kept by the caller across rebuilds of the same graph with other switch values (level0_follows_switchable must be false) */,
                            const std::vector<int64_t>* fine_rowptr = nullptr, const std::vector<int32_t>* fine_col = nullptr /* both given: the transition keyframes -> level 1 is SMOOTHED
                            too; the keyframe level's block pattern as the solver holds it (row i: block (i, i) first, then one block per incident edge) */,
                            const Owners* owners = nullptr /* several ranks, distributed cycle: aggregates never mix owners, levels numbered owner-major (HostLevel::own_ptr) */,
                            const std::vector<uint8_t>* fine_keep = nullptr /* with fine_rowptr / fine_col: blocks of the keyframe level that enter the smoothed prolongator (smoothed_structure) */) {
    H = Hierarchy{};
    const bool owned = owners && owners->touch_mask && owners->owner && owners->world > 1;
    const int world = owned ? owners->world : 1;
    H.world = world;
    auto kf_owner = [&](int64_t g) { return (*owners->owner)[(size_t)g]; };
    PGO_MG_T0();
    const int64_t Er = (int64_t)rc1.size(), Es = (int64_t)sc1.size();
    BuildCache own_cache;
    BuildCache& Cc = (cache && !level0_follows_switchable) ? *cache : own_cache;
    auto collapse = [](const std::vector<WEdge>& in, const std::vector<int32_t>& par) {
        std::vector<WEdge> out;
        out.reserve(in.size());
        for (const WEdge& e : in) { const int32_t u = par[e.u], v = par[e.v]; if (u >= 0 && v >= 0 && u != v) out.push_back({u, v, e.w}); }
        return out;
    };
    if (!Cc.valid) {
        std::vector<WEdge> rel_edges;
        rel_edges.reserve((size_t)Er);
        for (int64_t e = 0; e < Er; ++e) if (node_free[rc1[e]] && node_free[rc2[e]]) {
            const double w = rel_w[(size_t)rel_w_stride * e];
            if (owned && kf_owner(rc1[e]) != kf_owner(rc2[e])) continue;          // several ranks: an aggregate holds keyframes of ONE owner
            if (w * w > 1e-8) rel_edges.push_back({rc1[e], rc2[e], w * w});       // an odometry edge the yaw policy has (all but) switched off ties nothing together
        }
        std::vector<uint8_t> skip((size_t)N);
        for (int64_t i = 0; i < N; ++i) skip[i] = node_free[i] ? 0 : 1;
        int32_t n1 = 0;
        if (level0_follows_switchable) {
            // (research setting: keyframes matched across loop closures too — measured 2-3x the iterations once outliers are switched off; not cached)
            std::vector<WEdge> edges = rel_edges;
            for (int64_t e = 0; e < Es; ++e) if (node_free[sc1[e]] && node_free[sc2[e]] && !(owned && kf_owner(sc1[e]) != kf_owner(sc2[e]))) { const double w = sw_weight ? sw_weight[e] : 1.0; if (w > 1e-8) edges.push_back({sc1[e], sc2[e], w}); }
            Cc.agg0_prov = match_passes((int32_t)N, edges, passes0, &skip, n1);
        } else {
            // keyframes are grouped along relative-pose (odometry) edges only: a switchable loop closure may be an outlier the solver is about to
            // switch off, and an aggregate held together by nothing else would stop being a rigid piece; the levels above match along the summed
            // couplings of whole groups, where a single dead edge no longer decides anything
            Cc.agg0_prov = match_passes((int32_t)N, rel_edges, passes0, &skip, n1);
        }
        if (level0_block > 0 && n1 >= 1) {
            // The PCG's vector-update kernel works on consecutive runs of level0_block keyframes and restricts the new residual to the level-1 aggregates a run
            // holds COMPLETELY (no restriction kernel of its own: -5 us per iteration).  Regular odometry chains match into such aggregates by themselves; an
            // aggregate that does straddle a run boundary is cut there (one more, smaller aggregate per boundary at most).  Matching restricted to the runs from the
            // start was measured instead and dropped: on the 4-world benchmark graph it changed the whole greedy matching and cost 12 % more PCG iterations.
            std::vector<int64_t> first_run((size_t)n1, -1);
            const int64_t runs = (N + level0_block - 1) / level0_block;
            std::vector<std::pair<int64_t, int64_t>> parts;       // (aggregate * runs + run, keyframe) of keyframes outside their aggregate's first run
            for (int64_t i = 0; i < N; ++i) if (Cc.agg0_prov[i] >= 0) {
                const int64_t r = i / level0_block; const int32_t a = Cc.agg0_prov[i];
                if (first_run[a] < 0) first_run[a] = r;
                else if (first_run[a] != r) parts.push_back({(int64_t)a * runs + r, i});
            }
            std::sort(parts.begin(), parts.end());
            int64_t prev = -1;
            for (const auto& pr : parts) { if (pr.first != prev) { prev = pr.first; ++n1; } Cc.agg0_prov[pr.second] = n1 - 1; }
        }
        Cc.n1 = n1;
        if (n1 < 1) return false;
        Cc.own1_prov.clear();
        if (owned) { Cc.own1_prov.assign((size_t)n1, 0); for (int64_t i = 0; i < N; ++i) if (Cc.agg0_prov[i] >= 0) Cc.own1_prov[(size_t)Cc.agg0_prov[i]] = std::max(0, kf_owner(i)); }
        PGO_MG_T("level-0 matching + run cuts");
        // block structure and Galerkin contribution lists of level 1, in the provisional numbering: needs the keyframes' aggregates only, so it runs on a thread of its own
        // beside the level-1 couplings below (and, at a first build, its result is needed only after the levels above have been matched and numbered)
        auto level1_structure = [&]() {
            const std::vector<int32_t>& A0 = Cc.agg0_prov;
            std::vector<std::pair<int64_t, int64_t>> trip;
            trip.reserve((size_t)N + 2 * (size_t)(Er + Es));
            // (entry -1: the block exists — its structure is the global one on every rank — but the contribution is another rank's)
            for (int64_t i = 0; i < N; ++i) if (A0[i] >= 0) trip.push_back({(int64_t)A0[i] * n1 + A0[i], local ? (int64_t)-1 : ((i << 3) | 0)});
            auto edge = [&](int64_t e, int32_t c1, int32_t c2, int kind_fwd) {
                const int32_t a = A0[c1], b = A0[c2];
                if (a < 0 || b < 0) return;                     // rows and columns of fixed keyframes are not part of the system
                trip.push_back({(int64_t)a * n1 + b, e < 0 ? (int64_t)-1 : ((e << 3) | kind_fwd)});
                trip.push_back({(int64_t)b * n1 + a, e < 0 ? (int64_t)-1 : ((e << 3) | (kind_fwd + 1))});
            };
            for (int64_t e = 0; e < Er; ++e) edge(local ? -1 : e, rc1[e], rc2[e], 1);
            for (int64_t e = 0; e < Es; ++e) edge(local ? -1 : e, sc1[e], sc2[e], 3);
            if (local) {
                const int64_t Nl = (int64_t)local->l2g->size();
                for (int64_t l = 0; l < Nl; ++l) { const int32_t a = A0[(*local->l2g)[l]]; if (a >= 0 && (*local->own)[l] != 0.0) trip.push_back({(int64_t)a * n1 + a, (l << 3) | 0}); }
                for (int64_t e = 0; e < (int64_t)local->rc1->size(); ++e) edge(e, (*local->rc1)[e], (*local->rc2)[e], 1);
                for (int64_t e = 0; e < (int64_t)local->sc1->size(); ++e) edge(e, (*local->sc1)[e], (*local->sc2)[e], 3);
            }
            Cc.L1prov = HostLevel{};
            build_blocks(n1, trip, Cc.L1prov);
            Cc.L1prov.n = n1;
        };
        std::thread l1_thread;
        std::atomic<bool> l1_failed{false};
        if (host_threads() > 1) { try { l1_thread = std::thread([&]() { try { level1_structure(); } catch (...) { l1_failed.store(true); } }); } catch (...) {} }
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } l1_join{l1_thread};
        const bool l1_async = l1_thread.joinable();
        // level-1 couplings: the relative-pose part as collapsed edges, the switchable part per pair of level-1 nodes with the list of its edges (their weights change)
        Cc.rel1 = collapse(rel_edges, Cc.agg0_prov);
        {   // parallel couplings merged once (match_passes would sort all of them again at every rebuild)
            for (WEdge& e : Cc.rel1) if (e.u > e.v) std::swap(e.u, e.v);
            std::sort(Cc.rel1.begin(), Cc.rel1.end(), [](const WEdge& a, const WEdge& b) { return a.u != b.u ? a.u < b.u : a.v < b.v; });
            size_t m = 0;
            for (size_t k = 0; k < Cc.rel1.size(); ++k) { if (m > 0 && Cc.rel1[m - 1].u == Cc.rel1[k].u && Cc.rel1[m - 1].v == Cc.rel1[k].v) Cc.rel1[m - 1].w += Cc.rel1[k].w; else Cc.rel1[m++] = Cc.rel1[k]; }
            Cc.rel1.resize(m);
        }
        {
            std::vector<std::pair<int64_t, int32_t>> pe;      // (u * n1 + v, edge), u < v
            pe.reserve((size_t)Es);
            for (int64_t e = 0; e < Es; ++e) if (node_free[sc1[e]] && node_free[sc2[e]]) {
                int32_t u = Cc.agg0_prov[sc1[e]], v = Cc.agg0_prov[sc2[e]];
                if (u < 0 || v < 0 || u == v) continue;
                if (u > v) std::swap(u, v);
                pe.push_back({(int64_t)u * n1 + v, (int32_t)e});
            }
            std::sort(pe.begin(), pe.end());
            Cc.sw_u.clear(); Cc.sw_v.clear(); Cc.sw_ptr.clear(); Cc.sw_edge.clear();
            int64_t prev = -1;
            for (const auto& x : pe) {
                if (x.first != prev) { Cc.sw_u.push_back((int32_t)(x.first / n1)); Cc.sw_v.push_back((int32_t)(x.first % n1)); Cc.sw_ptr.push_back((int64_t)Cc.sw_edge.size()); prev = x.first; }
                Cc.sw_edge.push_back(x.second);
            }
            Cc.sw_ptr.push_back((int64_t)Cc.sw_edge.size());
        }
        PGO_MG_T("level-1 couplings");
        if (!l1_async) level1_structure();
        else { l1_thread.join(); if (l1_failed.load()) throw std::bad_alloc(); }      // (simple: joined here; the matching of the levels above is short next to it)
        Cc.valid = true;
        PGO_MG_T("level-1 block structure");
    }
    const int32_t n1 = Cc.n1;
    if (n1 < 1) return false;
    // level-1 edge list with the CURRENT switch weights.  The switchable functor ignores its edge weight (CeresResidues.h:198) and scales the whole block by its switch:
    // a loop closure the solver has switched off ties nothing together any more.
    // A single loop closure between two level-1 nodes may be an outlier that the solver switches off a few LM steps later; an aggregate of the levels above held
    // together by nothing else then stops being a rigid piece (the hierarchy is built before the switches are known, and regrouped at most a few times per solve).  Two or
    // more loop closures between the same two nodes — revisited places: parallel passes — are not all outliers: the matching above level 1 counts a pair's loop closures
    // minus `loop_discount`.
    std::vector<WEdge> cur = Cc.rel1;
    for (size_t pi = 0; pi < Cc.sw_u.size(); ++pi) {
        double w = 0.0;
        for (int64_t k = Cc.sw_ptr[pi]; k < Cc.sw_ptr[pi + 1]; ++k) { const double we = sw_weight ? sw_weight[Cc.sw_edge[(size_t)k]] : 1.0; if (we > 1e-8) w += we; }
        w -= loop_discount;
        if (w > 1e-8) cur.push_back({Cc.sw_u[pi], Cc.sw_v[pi], w});
    }
    // several ranks: couplings between nodes of different owners never match (every aggregate of every level stays with one rank); parents inherit the owner, so the
    // filter is needed once
    std::vector<int32_t> own_cur;                      // owner of each node of the level being matched (provisional numbering)
    if (owned) {
        own_cur = Cc.own1_prov;
        size_t m = 0;
        for (size_t k = 0; k < cur.size(); ++k) if (own_cur[(size_t)cur[k].u] == own_cur[(size_t)cur[k].v]) cur[m++] = cur[k];
        cur.resize(m);
    }
    // pass 1: aggregate level by level in provisional numbering; par[l] maps level l+1 (index l) to the level above
    std::vector<std::vector<int32_t>> par;
    std::vector<int32_t> n_of{n1};
    std::vector<std::vector<int32_t>> own_of;          // owned: own_of[l][i] = owner of provisional node i of level l+1
    if (owned) own_of.push_back(own_cur);
    for (int lvl = 1;; ++lvl) {
        const int32_t n = n_of.back();
        if (n <= std::max(dense_max, owned ? world : 0)) break;       // coarsest level: solved densely
        if (lvl >= max_levels) { if (timing()) std::fprintf(stderr, "[pgo] hierarchy (host): more than %d levels\n", max_levels); return false; }
        int32_t n_next = 0;
        par.push_back(match_passes(n, cur, passes, nullptr, n_next));      // (8-node aggregates from level 2 up, one level less: measured slower on C3 and C4 with the smoothed transition, 0.424 vs 0.411 s / 1.60 vs 1.47 s)
        if (owned && (double)n_next > 0.85 * (double)n) {
            // Several ranks: aggregates stay inside one owner, and a rank's part of the graph falls into pieces that nothing of the SAME rank ties together (a trajectory
            // crosses a cell of the partition many times) — once every piece is one node the matching has no partners left although thousands of nodes remain (config 5 on 8
            // ranks: 1 364 nodes at level 4, the single handle is at 518 there).  The nodes the matching left single are then grouped by owner in node order, up to
            // 2^passes per aggregate: a rigid aggregate of pieces that do not touch is a weaker coarse mode than a matched one, never a wrong one (the Galerkin product of any
            // aggregation is symmetric positive definite), and it only happens on the small top levels.
            std::vector<int32_t>& pr = par.back();
            std::vector<int32_t> cnt((size_t)n_next, 0);
            for (int32_t i = 0; i < n; ++i) cnt[(size_t)pr[(size_t)i]]++;
            const int group = 1 << std::max(1, passes);
            std::vector<int32_t> open_agg((size_t)world, -1), open_cnt((size_t)world, 0), target((size_t)n_next, -1);
            for (int32_t i = 0; i < n; ++i) {
                const int32_t a = pr[(size_t)i];
                if (cnt[(size_t)a] != 1) continue;
                const int o = own_of.back()[(size_t)i];
                if (open_agg[(size_t)o] < 0 || open_cnt[(size_t)o] >= group) { open_agg[(size_t)o] = a; open_cnt[(size_t)o] = 1; }
                else { target[(size_t)a] = open_agg[(size_t)o]; ++open_cnt[(size_t)o]; }
            }
            std::vector<int32_t> remap((size_t)n_next, -1);
            int32_t na = 0;
            for (int32_t i = 0; i < n; ++i) {
                int32_t a = pr[(size_t)i];
                if (target[(size_t)a] >= 0) a = target[(size_t)a];
                if (remap[(size_t)a] < 0) remap[(size_t)a] = na++;
                pr[(size_t)i] = remap[(size_t)a];
            }
            n_next = na;
        }
        if ((double)n_next > 0.85 * (double)n) {          // coarsening stalls
            if (owned && n <= 512) { par.pop_back(); break; }      // (several ranks: a level the dense solver can take then simply becomes the coarsest one)
            if (timing()) std::fprintf(stderr, "[pgo] hierarchy (host): coarsening stalls at level %d: %d -> %d nodes\n", lvl, n, n_next);
            return false;
        }
        cur = collapse(cur, par.back());
        n_of.push_back(n_next);
        if (owned) {
            std::vector<int32_t> on((size_t)n_next, 0);
            for (int32_t i = 0; i < n; ++i) on[(size_t)par.back()[(size_t)i]] = own_of.back()[(size_t)i];
            own_of.push_back(std::move(on));
        }
    }
    PGO_MG_T("upper-level matching");
    // pass 2, top down: number every level so that the members of a parent are contiguous and parents ascend
    const int nl = (int)n_of.size();
    H.L.resize((size_t)nl);
    std::vector<int32_t> newid_above;                  // final id of each provisional node of the level above (identity on the coarsest level)
    for (int l = nl - 1; l >= 0; --l) {
        HostLevel& Lv = H.L[l];
        const int32_t n = n_of[l];
        Lv.n = n;
        std::vector<int32_t> newid(n);
        if (l == nl - 1 && owned) {      // owner-major on the coarsest level; "children by parent" then makes every level below owner-major as well
            std::vector<int32_t> order(n);
            std::iota(order.begin(), order.end(), 0);
            const std::vector<int32_t>& ow = own_of[(size_t)l];
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return ow[(size_t)a] < ow[(size_t)b]; });
            for (int32_t k = 0; k < n; ++k) newid[order[(size_t)k]] = k;
        }
        else if (l == nl - 1) std::iota(newid.begin(), newid.end(), 0);
        else {
            const std::vector<int32_t>& pr = par[l];
            std::vector<int32_t> fpar(n), order(n);
            for (int32_t i = 0; i < n; ++i) fpar[i] = newid_above[pr[i]];
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return fpar[a] < fpar[b]; });
            Lv.parent.resize(n);
            for (int32_t k = 0; k < n; ++k) { newid[order[k]] = k; Lv.parent[k] = fpar[order[k]]; }
            const int32_t n_next = n_of[l + 1];
            Lv.agg_ptr.assign((size_t)n_next + 1, 0);
            for (int32_t k = 0; k < n; ++k) Lv.agg_ptr[(size_t)Lv.parent[k] + 1]++;
            for (int32_t a = 0; a < n_next; ++a) Lv.agg_ptr[(size_t)a + 1] += Lv.agg_ptr[a];
        }
        if (owned) {      // rows of each rank: the numbering is owner-major by construction
            Lv.own_ptr.assign((size_t)world + 1, 0);
            for (int32_t i = 0; i < n; ++i) Lv.own_ptr[(size_t)own_of[(size_t)l][(size_t)i] + 1]++;
            for (int r = 0; r < world; ++r) Lv.own_ptr[(size_t)r + 1] += Lv.own_ptr[(size_t)r];
            Lv.distributed = l + 1 < nl && n >= owners->dist_min_rows;
        }
        newid_above.swap(newid);
    }
    if (owned) for (int l = 0; l < nl; ++l) {      // (checked, not assumed: a rank's rows are one range)
        const HostLevel& Lv = H.L[(size_t)l];
        if (l + 1 < nl) for (int32_t k = 0; k < Lv.n; ++k) {
            const HostLevel& Up = H.L[(size_t)l + 1];
            const int32_t pa = Lv.parent[(size_t)k];
            const int ro = (int)(std::upper_bound(Lv.own_ptr.begin(), Lv.own_ptr.end(), k) - Lv.own_ptr.begin()) - 1;
            const int po = (int)(std::upper_bound(Up.own_ptr.begin(), Up.own_ptr.end(), pa) - Up.own_ptr.begin()) - 1;
            if (ro != po) { if (timing()) std::fprintf(stderr, "[pgo] hierarchy (host): level %d row %d: owner %d, its parent's %d\n", l + 1, k, ro, po); return false; }
        }
    }
    H.agg0 = Cc.agg0_prov;
    for (int32_t& a : H.agg0) if (a >= 0) a = newid_above[a];
    // level-1 membership lists (after the renumbering of level 1)
    H.mem0_ptr.assign((size_t)n1 + 1, 0);
    for (int64_t i = 0; i < N; ++i) if (H.agg0[i] >= 0) H.mem0_ptr[(size_t)H.agg0[i] + 1]++;
    for (int32_t a = 0; a < n1; ++a) H.mem0_ptr[(size_t)a + 1] += H.mem0_ptr[a];
    H.mem0.resize((size_t)H.mem0_ptr[n1]);
    { std::vector<int32_t> fill(H.mem0_ptr.begin(), H.mem0_ptr.end() - 1);
      for (int64_t i = 0; i < N; ++i) if (H.agg0[i] >= 0) H.mem0[(size_t)fill[H.agg0[i]]++] = (int32_t)i; }
    // block structure and Galerkin contribution lists of level 1: the cached provisional structure in the final numbering; the levels above bottom up
    PGO_MG_T("numbering + member lists");
    permute_level1(Cc.L1prov, n1, newid_above, H.L[0]);
    PGO_MG_T("permute level 1");
    if (fine_rowptr && fine_col && !local) {
        // smoothed transition keyframes -> level 1: level 1's pattern is that of Ps^T W (one hop wider than the aggregated edges), without contribution lists
        H.fine_smoothed = true;
        H.F = HostLevel{};
        H.F.n = (int32_t)N; H.F.rowptr = *fine_rowptr; H.F.col = *fine_col; H.F.parent = H.agg0;
        HostLevel B1; B1.n = n1;
        smoothed_structure(H.F, B1, fine_keep);
        H.L[0].rowptr.swap(B1.rowptr); H.L[0].col.swap(B1.col); H.L[0].g_ptr.swap(B1.g_ptr); H.L[0].g_ent.clear();
        PGO_MG_T("fine-level smoothed structure");
    }
    for (size_t l = 0; l + 1 < H.L.size(); ++l) {
        HostLevel& A = H.L[l];
        HostLevel& B = H.L[l + 1];
        if ((int)l < smoothed_levels) { smoothed_structure(A, B); continue; }
        std::vector<std::pair<int64_t, int64_t>> trip;
        trip.reserve(A.col.size());
        for (int32_t r = 0; r < A.n; ++r)
            for (int64_t k = A.rowptr[r]; k < A.rowptr[(size_t)r + 1]; ++k)
                trip.push_back({(int64_t)A.parent[r] * B.n + A.parent[A.col[k]], ((int64_t)r << 32) | k});
        build_blocks(B.n, trip, B);
    }
    PGO_MG_T("upper-level structures");
    // workgroup tiles of the level kernels: whole aggregates, <= tile_rows / seg rows; a level whose rows are long (the Galerkin product of a smoothed transition
    // has ~40 blocks per row instead of ~8) lets seg groups of lanes share each row, so that a row's blocks are streamed by seg x 6 lanes instead of 6
    for (size_t l = 0; l + 1 < H.L.size(); ++l) {
        HostLevel& Lv = H.L[l];
        const double mean_row = (double)Lv.col.size() / (double)std::max(1, Lv.n);
        int max_agg = 1;
        const int32_t n_next = H.L[l + 1].n;
        for (int32_t a = 0; a < n_next; ++a) max_agg = std::max(max_agg, Lv.agg_ptr[(size_t)a + 1] - Lv.agg_ptr[a]);
        // at most ~5 blocks per lane group, up to 8 groups per row (tiles of 4 rows): the level kernels are latency-bound, so what counts is how many of a row's loads are in
        // flight at once.  Measured, 20 LM steps, C3 / C4: <= 14 blocks per group and at most 4 groups (the first rule) 0.315 / 1.505 s; <= 10: 0.298 / 1.436; <= 7: 0.298 / 1.442;
        // <= 5: 0.291 / 1.417; <= 3.5: 0.290 / 1.471; <= 2.5: 0.299 / 1.468
        Lv.seg = 1;
        static const double blocks_per_group = []() {      // (debug override for scans: honoured only with PGO_ENABLE_DEBUG_HOOKS=1, values outside [1, 64] ignored)
            const char* m = std::getenv("PGO_ENABLE_DEBUG_HOOKS"); const char* e = std::getenv("PGO_DEBUG_SEG_BLOCKS");
            const double v = (m && m[0] == '1' && m[1] == 0 && e) ? std::atof(e) : 0.0; return v >= 1.0 && v <= 64.0 ? v : 5.0; }();
        while (Lv.seg < 8 && mean_row > blocks_per_group * Lv.seg) Lv.seg *= 2;
        while (Lv.seg > 1 && tile_rows / Lv.seg < max_agg) Lv.seg /= 2;         // an aggregate never straddles tiles
        const int cap = tile_rows / Lv.seg;
        Lv.tile_agg0.clear();
        Lv.tile_agg0.push_back(0);
        int rows = 0;
        const std::vector<int32_t>& up_own = H.L[l + 1].own_ptr;      // several ranks: a tile never mixes owners (aggregate a belongs to the rank whose range of the level above holds it)
        int r_cur = 0;                      // rank whose range of the level above holds the aggregates being packed
        for (int32_t a = 0; a < n_next; ++a) {
            const int sz = Lv.agg_ptr[(size_t)a + 1] - Lv.agg_ptr[a];
            bool owner_change = false;
            if (owned) while (r_cur + 1 < world && a >= up_own[(size_t)r_cur + 1]) { ++r_cur; owner_change = true; }
            if (rows > 0 && (rows + sz > cap || owner_change)) { Lv.tile_agg0.push_back(a); rows = 0; }
            rows += sz;
        }
        Lv.tile_agg0.push_back(n_next);
        if (owned) {      // tile ranges from the aggregates' owners (simple and checked: recomputed from the finished tile list)
            const int32_t nt = (int32_t)Lv.tile_agg0.size() - 1;
            Lv.tile_ptr.assign((size_t)world + 1, nt);
            Lv.tile_ptr[0] = 0;
            int32_t t = 0;
            for (int r = 0; r < world; ++r) {
                while (t < nt && Lv.tile_agg0[(size_t)t] < up_own[(size_t)r]) ++t;
                Lv.tile_ptr[(size_t)r] = t;
            }
            Lv.tile_ptr[(size_t)world] = nt;
            for (int r = 0; r < world; ++r) for (int32_t tt = Lv.tile_ptr[(size_t)r]; tt < Lv.tile_ptr[(size_t)r + 1]; ++tt)
                if (Lv.tile_agg0[(size_t)tt] < up_own[(size_t)r] || Lv.tile_agg0[(size_t)tt + 1] > up_own[(size_t)r + 1]) return false;      // a tile that straddles two ranks: never (guard)
        }
    }
    return true;
}

// ---- several ranks: who sends which rows to whom (round 6) ----
// Every exchange of the distributed solver is a NEIGHBOUR exchange: a rank sends the rows another rank reads and does not own, nothing else travels.  The plans are built on
// every rank from data all ranks hold (the keyframes' touch masks, the hierarchy built from the gathered graph, its ownership ranges), so sender and receiver agree on the
// contents and order of every segment without a handshake: segment (src -> dst) lists node ids in ascending order.
struct ExchangePlan {
    std::vector<int32_t> send_idx, recv_idx;       // concatenated per peer (peer 0 first), ascending inside a segment
    std::vector<int64_t> send_off, recv_off;       // [world+1] segment bounds, in rows
    std::vector<int64_t> pair_cnt;                 // [world*world] rows rank src sends to rank dst at [src * world + dst] — the layout of the all-reduce fallback and the byte counters
    int64_t n_send() const { return send_off.empty() ? 0 : send_off.back(); }
    int64_t n_recv() const { return recv_off.empty() ? 0 : recv_off.back(); }
};
inline void plan_from_keys(std::vector<uint64_t>& keys /* ((needer * world + provider) << 32) | node */, int world, int rank, ExchangePlan& P) {
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    P = ExchangePlan{};
    P.pair_cnt.assign((size_t)world * world, 0);
    P.send_off.assign((size_t)world + 1, 0); P.recv_off.assign((size_t)world + 1, 0);
    for (uint64_t k : keys) { const int np = (int)(k >> 32), needer = np / world, provider = np % world; P.pair_cnt[(size_t)provider * world + needer]++; }
    for (int q = 0; q < world; ++q) { P.send_off[(size_t)q + 1] = P.send_off[(size_t)q] + P.pair_cnt[(size_t)rank * world + q]; P.recv_off[(size_t)q + 1] = P.recv_off[(size_t)q] + P.pair_cnt[(size_t)q * world + rank]; }
    P.send_idx.resize((size_t)P.send_off[(size_t)world]); P.recv_idx.resize((size_t)P.recv_off[(size_t)world]);
    std::vector<int64_t> fs(P.send_off.begin(), P.send_off.end() - 1), fr(P.recv_off.begin(), P.recv_off.end() - 1);
    for (uint64_t k : keys) {      // keys ascend by (needer, provider, node): every segment comes out ascending
        const int np = (int)(k >> 32), needer = np / world, provider = np % world; const int32_t node = (int32_t)(k & 0xffffffffull);
        if (provider == rank) P.send_idx[(size_t)fs[(size_t)needer]++] = node;
        if (needer == rank) P.recv_idx[(size_t)fr[(size_t)provider]++] = node;
    }
}
// plans[l], l < n_levels = exchange of the vectors of level l+1 (the last one: the dense level's residual); plans[n_levels] = the prolongation to the keyframes (x of level 1 at
// the aggregates of every keyframe the rank touches — a subset of level 1's halo, sent on its own).  What a rank reads on a level it does not own there:
//   a distributed level:  the columns of its rows (row products of the down- and the up-sweep), with an explicit transfer operator above also the rows of R (restriction of its
//                         coarse rows) and, one level up, the columns of R^T (x_next in the up-sweep); on level 1 the aggregates of every keyframe the rank touches (prolongation)
//   any other level:      everything (its kernels run all rows on every rank from gathered vectors)
inline void build_level_plans(const Hierarchy& H, const Owners& O, int rank, std::vector<ExchangePlan>& plans) {
    const int world = O.world, nl = (int)H.L.size();
    plans.assign((size_t)nl + 1, ExchangePlan{});
    std::vector<std::vector<int32_t>> own((size_t)nl);
    for (int l = 0; l < nl; ++l) {
        const HostLevel& A = H.L[(size_t)l];
        own[(size_t)l].resize((size_t)A.n);
        for (int r = 0; r < world; ++r) for (int32_t i = A.own_ptr[(size_t)r]; i < A.own_ptr[(size_t)r + 1]; ++i) own[(size_t)l][(size_t)i] = r;
    }
    std::vector<std::vector<uint64_t>> keys((size_t)nl + 1);
    auto need_in = [&](int slot, int l, int needer, int32_t node) { const int prov = own[(size_t)l][(size_t)node]; if (prov != needer) keys[(size_t)slot].push_back(((uint64_t)(needer * world + prov) << 32) | (uint32_t)node); };
    auto need = [&](int l, int needer, int32_t node) { need_in(l, l, needer, node); };
    for (int l = 0; l < nl; ++l) {
        const HostLevel& A = H.L[(size_t)l];
        if (!A.distributed) { for (int32_t i = 0; i < A.n; ++i) for (int q = 0; q < world; ++q) need(l, q, i); continue; }
        for (int32_t i = 0; i < A.n; ++i) { const int o = own[(size_t)l][(size_t)i]; for (int64_t k = A.rowptr[(size_t)i]; k < A.rowptr[(size_t)i + 1]; ++k) need(l, o, A.col[(size_t)k]); }
        if (A.smoothed) {
            const int32_t nb = (int32_t)A.rT_rowptr.size() - 1;
            for (int32_t c = 0; c < nb; ++c) { const int oc = own[(size_t)l + 1][(size_t)c]; for (int32_t k = A.rT_rowptr[(size_t)c]; k < A.rT_rowptr[(size_t)c + 1]; ++k) need(l, oc, A.rT_col[(size_t)k]); }
            if (H.L[(size_t)l + 1].distributed)
                for (int32_t i = 0; i < A.n; ++i) { const int o = own[(size_t)l][(size_t)i]; for (int32_t k = A.w_rowptr[(size_t)i]; k < A.w_rowptr[(size_t)i + 1]; ++k) need(l + 1, o, A.w_col[(size_t)k]); }
        }
        if (l == 0) for (size_t g = 0; g < H.agg0.size(); ++g) if (H.agg0[g] >= 0) { uint64_t m = (*O.touch_mask)[g]; while (m) { const int q = __builtin_ctzll(m); m &= m - 1; need_in(nl, 0, q, H.agg0[g]); } }
    }
    for (int l = 0; l <= nl; ++l) plan_from_keys(keys[(size_t)l], world, rank, plans[(size_t)l]);
}
// The keyframes' own exchange (rows of the matvec output, of the diagonal blocks, of the gradient ...): a keyframe touched by several ranks holds a PARTIAL row on each of them; every
// touching rank sends its part to every other one and sums what it has and what it gets in ascending rank order — the same bits on every rank.  Local ids (l2g ascending).
struct FinePlan {
    ExchangePlan x;                                // send_idx: LOCAL keyframes whose rows go to each peer (recv_idx unused: the segment from peer q lists the same keyframes in the same order)
    std::vector<int32_t> sh_loc;                   // the rank's shared keyframes (local ids, ascending)
    std::vector<int32_t> sum_ptr, sum_src;         // per shared keyframe its parts in ascending rank order: -1 = the rank's own row, else the row of the receive buffer
};
inline void build_fine_plan(const std::vector<uint64_t>& touch_mask, const std::vector<int32_t>& l2g, int rank, int world, FinePlan& F) {
    F = FinePlan{};
    ExchangePlan& P = F.x;
    P.pair_cnt.assign((size_t)world * world, 0);
    for (uint64_t m0 : touch_mask) {
        if (!(m0 & (m0 - 1))) continue;            // one rank (or none): nothing travels
        for (uint64_t a = m0; a; a &= a - 1) for (uint64_t b = m0; b; b &= b - 1) { const int ra = __builtin_ctzll(a), rb = __builtin_ctzll(b); if (ra != rb) P.pair_cnt[(size_t)ra * world + rb]++; }
    }
    P.send_off.assign((size_t)world + 1, 0); P.recv_off.assign((size_t)world + 1, 0);
    for (int q = 0; q < world; ++q) { P.send_off[(size_t)q + 1] = P.send_off[(size_t)q] + P.pair_cnt[(size_t)rank * world + q]; P.recv_off[(size_t)q + 1] = P.recv_off[(size_t)q] + P.pair_cnt[(size_t)q * world + rank]; }
    P.send_idx.resize((size_t)P.send_off[(size_t)world]);
    std::vector<int64_t> fill(P.send_off.begin(), P.send_off.end() - 1);
    F.sum_ptr.push_back(0);
    for (size_t l = 0; l < l2g.size(); ++l) {
        const uint64_t m0 = touch_mask[(size_t)l2g[l]];
        if (!(m0 & (m0 - 1)) || !((m0 >> rank) & 1)) continue;
        F.sh_loc.push_back((int32_t)l);
        for (uint64_t a = m0; a; a &= a - 1) {
            const int q = __builtin_ctzll(a);
            if (q == rank) { F.sum_src.push_back(-1); continue; }
            F.sum_src.push_back((int32_t)(P.recv_off[(size_t)q] + (fill[(size_t)q] - P.send_off[(size_t)q])));      // (segment q -> rank lists the same keyframes in the same order as rank -> q)
            P.send_idx[(size_t)fill[(size_t)q]++] = (int32_t)l;
        }
        F.sum_ptr.push_back((int32_t)F.sum_src.size());
    }
}

// ---- several ranks: the multigrid's SET-UP distributed like its cycle (round 6) ----
// Every rank forms the numbers of its OWN rows of every distributed level (Galerkin products, block-Jacobi inverses, smoothed prolongators, transfer operators); what a kernel
// reads of another rank's rows arrives by a neighbour exchange of 6x6 BLOCKS, listed by slot.  Two kinds:
//   BlockPlan   — blocks with several CONTRIBUTORS (level 1: every rank holding an edge between two aggregates adds to their block; the product Ps^T W of a smoothed transition:
//                 every rank owning a row of Ps's column) go to the ranks that NEED them and are summed there in ascending rank order (the same bits on every needer).  A block is
//                 needed by the owner of its row and, above the diagonal, by the owner of its column too (the cycle streams an exactly symmetric fp32 copy: the block below the
//                 diagonal is the transpose of the rounded block above it); on the first level every rank runs completely, by everybody.
//   ExchangePlan over slots — blocks with ONE producer copied to their readers: the rows of Ps a rank's rows of W = A Ps multiply (the columns of its rows of A), the blocks of
//                 R = (Ps - Dinv W)^T whose coarse row belongs to another rank than their fine row.
// All of it derived on every rank from what all ranks hold (hierarchy, ownership ranges, the gathered edge list with the rank of every edge): no handshake.
struct BlockPlan {
    ExchangePlan x;                                // send_idx: slots whose blocks go to each peer (ascending inside a segment); recv_idx unused
    std::vector<int32_t> dst;                      // slots this rank needs and does not produce alone, ascending
    std::vector<int32_t> sum_ptr, sum_src;         // per dst its parts in ascending rank order: -1 = this rank's own values, else the row of the receive buffer
};
// entries (slot ascending, contributors, needers as rank bit masks)
inline void block_plan(const std::vector<int32_t>& slot, const std::vector<uint64_t>& contrib, const std::vector<uint64_t>& need, int rank, int world, BlockPlan& B) {
    B = BlockPlan{};
    ExchangePlan& P = B.x;
    P.pair_cnt.assign((size_t)world * world, 0);
    for (size_t k = 0; k < slot.size(); ++k)
        for (uint64_t c = contrib[k]; c; c &= c - 1) { const int rc = __builtin_ctzll(c); for (uint64_t n = need[k] & ~(1ull << rc); n; n &= n - 1) P.pair_cnt[(size_t)rc * world + (size_t)__builtin_ctzll(n)]++; }
    P.send_off.assign((size_t)world + 1, 0); P.recv_off.assign((size_t)world + 1, 0);
    for (int q = 0; q < world; ++q) { P.send_off[(size_t)q + 1] = P.send_off[(size_t)q] + P.pair_cnt[(size_t)rank * world + q]; P.recv_off[(size_t)q + 1] = P.recv_off[(size_t)q] + P.pair_cnt[(size_t)q * world + rank]; }
    P.send_idx.resize((size_t)P.send_off[(size_t)world]);
    std::vector<int64_t> fs(P.send_off.begin(), P.send_off.end() - 1), fr(P.recv_off.begin(), P.recv_off.end() - 1);
    const uint64_t me = 1ull << rank;
    B.sum_ptr.push_back(0);
    for (size_t k = 0; k < slot.size(); ++k) {
        if (contrib[k] & me) for (uint64_t n = need[k] & ~me; n; n &= n - 1) P.send_idx[(size_t)fs[(size_t)__builtin_ctzll(n)]++] = slot[k];
        if ((need[k] & me) && (contrib[k] & ~me)) {
            B.dst.push_back(slot[k]);
            for (uint64_t c = contrib[k]; c; c &= c - 1) { const int rc = __builtin_ctzll(c); B.sum_src.push_back(rc == rank ? -1 : (int32_t)fr[(size_t)rc]++); }
            B.sum_ptr.push_back((int32_t)B.sum_src.size());
        }
    }
}
struct SetupPlans {
    int first_whole = 0;                           // index of the first level every rank sets up (and runs) completely; levels below it are distributed.  0: nothing is (the set-up stays replicated)
    std::vector<BlockPlan> val;                    // [first_whole + 1] the level's own blocks
    std::vector<ExchangePlan> ps, rv;              // [first_whole] smoothed transition above level l: blocks of Ps (slots of ps_val), blocks of R (slots of r_valf)
    std::vector<std::vector<int32_t>> prod;        // [first_whole] smoothed transition above level l: the blocks of level l + 1 this rank's rows contribute to (ascending)
};
// rel / sw: the GATHERED edge lists (every rank's edges, rank by rank: edges [rel_off[r], rel_off[r+1]) are rank r's)
inline void build_setup_plans(const Hierarchy& H, int rank, int world, const std::vector<int32_t>& rc1, const std::vector<int32_t>& rc2, const std::vector<int64_t>& rel_off,
                              const std::vector<int32_t>& sc1, const std::vector<int32_t>& sc2, const std::vector<int64_t>& sw_off, SetupPlans& S) {
    S = SetupPlans{};
    const int nl = (int)H.L.size();
    if (world <= 1 || H.world != world || nl < 2 || !H.L[0].distributed) return;
    PGO_MG_T0();
    int fw = 0;
    while (fw < nl && H.L[(size_t)fw].distributed) ++fw;
    S.first_whole = fw;
    S.val.assign((size_t)fw + 1, BlockPlan{}); S.ps.assign((size_t)fw, ExchangePlan{}); S.rv.assign((size_t)fw, ExchangePlan{}); S.prod.assign((size_t)fw, std::vector<int32_t>{});
    std::vector<std::vector<int32_t>> own((size_t)fw + 1);
    for (int l = 0; l <= fw; ++l) {
        const HostLevel& A = H.L[(size_t)l];
        own[(size_t)l].resize((size_t)A.n);
        for (int r = 0; r < world; ++r) for (int32_t i = A.own_ptr[(size_t)r]; i < A.own_ptr[(size_t)r + 1]; ++i) own[(size_t)l][(size_t)i] = r;
    }
    const uint64_t all = world >= 64 ? ~0ull : ((1ull << world) - 1);
    auto find_block = [](const HostLevel& A, int32_t a, int32_t b) -> int64_t {      // rows hold the diagonal block first, the others by ascending column
        if (a == b) return A.rowptr[(size_t)a];
        const int32_t* lo = A.col.data() + A.rowptr[(size_t)a] + 1; const int32_t* hi = A.col.data() + A.rowptr[(size_t)a + 1];
        const int32_t* f = std::lower_bound(lo, hi, b);
        return (f != hi && *f == b) ? (int64_t)(f - A.col.data()) : -1;
    };
    std::vector<uint64_t> cm;                      // contributors of every block of the level in hand
    for (int l = 0; l <= fw; ++l) {
        const HostLevel& A = H.L[(size_t)l];
        const std::vector<int32_t>& ow = own[(size_t)l];
        cm.assign(A.col.size(), 0);
        if (l == 0) {
            for (int32_t a = 0; a < A.n; ++a) cm[(size_t)A.rowptr[(size_t)a]] |= 1ull << ow[(size_t)a];      // the keyframes' own (summed) diagonal blocks: their owner
            // (config 5: 3 M edges, two block look-ups each — spread over the host threads; the masks are OR-ed atomically, the result does not depend on the thread count)
            auto edges = [&](const std::vector<int32_t>& c1, const std::vector<int32_t>& c2, const std::vector<int64_t>& off) {
                const int64_t E = off[(size_t)world];
                if (E > 0x7fffffffll) return;
                parallel_ranges((int32_t)E, host_threads(), [&](int, int32_t lo, int32_t hi) {
                    int r = (int)(std::upper_bound(off.begin(), off.end(), (int64_t)lo) - off.begin()) - 1;
                    for (int64_t e = lo; e < hi; ++e) {
                        while (r + 1 < world && e >= off[(size_t)r + 1]) ++r;
                        const int32_t a = H.agg0[(size_t)c1[(size_t)e]], b = H.agg0[(size_t)c2[(size_t)e]];
                        if (a < 0 || b < 0) continue;
                        const int64_t k1 = find_block(A, a, b), k2 = find_block(A, b, a);
                        if (k1 >= 0) __atomic_fetch_or(&cm[(size_t)k1], 1ull << r, __ATOMIC_RELAXED);
                        if (k2 >= 0) __atomic_fetch_or(&cm[(size_t)k2], 1ull << r, __ATOMIC_RELAXED);
                    }
                });
            };
            edges(rc1, rc2, rel_off); edges(sc1, sc2, sw_off);
            PGO_MG_T("set-up plans: level-1 contributors");
        } else {
            const HostLevel& Lo = H.L[(size_t)l - 1];
            const std::vector<int32_t>& olo = own[(size_t)l - 1];
            if (!Lo.smoothed) { for (int32_t a = 0; a < A.n; ++a) for (int64_t k = A.rowptr[(size_t)a]; k < A.rowptr[(size_t)a + 1]; ++k) cm[(size_t)k] = 1ull << ow[(size_t)a]; }
            else {
                // the symbolic product Ps^T W row by row (C3 on 4 ranks: 1.3 M block look-ups in rows of ~50), spread over the host threads; atomic ORs: the same masks whatever the thread count
                std::vector<uint8_t> mine(A.col.size(), 0);
                parallel_ranges(Lo.n, host_threads(), [&](int, int32_t lo, int32_t hi) {
                    for (int32_t i = lo; i < hi; ++i) {
                        const uint64_t bit = 1ull << olo[(size_t)i];
                        const bool me = olo[(size_t)i] == rank;
                        for (int32_t pk = Lo.ps_rowptr[(size_t)i]; pk < Lo.ps_rowptr[(size_t)i + 1]; ++pk)
                            for (int32_t wk = Lo.w_rowptr[(size_t)i]; wk < Lo.w_rowptr[(size_t)i + 1]; ++wk) {
                                const int64_t k = find_block(A, Lo.ps_col[(size_t)pk], Lo.w_col[(size_t)wk]);
                                if (k < 0) continue;
                                if (!(__atomic_load_n(&cm[(size_t)k], __ATOMIC_RELAXED) & bit)) __atomic_fetch_or(&cm[(size_t)k], bit, __ATOMIC_RELAXED);
                                if (me) __atomic_store_n(&mine[(size_t)k], (uint8_t)1, __ATOMIC_RELAXED);
                            }
                    }
                });
                for (size_t k = 0; k < mine.size(); ++k) if (mine[k]) S.prod[(size_t)l - 1].push_back((int32_t)k);
            }
        }
        std::vector<int32_t> slot; std::vector<uint64_t> cb, nd;
        for (int32_t a = 0; a < A.n; ++a) for (int64_t k = A.rowptr[(size_t)a]; k < A.rowptr[(size_t)a + 1]; ++k) {
            const int32_t b = A.col[(size_t)k];
            uint64_t need = l == fw ? all : (1ull << ow[(size_t)a]) | (b > a ? 1ull << ow[(size_t)b] : 0ull);
            const uint64_t c = cm[(size_t)k];
            if (!c) continue;
            if (!(c & (c - 1)) && need == c) continue;          // produced and read by one and the same rank: nothing travels
            slot.push_back((int32_t)k); cb.push_back(c); nd.push_back(need);
        }
        block_plan(slot, cb, nd, rank, world, S.val[(size_t)l]);
        PGO_MG_T("set-up plans: a level's blocks");
        if (l == fw || !A.smoothed) continue;
        // smoothed transition above level l: rows of Ps the rank's rows of W read; blocks of R whose coarse row is another rank's
        std::vector<uint64_t> need_row((size_t)A.n, 0);
        for (int32_t r = 0; r < A.n; ++r) for (int64_t k = A.rowptr[(size_t)r]; k < A.rowptr[(size_t)r + 1]; ++k) need_row[(size_t)A.col[(size_t)k]] |= 1ull << ow[(size_t)r];
        std::vector<uint64_t> keys;
        for (int32_t j = 0; j < A.n; ++j) {
            const int prov = ow[(size_t)j];
            for (uint64_t n = need_row[(size_t)j] & ~(1ull << prov); n; n &= n - 1) { const int q = __builtin_ctzll(n); for (int32_t pk = A.ps_rowptr[(size_t)j]; pk < A.ps_rowptr[(size_t)j + 1]; ++pk) keys.push_back(((uint64_t)(q * world + prov) << 32) | (uint32_t)pk); }
        }
        plan_from_keys(keys, world, rank, S.ps[(size_t)l]);
        keys.clear();
        const std::vector<int32_t>& oup = own[(size_t)l + 1];
        for (int32_t i = 0; i < A.n; ++i) for (int32_t wk = A.w_rowptr[(size_t)i]; wk < A.w_rowptr[(size_t)i + 1]; ++wk) {
            const int prov = ow[(size_t)i], q = oup[(size_t)A.w_col[(size_t)wk]];
            if (prov != q) keys.push_back(((uint64_t)(q * world + prov) << 32) | (uint32_t)A.rT_of_w[(size_t)wk]);
        }
        plan_from_keys(keys, world, rank, S.rv[(size_t)l]);
        PGO_MG_T("set-up plans: Ps and R of a level");
    }
}

}  // namespace pgo_mg
