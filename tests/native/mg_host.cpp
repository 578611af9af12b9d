// Test shim: the host-side hierarchy builder of the aggregation multigrid (csrc/pgo_mg_host.hpp) behind a C interface, so that the CPU
// suite can check its invariants.  Host logic only; the cycle itself runs in HIP kernels.
#include <cstring>
#include "pgo_mg_host.hpp"

extern "C" {
void* mgh_build(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                int passes0, int passes, int dense_max, int tile_rows, int max_levels, int level0_follows_switchable, int smoothed_levels) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> meas((size_t)Er * 8, 0.0);
    for (long long e = 0; e < Er; ++e) meas[8 * e + 7] = rw[e];
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    if (!pgo_mg::build_hierarchy(N, nf, a, b, meas.data() + 7, 8, c, d, nullptr, passes0, passes, dense_max, tile_rows, max_levels, *H, level0_follows_switchable != 0, 0, nullptr, smoothed_levels)) { delete H; return nullptr; }
    return H;
}
// Several ranks: the structure from the global graph, level 1's contribution lists from the edges dealt to `rank` (rank_rel / rank_sw: the rank holding each edge)
// and the keyframes it owns (owner[]), in that rank's LOCAL numbering: keyframes touched by its edges in ascending global order, its edges in global order.
// l2g_out [N] receives the local -> global keyframe map (n_local returned), rel_l2g_out / sw_l2g_out the local -> global edge maps.
void* mgh_build_sharded(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                        int passes0, int passes, int dense_max, int tile_rows, int max_levels, const int* rank_rel, const int* rank_sw, const int* owner, int rank,
                        int* l2g_out, long long* n_local, int* rel_l2g_out, long long* n_rel_local, int* sw_l2g_out, long long* n_sw_local) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> w(rw, rw + Er);
    std::vector<uint8_t> touched((size_t)N, 0);
    std::vector<int32_t> la, lb, lc, ld;
    long long nr = 0, ns = 0;
    for (long long e = 0; e < Er; ++e) if (rank_rel[e] == rank) { la.push_back(a[e]); lb.push_back(b[e]); touched[a[e]] = touched[b[e]] = 1; rel_l2g_out[nr++] = (int)e; }
    for (long long e = 0; e < Es; ++e) if (rank_sw[e] == rank) { lc.push_back(c[e]); ld.push_back(d[e]); touched[c[e]] = touched[d[e]] = 1; sw_l2g_out[ns++] = (int)e; }
    std::vector<int32_t> l2g; std::vector<double> own;
    for (long long g = 0; g < N; ++g) if (touched[g]) { l2g_out[l2g.size()] = (int)g; l2g.push_back((int32_t)g); own.push_back(owner[g] == rank ? 1.0 : 0.0); }
    *n_local = (long long)l2g.size(); *n_rel_local = nr; *n_sw_local = ns;
    pgo_mg::LocalContrib L{&l2g, &own, &la, &lb, &lc, &ld};
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    if (!pgo_mg::build_hierarchy(N, nf, a, b, w.data(), 1, c, d, nullptr, passes0, passes, dense_max, tile_rows, max_levels, *H, false, 0, &L)) { delete H; return nullptr; }
    return H;
}
// Regroup path: the hierarchy for switch weights `sw_w`; use_cache != 0: a first build with `sw_w_first` fills a BuildCache and the returned hierarchy is the REBUILD
// with `sw_w` from that cache (what pgo_solver.hip's regroup does); use_cache == 0: a fresh build with `sw_w`.  The two must be identical.
void* mgh_build_regroup(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                        const double* sw_w_first, const double* sw_w, int use_cache, int passes0, int passes, int dense_max, int tile_rows, int max_levels, int smoothed_levels, double loop_discount,
                        int level0_block) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> w(rw, rw + Er);
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    pgo_mg::BuildCache cache;
    bool ok = true;
    if (use_cache) {
        pgo_mg::Hierarchy first;
        ok = pgo_mg::build_hierarchy(N, nf, a, b, w.data(), 1, c, d, sw_w_first, passes0, passes, dense_max, tile_rows, max_levels, first, false, level0_block, nullptr, smoothed_levels, loop_discount, &cache) && cache.valid;
    }
    ok = ok && pgo_mg::build_hierarchy(N, nf, a, b, w.data(), 1, c, d, sw_w, passes0, passes, dense_max, tile_rows, max_levels, *H, false, level0_block, nullptr, smoothed_levels, loop_discount, use_cache ? &cache : nullptr);
    if (!ok) { delete H; return nullptr; }
    return H;
}
// Smoothed transition keyframes -> level 1: the keyframe level's block pattern built as pgo_solver.hip's build_graph builds it (row i: block (i, i), then one block per incident
// edge — relative-pose edges first, in edge order —, parallel edges repeating a column), handed to build_hierarchy with level0_block as the solver does.
void* mgh_build_fine(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                     int passes0, int passes, int dense_max, int tile_rows, int max_levels, int smoothed_levels, int level0_block) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> w(rw, rw + Er);
    std::vector<int64_t> rowptr((size_t)N + 1, 0);
    for (long long e = 0; e < Er; ++e) { rowptr[a[e] + 1]++; rowptr[b[e] + 1]++; }
    for (long long e = 0; e < Es; ++e) { rowptr[c[e] + 1]++; rowptr[d[e] + 1]++; }
    for (long long n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n] + 1;
    std::vector<int32_t> col((size_t)rowptr[N]);
    std::vector<int64_t> fill((size_t)N);
    for (long long n = 0; n < N; ++n) { col[rowptr[n]] = (int32_t)n; fill[n] = rowptr[n] + 1; }
    for (long long e = 0; e < Er; ++e) { col[fill[a[e]]++] = b[e]; col[fill[b[e]]++] = a[e]; }
    for (long long e = 0; e < Es; ++e) { col[fill[c[e]]++] = d[e]; col[fill[d[e]]++] = c[e]; }
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    if (!pgo_mg::build_hierarchy(N, nf, a, b, w.data(), 1, c, d, nullptr, passes0, passes, dense_max, tile_rows, max_levels, *H, false, level0_block, nullptr, smoothed_levels, 0.0, nullptr, &rowptr, &col)) { delete H; return nullptr; }
    return H;
}
// (level -1 in the accessors below = the keyframe level F of a hierarchy built with the smoothed keyframe transition)
static const pgo_mg::HostLevel& level_of(void* h, int l) { pgo_mg::Hierarchy* H = (pgo_mg::Hierarchy*)h; return l < 0 ? H->F : H->L[l]; }
long long mgh_fine_nnzb(void* h) { return ((pgo_mg::Hierarchy*)h)->fine_smoothed ? (long long)((pgo_mg::Hierarchy*)h)->F.col.size() : -1; }
void mgh_fine_pattern(void* h, long long* rowptr, int* col) { const pgo_mg::HostLevel& F = ((pgo_mg::Hierarchy*)h)->F; std::memcpy(rowptr, F.rowptr.data(), F.rowptr.size() * 8); std::memcpy(col, F.col.data(), F.col.size() * 4); }
void mgh_set_timing(int on) { pgo_mg::timing() = on != 0; }      // phase times of build_hierarchy on stderr (this thread's builds)
void mgh_free(void* h) { delete (pgo_mg::Hierarchy*)h; }
int mgh_levels(void* h) { return (int)((pgo_mg::Hierarchy*)h)->L.size(); }
void mgh_sizes(void* h, int l, long long* out /* n, nnzb, n_ent, n_parent, n_agg_ptr, n_tiles_plus_1 */) {
    const pgo_mg::HostLevel& A = ((pgo_mg::Hierarchy*)h)->L[l];
    out[0] = A.n; out[1] = (long long)A.col.size(); out[2] = (long long)A.g_ent.size(); out[3] = (long long)A.parent.size(); out[4] = (long long)A.agg_ptr.size(); out[5] = (long long)A.tile_agg0.size();
}
void mgh_level(void* h, int l, long long* rowptr, int* col, long long* g_ptr, long long* g_ent, int* parent, int* agg_ptr, int* tile_agg0) {
    const pgo_mg::HostLevel& A = ((pgo_mg::Hierarchy*)h)->L[l];
    std::memcpy(rowptr, A.rowptr.data(), A.rowptr.size() * 8); std::memcpy(col, A.col.data(), A.col.size() * 4);
    std::memcpy(g_ptr, A.g_ptr.data(), A.g_ptr.size() * 8); std::memcpy(g_ent, A.g_ent.data(), A.g_ent.size() * 8);
    if (!A.parent.empty()) std::memcpy(parent, A.parent.data(), A.parent.size() * 4);
    if (!A.agg_ptr.empty()) std::memcpy(agg_ptr, A.agg_ptr.data(), A.agg_ptr.size() * 4);
    if (!A.tile_agg0.empty()) std::memcpy(tile_agg0, A.tile_agg0.data(), A.tile_agg0.size() * 4);
}
// structure of a smoothed transition: sizes {n_ps, n_w, n_psT_ent}, then the arrays
void mgh_smoothed_sizes(void* h, int l, long long* out) {
    const pgo_mg::HostLevel& A = level_of(h, l);
    out[0] = A.smoothed ? (long long)A.ps_col.size() : -1; out[1] = (long long)A.w_col.size(); out[2] = (long long)A.psT_ent.size();
}
void mgh_smoothed(void* h, int l, int* ps_rowptr, int* ps_col, int* w_rowptr, int* w_col, long long* psT_ptr, long long* psT_ent) {
    const pgo_mg::HostLevel& A = level_of(h, l);
    std::memcpy(ps_rowptr, A.ps_rowptr.data(), A.ps_rowptr.size() * 4); std::memcpy(ps_col, A.ps_col.data(), A.ps_col.size() * 4);
    std::memcpy(w_rowptr, A.w_rowptr.data(), A.w_rowptr.size() * 4); std::memcpy(w_col, A.w_col.data(), A.w_col.size() * 4);
    std::memcpy(psT_ptr, A.psT_ptr.data(), A.psT_ptr.size() * 8); std::memcpy(psT_ent, A.psT_ent.data(), A.psT_ent.size() * 8);
}
// explicit transfer operator of a smoothed transition (R^T on W's pattern): Ps slot of every W block, W's pattern by coarse row, slot of every W block there, lane groups
void mgh_explicit(void* h, int l, int* ps_of_w, int* rT_rowptr, int* rT_col, int* rT_of_w, int* rT_seg) {
    const pgo_mg::HostLevel& A = level_of(h, l);
    std::memcpy(ps_of_w, A.ps_of_w.data(), A.ps_of_w.size() * 4); std::memcpy(rT_rowptr, A.rT_rowptr.data(), A.rT_rowptr.size() * 4);
    std::memcpy(rT_col, A.rT_col.data(), A.rT_col.size() * 4); std::memcpy(rT_of_w, A.rT_of_w.data(), A.rT_of_w.size() * 4);
    *rT_seg = A.rT_seg;
}
void mgh_level0(void* h, int* agg0, int* mem0_ptr, int* mem0) {
    const pgo_mg::Hierarchy& H = *(pgo_mg::Hierarchy*)h;
    std::memcpy(agg0, H.agg0.data(), H.agg0.size() * 4); std::memcpy(mem0_ptr, H.mem0_ptr.data(), H.mem0_ptr.size() * 4); std::memcpy(mem0, H.mem0.data(), H.mem0.size() * 4);
}

// ---- several ranks, distributed cycle (round 6): owner-pure aggregates, owner-major numbering, exchange plans ----
// masks [N]: bit r = rank r holds a residual block on the keyframe (what libpgo's build_graph all-reduces).  Built from the GLOBAL graph, as every rank does.
void* mgh_build_owned(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                      int passes0, int passes, int dense_max, int tile_rows, int max_levels, int smoothed_levels, const unsigned long long* masks, const int* owner, int world, int dist_min_rows) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> w(rw, rw + Er);
    std::vector<uint64_t> m(masks, masks + N);
    std::vector<int32_t> ow(owner, owner + N);
    pgo_mg::Owners O; O.touch_mask = &m; O.owner = &ow; O.world = world; O.dist_min_rows = dist_min_rows;
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    if (!pgo_mg::build_hierarchy(N, nf, a, b, w.data(), 1, c, d, nullptr, passes0, passes, dense_max, tile_rows, max_levels, *H, false, 0, nullptr, smoothed_levels, 0.0, nullptr, nullptr, nullptr, &O)) { delete H; return nullptr; }
    return H;
}
int mgh_world(void* h) { return ((pgo_mg::Hierarchy*)h)->world; }
void mgh_ownership(void* h, int l, int* own_ptr /* world+1 */, int* tile_ptr /* world+1; zeros on the coarsest level */, int* distributed) {
    const pgo_mg::Hierarchy& H = *(pgo_mg::Hierarchy*)h; const pgo_mg::HostLevel& A = H.L[l];
    for (int r = 0; r <= H.world; ++r) { own_ptr[r] = A.own_ptr.empty() ? 0 : A.own_ptr[r]; tile_ptr[r] = A.tile_ptr.empty() ? 0 : A.tile_ptr[r]; }
    *distributed = A.distributed ? 1 : 0;
}
struct PlanSet { std::vector<pgo_mg::ExchangePlan> plans; pgo_mg::FinePlan fine; };
void* mgh_plans(void* h, const unsigned long long* masks, long long N, int world, int dist_min_rows, int rank) {
    const pgo_mg::Hierarchy& H = *(pgo_mg::Hierarchy*)h;
    std::vector<uint64_t> m(masks, masks + N);
    pgo_mg::Owners O; O.touch_mask = &m; O.world = world; O.dist_min_rows = dist_min_rows;      // (the plans read the ownership ranges of the hierarchy, not the owner array)
    PlanSet* P = new PlanSet();
    pgo_mg::build_level_plans(H, O, rank, P->plans);
    std::vector<int32_t> l2g;
    for (long long g = 0; g < N; ++g) if ((m[g] >> rank) & 1) l2g.push_back((int32_t)g);
    pgo_mg::build_fine_plan(m, l2g, rank, world, P->fine);
    return P;
}
void mgh_plans_free(void* ps) { delete (PlanSet*)ps; }
static const pgo_mg::ExchangePlan& plan_of(void* ps, int l) { PlanSet* P = (PlanSet*)ps; return l < 0 ? P->fine.x : P->plans[l]; }      // l = -1: the keyframes' plan
void mgh_plan_sizes(void* ps, int l, long long* n_send, long long* n_recv) { const pgo_mg::ExchangePlan& X = plan_of(ps, l); *n_send = X.n_send(); *n_recv = X.n_recv(); }
void mgh_plan_get(void* ps, int l, int* send_idx, long long* send_off, int* recv_idx, long long* recv_off, long long* pair_cnt) {
    const pgo_mg::ExchangePlan& X = plan_of(ps, l);
    if (!X.send_idx.empty()) std::memcpy(send_idx, X.send_idx.data(), X.send_idx.size() * 4);
    if (!X.recv_idx.empty()) std::memcpy(recv_idx, X.recv_idx.data(), X.recv_idx.size() * 4);
    std::memcpy(send_off, X.send_off.data(), X.send_off.size() * 8); std::memcpy(recv_off, X.recv_off.data(), X.recv_off.size() * 8);
    std::memcpy(pair_cnt, X.pair_cnt.data(), X.pair_cnt.size() * 8);
}
void mgh_fine_sizes(void* ps, long long* n_shared, long long* n_src) { PlanSet* P = (PlanSet*)ps; *n_shared = (long long)P->fine.sh_loc.size(); *n_src = (long long)P->fine.sum_src.size(); }
void mgh_fine_get(void* ps, int* sh_loc, int* sum_ptr, int* sum_src) {
    PlanSet* P = (PlanSet*)ps;
    if (!P->fine.sh_loc.empty()) std::memcpy(sh_loc, P->fine.sh_loc.data(), P->fine.sh_loc.size() * 4);
    std::memcpy(sum_ptr, P->fine.sum_ptr.data(), P->fine.sum_ptr.size() * 4);
    if (!P->fine.sum_src.empty()) std::memcpy(sum_src, P->fine.sum_src.data(), P->fine.sum_src.size() * 4);
}

// ---- the distributed set-up's block plans (pgo_mg_host.hpp: build_setup_plans) ----
void* mgh_setup_plans(void* h, int rank, int world, long long Er, const int* rc1, const int* rc2, const long long* rel_off, long long Es, const int* sc1, const int* sc2, const long long* sw_off) {
    const pgo_mg::Hierarchy& H = *(pgo_mg::Hierarchy*)h;
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<int64_t> ro(rel_off, rel_off + world + 1), so(sw_off, sw_off + world + 1);
    pgo_mg::SetupPlans* S = new pgo_mg::SetupPlans();
    pgo_mg::build_setup_plans(H, rank, world, a, b, ro, c, d, so, *S);
    return S;
}
void mgh_setup_free(void* sp) { delete (pgo_mg::SetupPlans*)sp; }
int mgh_setup_first_whole(void* sp) { return ((pgo_mg::SetupPlans*)sp)->first_whole; }
static const pgo_mg::ExchangePlan& setup_plan(void* sp, int kind, int l) { pgo_mg::SetupPlans* S = (pgo_mg::SetupPlans*)sp; return kind == 0 ? S->val[l].x : kind == 1 ? S->ps[l] : S->rv[l]; }
void mgh_setup_sizes(void* sp, int kind /* 0 the level's blocks (sum), 1 Ps, 2 R */, int l, long long* n_send, long long* n_recv, long long* n_dst, long long* n_src) {
    const pgo_mg::ExchangePlan& X = setup_plan(sp, kind, l);
    *n_send = X.n_send(); *n_recv = X.n_recv(); *n_dst = 0; *n_src = 0;
    if (kind == 0) { const pgo_mg::BlockPlan& B = ((pgo_mg::SetupPlans*)sp)->val[l]; *n_dst = (long long)B.dst.size(); *n_src = (long long)B.sum_src.size(); }
}
void mgh_setup_get(void* sp, int kind, int l, int* send_idx, long long* send_off, int* recv_idx, long long* recv_off, long long* pair_cnt, int* dst, int* sum_ptr, int* sum_src) {
    const pgo_mg::ExchangePlan& X = setup_plan(sp, kind, l);
    if (!X.send_idx.empty()) std::memcpy(send_idx, X.send_idx.data(), X.send_idx.size() * 4);
    if (!X.recv_idx.empty()) std::memcpy(recv_idx, X.recv_idx.data(), X.recv_idx.size() * 4);
    std::memcpy(send_off, X.send_off.data(), X.send_off.size() * 8); std::memcpy(recv_off, X.recv_off.data(), X.recv_off.size() * 8);
    std::memcpy(pair_cnt, X.pair_cnt.data(), X.pair_cnt.size() * 8);
    if (kind == 0) {
        const pgo_mg::BlockPlan& B = ((pgo_mg::SetupPlans*)sp)->val[l];
        if (!B.dst.empty()) std::memcpy(dst, B.dst.data(), B.dst.size() * 4);
        std::memcpy(sum_ptr, B.sum_ptr.data(), B.sum_ptr.size() * 4);
        if (!B.sum_src.empty()) std::memcpy(sum_src, B.sum_src.data(), B.sum_src.size() * 4);
    }
}
long long mgh_setup_prod_size(void* sp, int l) { return (long long)((pgo_mg::SetupPlans*)sp)->prod[l].size(); }
void mgh_setup_prod_get(void* sp, int l, int* out) { const std::vector<int32_t>& v = ((pgo_mg::SetupPlans*)sp)->prod[l]; if (!v.empty()) std::memcpy(out, v.data(), v.size() * 4); }
}
