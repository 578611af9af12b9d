// Test shim: the host-side hierarchy builder of the aggregation multigrid (csrc/pgo_mg_host.hpp) behind a C interface, so that the CPU
// suite can check its invariants.  Host logic only; the cycle itself runs in HIP kernels.
#include <cstring>
#include "pgo_mg_host.hpp"

extern "C" {
void* mgh_build(long long N, const unsigned char* node_free, long long Er, const int* rc1, const int* rc2, const double* rw, long long Es, const int* sc1, const int* sc2,
                int passes0, int passes, int dense_max, int tile_rows, int max_levels, int level0_follows_switchable) {
    std::vector<uint8_t> nf(node_free, node_free + N);
    std::vector<int32_t> a(rc1, rc1 + Er), b(rc2, rc2 + Er), c(sc1, sc1 + Es), d(sc2, sc2 + Es);
    std::vector<double> meas((size_t)Er * 8, 0.0);
    for (long long e = 0; e < Er; ++e) meas[8 * e + 7] = rw[e];
    pgo_mg::Hierarchy* H = new pgo_mg::Hierarchy();
    if (!pgo_mg::build_hierarchy(N, nf, a, b, meas.data(), c, d, nullptr, passes0, passes, dense_max, tile_rows, max_levels, *H, level0_follows_switchable != 0)) { delete H; return nullptr; }
    return H;
}
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
void mgh_level0(void* h, int* agg0, int* mem0_ptr, int* mem0) {
    const pgo_mg::Hierarchy& H = *(pgo_mg::Hierarchy*)h;
    std::memcpy(agg0, H.agg0.data(), H.agg0.size() * 4); std::memcpy(mem0_ptr, H.mem0_ptr.data(), H.mem0_ptr.size() * 4); std::memcpy(mem0, H.mem0.data(), H.mem0.size() * 4);
}
}
