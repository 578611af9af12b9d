// pgo_mg_kernels.hpp — device side of the aggregation-multigrid preconditioner (MgDev / MgLevelDev, pgo_internal.hpp; hierarchy:
// pgo_mg_host.hpp).  Included at the end of pgo_kernels.hip, inside namespace pgo (it uses that file's reductions, the block-CSR row
// product and the rigid-mode Galerkin entry).  gfx950, fp64 VALU.
//
// Per PCG iteration, after cg_update has left the new residual r and z = D^-1 r:
//   restrict0   r_1 = P_0^T r (gather over the keyframes of every level-1 aggregate), x_1 = w D_1^-1 r_1
//   down(l)     t = r_l - A_l x_l ; r_{l+1} = P_l^T t ; x_{l+1} = w D_{l+1}^-1 r_{l+1}          l = 1 .. n_levels-1 (one kernel per level)
//   dense       x_top = A_top^-1 r_top (explicit inverse) ; xt = x + s P x_top on the level below   (s = mg_correction_scale)
//   up(l)       x_l = xt_l + w D_l^-1 (r_l - A_l xt_l) ; xt_{l-1} = x_{l-1} + P x_l                  l = n_levels-1 .. 1
//   prolong0    z += P_0 x_1, r.z partials updated in cg_update's slots
// i.e. 2 n_levels + 1 small kernels — 2 n_levels - 1 in the product's default form: restrict0 rides in the PCG's vector update (cg_update_mg_kernel) and prolong0 in
// level 1's up-sweep; every sum runs in a fixed order (bitwise reproducible).  The level kernels work on workgroup tiles
// of whole aggregates (<= 32 rows, members of an aggregate are contiguous by construction), so restriction and prolongation never leave
// the workgroup.  All of it is latency-bound (the levels hold 17 %, 6 %, 2 % ... of the keyframes): what counts is the kernel count.

__device__ __forceinline__ int bsr_idx(int row, int col) { return (row >> 1) * 12 + col * 2 + (row & 1); }

// One block row times x for the lane owning column c (the block-CSR PCG's row product).  Measured alternatives that changed nothing (a level
// kernel stays at 8-10 us whatever its size: 40 or 400 workgroups): taking up to 8 blocks at once with every load in flight before the first
// use (one col -> x round trip per row instead of three: 9.0 -> 9.2 us at 164 VGPRs), testing the stop flag only after the first loads.
__device__ __forceinline__ void mg_row_accumulate(int64_t b, int64_t e, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                  const double* __restrict__ x, int c, double* acc) {
    int64_t k = b;
    for (; k + 4 <= e; k += 4) spmv_chunk<4, false>(col + k, val + (size_t)k * 36, c, x, nullptr, 0.0, acc);
    if (k + 2 <= e) { spmv_chunk<2, false>(col + k, val + (size_t)k * 36, c, x, nullptr, 0.0, acc); k += 2; }
    if (k < e) spmv_chunk<1, false>(col + k, val + (size_t)k * 36, c, x, nullptr, 0.0, acc);
}

// The cycle streams the level matrices in fp32 (valf: half the bytes of its dominant stream; a preconditioner needs no more), accumulating in fp64.
template <int U>
__device__ __forceinline__ void mg_chunk_f32(const int32_t* __restrict__ colp, const float* __restrict__ valp, int c, const double* __restrict__ x, double* acc) {
    int32_t col[U];
    float2 v[U][3];
    double xx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) col[u] = colp[u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float2* vp = reinterpret_cast<const float2*>(valp + (size_t)u * 36) + c;
        v[u][0] = vp[0]; v[u][1] = vp[6]; v[u][2] = vp[12];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = x[(size_t)col[u] * 6 + c];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc[0] += (double)v[u][0].x * xx[u]; acc[1] += (double)v[u][0].y * xx[u]; acc[2] += (double)v[u][1].x * xx[u];
        acc[3] += (double)v[u][1].y * xx[u]; acc[4] += (double)v[u][2].x * xx[u]; acc[5] += (double)v[u][2].y * xx[u];
    }
}
#ifndef PGO_MG_NARROW
// Round 4: the six lanes of a (row, lane group) slot split the part's blocks like this — lane (h, j) = (c / 3, c % 3) takes every second block (h) and
// the column PAIR j of it: three 16-B loads of the fp32 block (in the row-pair-major layout the four entries (rows 2p, 2p+1) x (columns 2j, 2j+1) are one aligned float4) and one 16-B
// load of x, instead of (before: -DPGO_MG_NARROW) one lane per column with three 8-B loads and one 8-B gather per block.  Same bytes, half the memory instructions per lane: the level
// kernels are bound by the issue rate of their scattered loads (one texture-address unit per CU serves ~11 wavefronts of a level kernel), not by the col -> x dependency — three
// variants that requested column indices earlier all LOST 7 us per iteration, this one gains: level kernels 93.3 -> 88.5 us per cycle on C3, 138.6 -> 127.4 us on C4 (20 LM steps
// 1.37 -> 1.30 s), measured by A/B inside one call.  Every lane still ends with partial sums for all six rows, so the gather through LDS is unchanged.
template <int U>
__device__ __forceinline__ void mg_chunk_f32_wide(const int32_t* __restrict__ colp, const float* __restrict__ valp, int j, const double* __restrict__ x, double* acc) {
    int32_t col[U];
    float4 v[U][3];
    double2 xx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) col[u] = colp[2 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float4* vp = reinterpret_cast<const float4*>(valp + (size_t)(2 * u) * 36) + j;
        v[u][0] = vp[0]; v[u][1] = vp[3]; v[u][2] = vp[6];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = reinterpret_cast<const double2*>(x + (size_t)col[u] * 6)[j];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            acc[2 * p] += (double)v[u][p].x * xx[u].x + (double)v[u][p].z * xx[u].y;
            acc[2 * p + 1] += (double)v[u][p].y * xx[u].x + (double)v[u][p].w * xx[u].y;
        }
    }
}
#endif
__device__ __forceinline__ void mg_row_accumulate_f32(int64_t b, int64_t e, const int32_t* __restrict__ col, const float* __restrict__ val,
                                                      const double* __restrict__ x, int c, double* acc) {
#ifndef PGO_MG_NARROW
    {
        const int h = c >= 3 ? 1 : 0, j = c - 3 * h;
        int64_t kw = b + h;                                         // this lane's blocks: b + h, b + h + 2, ...
        for (; kw + 2 < e; kw += 4) mg_chunk_f32_wide<2>(col + kw, val + (size_t)kw * 36, j, x, acc);
        if (kw < e) mg_chunk_f32_wide<1>(col + kw, val + (size_t)kw * 36, j, x, acc);
        return;
    }
#endif
    int64_t k = b;
// (Also measured and dropped: the part's column indices requested together ahead of the unchanged chunks — 127 -> 135 us per multigrid iteration on C3, 229 -> 237 on C4.  Every
    // variant that puts more memory instructions in flight per lane loses: these kernels are bound by the issue rate of their scattered 8-byte gathers, not by the col -> x chain.)
// (Measured and dropped, round 4: the first eight blocks of a part with ALL their column indices requested first, then all blocks and x entries — 128 -> 157 us per multigrid
    // iteration.  More loads in flight per lane means more registers, and these kernels need their 4 waves per SIMD: ~2 800 wavefronts of a level must be resident in one round.)
    for (; k + 4 <= e; k += 4) mg_chunk_f32<4>(col + k, val + (size_t)k * 36, c, x, acc);
    if (k + 2 <= e) { mg_chunk_f32<2>(col + k, val + (size_t)k * 36, c, x, acc); k += 2; }
    if (k < e) mg_chunk_f32<1>(col + k, val + (size_t)k * 36, c, x, acc);
}
// (Measured and dropped in round 4: a 32-B slot descriptor carrying the block range AND the column indices of the group's first six blocks, so that the x gathers are issued with
// the block loads — one dependent round trip less per level kernel on paper; on C3 the multigrid PCG iteration went 133 -> 142 us, with predicated and with branch-free loads alike:
// six blocks' loads in flight per lane cost more than the round trip they save.)
// (A third variant — the first chunk's column indices requested before the kernel's other operands, so that the x gathers do not queue behind operand loads that miss to HBM —
// lost the same 7 us per iteration as the two above.)
#define MG_ROW_PRODUCT(tile_, xvec_) do { if (rowlive) { int kb, ke; mg_split_row(rb, sg, A.seg_shift, kb, ke); mg_row_accumulate_f32(kb, ke, A.col, A.valf, xvec_, c, acc); } } while (0)
// Row products of the level kernels.  192 lanes = 32 (row, lane-group) slots x 6 columns.  With seg_shift = 0 a slot is a row; with seg_shift = s the tile holds
// R = 32 >> s rows and 2^s lane groups share each row, group g streaming the g-th part of its blocks (long rows: Galerkin products of smoothed transitions).
// mg_split_row: this lane's part [b, e) of the row's block range.  mg_gather_row: the row's result for column-lane c, summed over the 6 column lanes and the groups
// in a fixed order (valid for the lanes of group 0).
__device__ __forceinline__ void mg_split_row(int2 rb, int sg, int seg_shift, int& b, int& e) {
    const int len = rb.y - rb.x, part = (len + (1 << seg_shift) - 1) >> seg_shift;
    b = rb.x + sg * part; e = b + part < rb.y ? b + part : rb.y;
    if (b > e) b = e;
}
__device__ __forceinline__ double mg_gather_row(const double* __restrict__ xch, int li, int c, int seg_shift) {
    const int R = MG_TILE_ROWS >> seg_shift;
    double q = 0.0;
    for (int g = 0; g < (1 << seg_shift); ++g) {
        const double* grp = xch + (size_t)((g * R + li) * 6) * 7;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) q += grp[cc * 7 + c];
    }
    return q;
}
// valf <- val, EXACTLY symmetric: a block below the diagonal is the transpose of the rounded block above it, a diagonal block takes its upper
// triangle.  One wavefront per BLOCK (the smoothed Galerkin products have ~50 blocks per row: a wavefront per row ran 0.37 ms on C3's level 2); lane l < 36 owns
// element l of the block (column-pair-major: (row, col) at (row/2)*12 + col*2 + (row&1)).
__global__ __launch_bounds__(256) void mg_val_f32_kernel(MgLevelDev A) {
    const int64_t k = wave_in_grid() + A.su_blk0;    // (the level's set-up share: all of it on one GPU)
    const int lane = threadIdx.x & 63;
    if (k >= A.su_blk1 || lane >= 36) return;
    const int i = A.row_of[k], j = A.col[k];         // (tables: a binary search over rowptr was 14 dependent loads at the head of every wavefront)
    const int64_t kt = A.tr_of[k];                   // the transposed block (j, i); k itself when the pattern does not hold it
    const int pr = lane / 12, rem = lane - pr * 12, col = rem >> 1, row = pr * 2 + (rem & 1);      // element (row, col) of the block
    const int tr = bsr_idx(col, row);                                                               // where (col, row) lives
    double v;
    if (j > i) v = A.val[(size_t)k * 36 + lane];
    else if (j == i) v = row <= col ? A.val[(size_t)k * 36 + lane] : A.val[(size_t)k * 36 + tr];
    else v = kt != k ? A.val[(size_t)kt * 36 + tr] : A.val[(size_t)k * 36 + lane];
    A.valf[(size_t)k * 36 + lane] = (float)v;
}

// (P_i y)[k]:  dtheta_i = y_theta ; dt_i = y_t - 2 d_i x y_theta
__device__ __forceinline__ double mg_prolong_comp(const double* y, const double* d, int k) {
    if (k < 3) return y[k];
    const int a = k - 3, b = a == 2 ? 0 : a + 1, c = b == 2 ? 0 : b + 1;
    return y[k] - 2.0 * (d[b] * y[c] - d[c] * y[b]);
}
// (P_i^T t)[k]:  [t_theta + 2 d_i x t_t ; t_t]
__device__ __forceinline__ double mg_restrict_comp(const double* t, const double* d, int k) {
    if (k >= 3) return t[k];
    const int b = k == 2 ? 0 : k + 1, c = b == 2 ? 0 : b + 1;
    return t[k] + 2.0 * (d[b] * t[3 + c] - d[c] * t[3 + b]);
}

// ---- geometry: positions of the coarse nodes (centroids) and the offsets d of every node to its parent ----
__global__ __launch_bounds__(256) void mg_geometry0_kernel(MgDev M, MgLevelDev A1, const double* __restrict__ pose8) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M.n1) return;
    const int m0 = M.mem0_ptr[a], m1 = M.mem0_ptr[a + 1];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int m = m0; m < m1; ++m) { const double* t = pose8 + (size_t)M.mem0[m] * 8 + 4; sx += t[0]; sy += t[1]; sz += t[2]; }
    const double inv = 1.0 / (double)(m1 - m0);
    sx *= inv; sy *= inv; sz *= inv;
    A1.pos[a * 3] = sx; A1.pos[a * 3 + 1] = sy; A1.pos[a * 3 + 2] = sz;
    for (int m = m0; m < m1; ++m) {
        const int i = M.mem0[m];
        const double* t = pose8 + (size_t)i * 8 + 4;
        M.d0[(size_t)i * 3] = t[0] - sx; M.d0[(size_t)i * 3 + 1] = t[1] - sy; M.d0[(size_t)i * 3 + 2] = t[2] - sz;
    }
}
__global__ __launch_bounds__(256) void mg_geometry_kernel(MgLevelDev A, double* __restrict__ pos_next) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= A.n_next) return;
    const int m0 = A.agg_ptr[a], m1 = A.agg_ptr[a + 1];
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int m = m0; m < m1; ++m) { sx += A.pos[m * 3]; sy += A.pos[m * 3 + 1]; sz += A.pos[m * 3 + 2]; }
    const double inv = 1.0 / (double)(m1 - m0);
    sx *= inv; sy *= inv; sz *= inv;
    pos_next[a * 3] = sx; pos_next[a * 3 + 1] = sy; pos_next[a * 3 + 2] = sz;
    for (int m = m0; m < m1; ++m) { A.d[m * 3] = A.pos[m * 3] - sx; A.d[m * 3 + 1] = A.pos[m * 3 + 1] - sy; A.d[m * 3 + 2] = A.pos[m * 3 + 2] - sz; }
}
void launch_mg_geometry(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st) {
    (void)G;
    hipLaunchKernelGGL(mg_geometry0_kernel, dim3((unsigned)((M.n1 + 255) / 256)), dim3(256), 0, st, M, levels[0], pose8);
    for (int l = 0; l + 1 < M.n_levels; ++l)
        hipLaunchKernelGGL(mg_geometry_kernel, dim3((unsigned)((levels[l].n_next + 255) / 256)), dim3(256), 0, st, levels[l], levels[l + 1].pos);
}
// several ranks: a level-1 node's members live on several ranks; every keyframe is counted by its owner
__global__ __launch_bounds__(256) void mg_geometry0_sum_kernel(MgDev M, MgLevelDev A1, const double* __restrict__ pose8, const double* __restrict__ own) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M.n1) return;
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int m = M.mem0_ptr[a]; m < M.mem0_ptr[a + 1]; ++m) { const int i = M.mem0[m]; const double w = own[i]; const double* t = pose8 + (size_t)i * 8 + 4; sx += w * t[0]; sy += w * t[1]; sz += w * t[2]; }
    A1.pos[a * 3] = sx; A1.pos[a * 3 + 1] = sy; A1.pos[a * 3 + 2] = sz;
}
__global__ __launch_bounds__(256) void mg_geometry0_finish_kernel(MgDev M, MgLevelDev A1, const double* __restrict__ pose8) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= M.n1) return;
    const double inv = M.inv_cnt[a];
    const double sx = A1.pos[a * 3] * inv, sy = A1.pos[a * 3 + 1] * inv, sz = A1.pos[a * 3 + 2] * inv;
    A1.pos[a * 3] = sx; A1.pos[a * 3 + 1] = sy; A1.pos[a * 3 + 2] = sz;
    for (int m = M.mem0_ptr[a]; m < M.mem0_ptr[a + 1]; ++m) {
        const int i = M.mem0[m];
        const double* t = pose8 + (size_t)i * 8 + 4;
        M.d0[(size_t)i * 3] = t[0] - sx; M.d0[(size_t)i * 3 + 1] = t[1] - sy; M.d0[(size_t)i * 3 + 2] = t[2] - sz;
    }
}
void launch_mg_geometry0_sum(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st) {
    hipLaunchKernelGGL(mg_geometry0_sum_kernel, dim3((unsigned)((M.n1 + 255) / 256)), dim3(256), 0, st, M, levels[0], pose8, G.own);
}
void launch_mg_geometry_finish(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st) {
    (void)G;
    hipLaunchKernelGGL(mg_geometry0_finish_kernel, dim3((unsigned)((M.n1 + 255) / 256)), dim3(256), 0, st, M, levels[0], pose8);
    for (int l = 0; l + 1 < M.n_levels; ++l)
        hipLaunchKernelGGL(mg_geometry_kernel, dim3((unsigned)((levels[l].n_next + 255) / 256)), dim3(256), 0, st, levels[l], levels[l + 1].pos);
}

// ---- Galerkin products ----
// level 1 from the keyframe system: one wavefront per block; contributions as in coarse_assemble_kernel (reduced diagonal blocks C.Dtot and,
// per edge, J1^T J2 - c1 c2^T / a — Hoff when the solver has had it formed for this linearisation (one edge-parallel, coalesced pass over K1's Jacobians: one load per
// lane instead of twelve strided ones, the six products added in k2_edge_kernel's order), else recomputed from K1's Jacobians), summed in list order.
// The contribution list of a block is taken eight entries at a time, and every hop of a contribution's chain is issued for the whole chunk before anything waits:
// the fine values (entry -> block) by the 36 owning lanes, the two keyframes' offsets to the aggregates' centroids (entry -> endpoint -> offset) by one lane per
// (contribution, component) into LDS — three dependent round trips per chunk instead of four per pair of contributions.  The products are added in list order (same bits).
template <bool HOFF>
__global__ __launch_bounds__(256) void mg_galerkin0_kernel(GraphDev G, LinDev L, ScaleDev Sc, CgDev C, MgDev M, MgLevelDev A) {
    constexpr int CH = HOFF ? 8 : 2;      // (recomputing J1^T J2 from K1's Jacobians takes twelve strided loads per contribution: two at a time, as before)
    __shared__ double Hs[4][CH][36];
    __shared__ double ds[4][CH][6];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    int64_t slot = wave_in_grid();
    if (slot >= (M.g0_slots ? (int64_t)M.n_g0 : A.nnzb)) return;
    if (M.g0_slots) slot = M.g0_slots[slot];      // several ranks, distributed set-up: the blocks this rank has contributions for (the others are never read here: their parts arrive by the block exchange)
    const int r = lane / 6, c = lane - r * 6;
    const bool own = lane < 36;
    const int du = lane / 6, dm = lane - du * 6;
    double acc = 0.0;
    const int64_t kend = A.g_ptr[slot + 1];
    for (int64_t k0 = A.g_ptr[slot]; k0 < kend; k0 += CH) {
        const int n = (int)(kend - k0 < CH ? kend - k0 : CH);
        if (du < n) {       // lanes 0 .. 6 n - 1: component dm of contribution du's pair of offsets (dm < 3: the row keyframe)
            const int64_t e = A.g_ent[k0 + du];
            const int kind = (int)(e & 7);
            int64_t node = e >> 3;
            if (kind != 0) {
                const bool is_sw = kind >= 3, first = (dm < 3) != (kind == 2 || kind == 4);
                const int32_t* cp = is_sw ? (first ? G.sw.c1 : G.sw.c2) : (first ? G.rel.c1 : G.rel.c2);
                node = cp[node];
            }
            ds[wv][du][dm] = M.d0[(size_t)node * 3 + (dm < 3 ? dm : dm - 3)];
        }
        double h[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) h[u] = fine_block_value<HOFF>(G, L, Sc, C, A.g_ent[k0 + u < kend ? k0 + u : kend - 1], lane, r, c, own);
        if (own) {
#pragma unroll
            for (int u = 0; u < CH; ++u) Hs[wv][u][lane] = h[u];
        }
        __builtin_amdgcn_wave_barrier();
        if (own) {
#pragma unroll
            for (int u = 0; u < CH; ++u) if (u < n) acc += coarse_entry(Hs[wv][u], ds[wv][u], ds[wv][u] + 3, r, c);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (own) A.val[(size_t)slot * 36 + bsr_idx(r, c)] = acc;
}
// level l+1 (B) from level l (A): contributions are blocks of A, entry = (row << 32) | slot; chunks as in mg_galerkin0_kernel
__global__ __launch_bounds__(256) void mg_galerkin_kernel(MgLevelDev A, MgLevelDev B) {
    constexpr int CH = 8;
    __shared__ double Hs[4][CH][36];
    __shared__ double ds[4][CH][6];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    const int64_t slot = wave_in_grid() + B.su_blk0;      // (several ranks: the blocks of the rank's own rows of B — their contributions are blocks of its own rows of A)
    if (slot >= B.su_blk1) return;
    const int r = lane / 6, c = lane - r * 6;
    const bool own = lane < 36;
    // storage offset `lane` of a block holds element (row, col) with row = 2 (lane / 12) + (lane & 1), col = (lane % 12) / 2
    const int srow = 2 * (lane / 12) + (lane & 1), scol = (lane % 12) >> 1;
    const int du = lane / 6, dm = lane - du * 6;
    double acc = 0.0;
    const int64_t kend = B.g_ptr[slot + 1];
    for (int64_t k0 = B.g_ptr[slot]; k0 < kend; k0 += CH) {
        const int n = (int)(kend - k0 < CH ? kend - k0 : CH);
        if (du < n) {
            const int64_t e = B.g_ent[k0 + du];
            const int64_t node = dm < 3 ? (e >> 32) : (int64_t)A.col[e & 0xffffffffll];
            ds[wv][du][dm] = A.d[(size_t)node * 3 + (dm < 3 ? dm : dm - 3)];
        }
        double h[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int64_t fs = B.g_ent[k0 + u < kend ? k0 + u : kend - 1] & 0xffffffffll;
            h[u] = own ? A.val[(size_t)fs * 36 + lane] : 0.0;
        }
        if (own) {
#pragma unroll
            for (int u = 0; u < CH; ++u) Hs[wv][u][srow * 6 + scol] = h[u];
        }
        __builtin_amdgcn_wave_barrier();
        if (own) {
#pragma unroll
            for (int u = 0; u < CH; ++u) if (u < n) acc += coarse_entry(Hs[wv][u], ds[wv][u], ds[wv][u] + 3, r, c);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (own) B.val[(size_t)slot * 36 + bsr_idx(r, c)] = acc;
}
// omega x inverse of every diagonal block (Cholesky; a block that is not positive definite raises *fail)
// `diag` (keyframe level, filtered smoothed transition): the blocks to invert instead of the level's own diagonal blocks — the lumped ones; a lumped block that is not positive
// definite falls back to the level's own block (the lumping keeps A's action on the rigid modes, it does not have to keep every block definite)
__global__ __launch_bounds__(256) void mg_dinv_kernel(MgLevelDev A, double omega, int32_t* __restrict__ fail, int skip_orphans /* keyframe level: rows with parent -1 are outside the system */,
                                                      const double* __restrict__ diag = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + A.su_row0;
    if (i >= A.su_row1) return;
    if (skip_orphans && A.parent[i] < 0) { double* out = A.Dinv + (size_t)i * 36; for (int e = 0; e < 36; ++e) out[e] = 0.0; return; }
    const double* v = diag ? diag + (size_t)i * 36 : A.val + (size_t)A.rowptr[i] * 36;
    double Lm[36];
    bool ok = true;
    for (int attempt = 0; attempt < 2; ++attempt) {
    ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = v[bsr_idx(j, j)];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= Lm[j * 6 + k] * Lm[j * 6 + k];
        ok = ok && (d > 0.0);
        d = sqrt(d);
        Lm[j * 6 + j] = 1.0 / d;                      // the diagonal is kept inverted
#pragma unroll
        for (int r = 0; r < 6; ++r) if (r > j) {
            double s = v[bsr_idx(r, j)];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) s -= Lm[r * 6 + k] * Lm[j * 6 + k];
            Lm[r * 6 + j] = s * Lm[j * 6 + j];
        }
    }
    if (ok || !diag || attempt == 1) break;
    v = A.val + (size_t)A.rowptr[i] * 36;      // the lumped block is not positive definite: the level's own
    }
    if (!ok) atomicOr(fail, 1);
    double* out = A.Dinv + (size_t)i * 36;
#pragma unroll
    for (int e = 0; e < 6; ++e) {                     // column e of the inverse: L L^T z = unit vector e
        double y[6], z[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double s = r == e ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < r) s -= Lm[r * 6 + k] * y[k];
            y[r] = s * Lm[r * 6 + r];
        }
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double s = y[r];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k > r) s -= Lm[k * 6 + r] * z[k];
            z[r] = s * Lm[r * 6 + r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) out[r * 6 + e] = omega * z[r];
    }
}
// coarsest level: its blocks into the dense operator (zeroed, identity-padded by the launcher)
__global__ __launch_bounds__(256) void mg_dense_scatter_kernel(MgLevelDev A, CoarseDev K) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)A.n * 36) return;
    const int i = (int)(t / 36), l = (int)(t - (int64_t)i * 36);
    const int row = 2 * (l / 12) + (l & 1), col = (l % 12) >> 1;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
        K.Ac[(size_t)(i * 6 + row) * K.nc + (size_t)A.col[k] * 6 + col] = A.val[(size_t)k * 36 + l];
}
// ---- smoothed transitions (smoothed aggregation): Ps = (I - c Dinv A) P, W = A Ps, level above = Ps^T W ----
// element (r, c) of B_k, the rigid prolongation block of node k with offset d to its parent: dtheta_k = y_theta ; dt_k = y_t - 2 d x y_theta
__device__ __forceinline__ double mg_pblock(const double* __restrict__ d, int r, int c) {
    if (r < 3 || c >= 3) return r == c ? 1.0 : 0.0;
    const int a = r - 3;                              // X = -2 [d]x
    if (a == c) return 0.0;
    const int o = 3 - a - c;                          // the third axis
    const double sgn = ((a + 1) % 3 == c) ? 1.0 : -1.0;     // X[a][a+1] = +2 d[a+2] ; X[a][a+2] = -2 d[a+1]
    return sgn * 2.0 * d[o];
}
// one wavefront per block (i, a) of Ps; lane l < 36 owns element (l / 6, l % 6).  Row i of A is short: its blocks are scanned for columns whose parent is a.
// FILT (keyframe level, round 6): the FILTERED matrix — blocks of switchable loop closures (kind 3 / 4 in g_ent) do not enter, the diagonal block is the lumped one (A.dlump)
template <bool FILT>
__global__ __launch_bounds__(256) void mg_ps_kernel(MgLevelDev A, double cs) {
    __shared__ double acc_s[4][36];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    const int64_t slot = wave_in_grid() + A.su_ps0;
    if (slot >= A.su_ps1) return;
    const bool own = lane < 36;
    const int r = own ? lane / 6 : 0, c = own ? lane % 6 : 0;
    const int i = A.ps_row[slot], a = A.ps_col[slot];
    double acc = 0.0;
    for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
        const int j = A.col[k];
        if (A.parent[j] != a) continue;
        if (FILT && (A.g_ent[k] & 7) >= 3) continue;
        if (own) {
            const double* dj = A.d + (size_t)j * 3;
            const double* blk = (FILT && k == A.rowptr[i]) ? A.dlump + (size_t)i * 36 : A.val + (size_t)k * 36;
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += blk[bsr_idx(r, m)] * mg_pblock(dj, m, c);
        }
    }
    if (own) acc_s[wv][lane] = acc;
    __builtin_amdgcn_wave_barrier();
    if (own) {
        double v = A.parent[i] == a ? mg_pblock(A.d + (size_t)i * 3, r, c) : 0.0;
        const double* Dk = A.Dinv + (size_t)i * 36 + r * 6;
#pragma unroll
        for (int m = 0; m < 6; ++m) v -= cs * Dk[m] * acc_s[wv][m * 6 + c];
        A.ps_val[(size_t)slot * 36 + lane] = v;
    }
}
// W = A Ps: one wavefront per block (i, b) of W: sum over the blocks k of A's row i whose column's Ps row holds b.  The row is taken MG_SETUP_CHUNK blocks at a time
// and every hop of the chain column -> Ps row range -> Ps columns (searched by ballot) -> Ps block is issued for the whole chunk before the next one starts: four dependent
// round trips per chunk instead of four per block of A (the kernel is nothing but these chains: 0.50 -> see DESIGN.md).  The products are added in the order of A's row, as before (same bits).
constexpr int MG_SETUP_CHUNK = 8;
__global__ __launch_bounds__(256) void mg_w_kernel(MgLevelDev A) {
    __shared__ double pb[4][MG_SETUP_CHUNK][36], ab[4][MG_SETUP_CHUNK][36];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    const int64_t slot = wave_in_grid() + A.su_w0;
    if (slot >= A.su_w1) return;
    const bool own = lane < 36;
    const int r = own ? lane / 6 : 0, c = own ? lane % 6 : 0;
    const int i = A.w_row[slot], b = A.w_col[slot];
    double acc = 0.0;
    const int64_t kend = A.rowptr[i + 1];
    for (int64_t k0 = A.rowptr[i]; k0 < kend; k0 += MG_SETUP_CHUNK) {
        const int n = (int)(kend - k0 < MG_SETUP_CHUNK ? kend - k0 : MG_SETUP_CHUNK);
        int j[MG_SETUP_CHUNK], p0[MG_SETUP_CHUNK], p1[MG_SETUP_CHUNK], ps[MG_SETUP_CHUNK];
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) j[u] = u < n ? A.col[k0 + u] : 0;
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) { p0[u] = u < n ? A.ps_rowptr[j[u]] : 0; p1[u] = u < n ? A.ps_rowptr[j[u] + 1] : 0; }
        // the Ps rows are short (the parents of one row's columns): one ballot per row finds b; a longer row takes further steps (uniform over the wavefront)
        int cand[MG_SETUP_CHUNK];
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) cand[u] = (p0[u] + lane < p1[u]) ? A.ps_col[p0[u] + lane] : -1;
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) {
            unsigned long long hit = __ballot(cand[u] == b);
            ps[u] = hit ? p0[u] + __ffsll((long long)hit) - 1 : -1;
            for (int base = p0[u] + 64; ps[u] < 0 && base < p1[u]; base += 64) {
                hit = __ballot(base + lane < p1[u] && A.ps_col[base + lane] == b);
                if (hit) ps[u] = base + __ffsll((long long)hit) - 1;
            }
        }
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) if (ps[u] >= 0 && own) pb[wv][u][lane] = A.ps_val[(size_t)ps[u] * 36 + lane];
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) if (ps[u] >= 0 && own) ab[wv][u][lane] = A.val[(size_t)(k0 + u) * 36 + lane];      // (storage order of the block-CSR blocks)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) {
            if (ps[u] >= 0 && own) {
#pragma unroll
                for (int m = 0; m < 6; ++m) acc += ab[wv][u][bsr_idx(r, m)] * pb[wv][u][m * 6 + c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (own) A.w_val[(size_t)slot * 36 + lane] = acc;
}
// Explicit transfer operator of a smoothed transition: R^T[i, b] = Ps[i, b] - Dinv_i W[i, b] for every block of W (Ps[i, b] = 0 outside its own, smaller pattern), rounded to
// fp32 ONCE and stored twice — on W's block-CSR for the prolongation x = v + R^T x_next, transposed by coarse row for the restriction r_next = R r: the two are exact
// transposes of each other, so the cycle stays symmetric.  One wavefront per block, lane l < 36 owns element (l / 6, l % 6); both copies in the level kernels' block layout (bsr_idx).
__global__ __launch_bounds__(256) void mg_rt_kernel(MgLevelDev A) {
    const int lane = threadIdx.x & 63;
    const int64_t k = wave_in_grid() + A.su_w0;
    if (k >= A.su_w1 || lane >= 36) return;
    const int r = lane / 6, c = lane % 6;
    const int i = A.w_row[k];
    const int32_t ps = A.ps_of_w[k];
    const int32_t tr = A.rT_of_w[k];
    double v = ps >= 0 ? A.ps_val[(size_t)ps * 36 + lane] : 0.0;
    const double* Dk = A.Dinv + (size_t)i * 36 + r * 6;
    const double* W = A.w_val + (size_t)k * 36;
#pragma unroll
    for (int m = 0; m < 6; ++m) v -= Dk[m] * W[m * 6 + c];
    const float f = (float)v;
    A.rt_valf[(size_t)k * 36 + bsr_idx(r, c)] = f;
    A.r_valf[(size_t)tr * 36 + bsr_idx(c, r)] = f;
}
// level above: block (a, b) = sum over the rows i with Ps[i, a] != 0 of Ps[i, a]^T W[i, b] — the rows of Ps's column a in chunks, every hop (entry -> W row range -> W columns,
// searched by ballot -> the two blocks) issued for the whole chunk at once, the products added in list order (same bits as the entry-at-a-time loop)
__global__ __launch_bounds__(256) void mg_psTw_kernel(MgLevelDev A, MgLevelDev B) {
    __shared__ double pa[4][MG_SETUP_CHUNK][36], wb[4][MG_SETUP_CHUNK][36];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    // several ranks (A.su_prod): the blocks this rank's rows of Ps / W contribute to, and only those rows' part of the sum — the parts are added where the block is needed
    int64_t slot = wave_in_grid();
    if (slot >= (A.su_prod ? (int64_t)A.n_su_prod : B.nnzb)) return;
    if (A.su_prod) slot = A.su_prod[slot];
    const bool own = lane < 36;
    const int r = own ? lane / 6 : 0, c = own ? lane % 6 : 0;
    const int a = B.row_of[slot], b = B.col[slot];
    double acc = 0.0;
    const int64_t eend = A.psT_ptr[a + 1];
    for (int64_t e0 = A.psT_ptr[a]; e0 < eend; e0 += MG_SETUP_CHUNK) {
        const int n = (int)(eend - e0 < MG_SETUP_CHUNK ? eend - e0 : MG_SETUP_CHUNK);
        int64_t ent[MG_SETUP_CHUNK];
        int w0[MG_SETUP_CHUNK], w1[MG_SETUP_CHUNK], ws[MG_SETUP_CHUNK], cand[MG_SETUP_CHUNK];
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) ent[u] = u < n ? A.psT_ent[e0 + u] : 0;
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) { const int i = (int)(ent[u] >> 32); const bool in = u < n && i >= A.su_row0 && i < A.su_row1; w0[u] = in ? A.w_rowptr[i] : 0; w1[u] = in ? A.w_rowptr[i + 1] : 0; }
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) cand[u] = (w0[u] + lane < w1[u]) ? A.w_col[w0[u] + lane] : -1;
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) {
            unsigned long long hit = __ballot(cand[u] == b);
            ws[u] = hit ? w0[u] + __ffsll((long long)hit) - 1 : -1;
            for (int base = w0[u] + 64; ws[u] < 0 && base < w1[u]; base += 64) {      // rows of W longer than a wavefront
                hit = __ballot(base + lane < w1[u] && A.w_col[base + lane] == b);
                if (hit) ws[u] = base + __ffsll((long long)hit) - 1;
            }
        }
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u)
            if (ws[u] >= 0 && own) { pa[wv][u][lane] = A.ps_val[(size_t)(ent[u] & 0xffffffffll) * 36 + lane]; wb[wv][u][lane] = A.w_val[(size_t)ws[u] * 36 + lane]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < MG_SETUP_CHUNK; ++u) {
            if (ws[u] >= 0 && own) {
#pragma unroll
                for (int m = 0; m < 6; ++m) acc += pa[wv][u][m * 6 + r] * wb[wv][u][m * 6 + c];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (own) B.val[(size_t)slot * 36 + bsr_idx(r, c)] = acc;
}
// The smoothed prolongator inside the cycle, implicitly: one row product with the level's own matrix.
//   w = in_r - A in_x  (in_r null: 0) ;  out_w = w (optional) ;  out = add1 + add2 + cs Dinv w   (either add may be null; Dinv holds omega D^-1, cs = w_p / omega)
// before the restriction:  t = r - A x_pre,  u = cs Dinv t,  then the restriction kernel forms P^T (t - A u) = Ps^T t
// after the prolongation:  e = P x_next,  y = x_pre + e - cs Dinv (A e) = x_pre + Ps x_next,  then the post-smoothing kernel runs on y
__device__ __forceinline__ void mg_smooth_step_tile(const MgLevelDev& A, int tile, const double* __restrict__ in_r, const double* __restrict__ in_x, double* __restrict__ out_w,
                                                    const double* __restrict__ add1, const double* __restrict__ add2, double* __restrict__ out, double cs, const int32_t* __restrict__ stop,
                                                    double* xch /* CG_BLOCK x 7 */, double* tb /* CG_BLOCK */) {
    const int stopped = stop ? *stop : 0;
    const int q6 = threadIdx.x / 6, c = threadIdx.x % 6;
    const int li = q6 & ((MG_TILE_ROWS >> A.seg_shift) - 1), sg = q6 >> (5 - A.seg_shift);
    const int4 ti = A.tile_info[tile];
    const int2 rb = A.tile_rows[tile * MG_TILE_ROWS + li];
    if (stopped) return;
    const int i0 = ti.z, i1 = ti.w;
    const int row = i0 + li;
    const bool rowlive = row < i1;
    const bool live = rowlive && sg == 0;
    double rv = 0.0, av = 0.0;
    double Dk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (live) {
        if (in_r) rv = in_r[(size_t)row * 6 + c];
        if (add1) av += add1[(size_t)row * 6 + c];
        if (add2) av += add2[(size_t)row * 6 + c];
        const double2* Dp = reinterpret_cast<const double2*>(A.Dinv + (size_t)row * 36 + c * 6);
        const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2];
        Dk[0] = u0.x; Dk[1] = u0.y; Dk[2] = u1.x; Dk[3] = u1.y; Dk[4] = u2.x; Dk[5] = u2.y;
    }
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    MG_ROW_PRODUCT(tile, in_x);
    double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
    for (int q = 0; q < 6; ++q) mine[q] = acc[q];
    __syncthreads();
    double w = 0.0;
    if (live) {
        w = rv - mg_gather_row(xch, li, c, A.seg_shift);
        tb[threadIdx.x] = w;
        if (out_w) out_w[(size_t)row * 6 + c] = w;
    }
    __syncthreads();
    if (live) {
        const double* ta = tb + (threadIdx.x - c);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x += Dk[j] * ta[j];
        out[(size_t)row * 6 + c] = av + cs * x;
    }
}
__global__ __launch_bounds__(CG_BLOCK) void mg_smooth_step_kernel(MgLevelDev A, const double* __restrict__ in_r, const double* __restrict__ in_x, double* __restrict__ out_w,
                                                                   const double* __restrict__ add1, const double* __restrict__ add2, double* __restrict__ out, double cs, const int32_t* __restrict__ stop) {
    __shared__ double xch[CG_BLOCK * 7];
    __shared__ double tb[CG_BLOCK];
    mg_smooth_step_tile(A, (int)blockIdx.x + A.tile0, in_r, in_x, out_w, add1, add2, out, cs, stop, xch, tb);
}
// Down-sweep of a level with a smoothed transition above it, explicit form (MgLevelDev::rt_valf; pgo_mg_host.hpp) — ONE launch, two independent kinds of workgroups:
//   the level's own tiles:           v = x_pre + Dinv (r - A x_pre)                                  -> A.y   (the smoothing step; the up-sweep only adds R^T x_next to it)
//   tiles of consecutive coarse rows: r_next = R r  (R = (Ps - Dinv W)^T, fp32 blocks by coarse row)  -> r_next, and x_next = Dinv_next r_next when the level above is a sparse one
// instead of the implicit form's two dependent launches (t = r - A x_pre, u = c Dinv t;  r_next = P^T (t - A u)): the same r_next = Ps^T (r - A x_pre) algebraically.
__global__ __launch_bounds__(CG_BLOCK) void mg_sdown_kernel(MgLevelDev A, double* __restrict__ r_next, double* __restrict__ x_next, const double* __restrict__ Dinv_next, const int32_t* __restrict__ stop) {
    __shared__ double xch[CG_BLOCK * 7];
    __shared__ double tb[CG_BLOCK];
    if ((int)blockIdx.x < A.tiles_own) { mg_smooth_step_tile(A, (int)blockIdx.x + A.tile0, A.r, A.x, nullptr, A.x, nullptr, A.y, 1.0, stop, xch, tb); return; }
    const int stopped = stop ? *stop : 0;
    const int tile = (int)blockIdx.x - A.tiles_own;
    const int ss = A.rT_seg_shift, rpt = MG_TILE_ROWS >> ss;
    const int q6 = threadIdx.x / 6, c = threadIdx.x % 6;
    const int li = q6 & (rpt - 1), sg = q6 >> (5 - ss);
    const int2 rb = A.rT_rows[tile * MG_TILE_ROWS + li];
    const int row = A.rT_row0 + tile * rpt + li;      // (several ranks: the rank's own coarse rows; rT_rows is the table of that range)
    const bool live = row < A.rT_row1 && sg == 0;
    double Dk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (live && x_next) {
        const double2* Dp = reinterpret_cast<const double2*>(Dinv_next + (size_t)row * 36 + c * 6);
        const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2];
        Dk[0] = u0.x; Dk[1] = u0.y; Dk[2] = u1.x; Dk[3] = u1.y; Dk[4] = u2.x; Dk[5] = u2.y;
    }
    if (stopped) return;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (row < A.rT_row1) { int kb, ke; mg_split_row(rb, sg, ss, kb, ke); mg_row_accumulate_f32(kb, ke, A.rT_col, A.r_valf, A.r, c, acc); }
    double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
    for (int q = 0; q < 6; ++q) mine[q] = acc[q];
    __syncthreads();
    double sum = 0.0;
    if (live) { sum = mg_gather_row(xch, li, c, ss); r_next[(size_t)row * 6 + c] = sum; }
    if (!x_next) return;
    tb[threadIdx.x] = sum;
    __syncthreads();
    if (live) {
        const double* ra = tb + (threadIdx.x - c);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x += Dk[j] * ra[j];
        x_next[(size_t)row * 6 + c] = x;
    }
}

// ---- smoothed transition keyframes -> level 1 (round 5, experimental: pgo_options::mg_smoothed_fine) ----
// The keyframe level as a block-CSR level F of its own (rowptr / col: one block per keyframe and per incident edge; g_ent[k] = the contribution block k IS: (index << 3) | kind as
// in level 1's lists), so that the set-up kernels of a smoothed transition (mg_ps_kernel, mg_w_kernel, mg_psTw_kernel) form Ps_0 = (I - c Dinv A) P_0, W_0 = A Ps_0 and level 1 =
// Ps_0^T W_0 exactly as they do one level up.  Inside the cycle the keyframe level stays matrix-free and additive:  z = D^-1 r + s Ps_0 V_1(Ps_0^T r)  — Ps_0 as fp32 blocks by
// keyframe row (prolongation) and, the same rounded numbers transposed, by level-1 row (restriction): ~2 blocks per keyframe each way.
template <bool HOFF>
__global__ __launch_bounds__(256) void mg_fine_blocks_kernel(GraphDev G, LinDev L, ScaleDev Sc, CgDev C, MgLevelDev F) {
    const int lane = threadIdx.x & 63;
    const int64_t k = wave_in_grid();
    if (k >= F.nnzb) return;
    const int r = lane / 6, c = lane - r * 6;
    const bool own = lane < 36;
    const double h = fine_block_value<HOFF>(G, L, Sc, C, F.g_ent[k], lane, r, c, own);
    if (own) F.val[(size_t)k * 36 + bsr_idx(r, c)] = h;
}
// Filtered smoothed keyframe transition (round 6): the diagonal blocks of the filtered matrix A_f (odometry blocks kept, switchable loop closures dropped) with the dropped blocks
// LUMPED in so that A_f keeps A's action on the rigid-body modes:  A_f,ii = A_ii + sym( sum over dropped j of A_ij T_ji ),  T_ji = the rigid motion seen at keyframe j for a
// motion of keyframe i (dtheta_j = dtheta_i, dt_j = dt_i - 2 [t_j - t_i]x dtheta_i).  Without the lumping the filtered prolongator is useless (CPU probe
// scripts/research/r5_filtered_fine_probe.py: 1 878 PCG iterations against 146 with it, 127 unfiltered, 265 with the plain transition).  One wavefront per keyframe, lane l < 36 owns
// element (l / 6, l % 6); blocks in the level's layout (bsr_idx).
__global__ __launch_bounds__(256) void mg_fine_lump_kernel(MgLevelDev F, const double* __restrict__ pose8) {
    __shared__ double cs_[4][36];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    const int64_t i = wave_in_grid();
    if (i >= F.n) return;
    const bool own = lane < 36;
    const int r = own ? lane / 6 : 0, c = own ? lane % 6 : 0;
    const int64_t k0 = F.rowptr[i], k1 = F.rowptr[i + 1];
    double acc = 0.0;
    if (F.parent[i] >= 0) {
        const double* ti = pose8 + (size_t)i * 8 + 4;
        for (int64_t k = k0 + 1; k < k1; ++k) {
            if ((F.g_ent[k] & 7) < 3) continue;
            const int j = F.col[k];
            if (F.parent[j] < 0 || j == (int)i) continue;
            if (own) {
                const double* tj = pose8 + (size_t)j * 8 + 4;
                const double d[3] = {tj[0] - ti[0], tj[1] - ti[1], tj[2] - ti[2]};
#pragma unroll
                for (int m = 0; m < 6; ++m) acc += F.val[(size_t)k * 36 + bsr_idx(r, m)] * mg_pblock(d, m, c);
            }
        }
    }
    if (own) cs_[wv][lane] = acc;
    __builtin_amdgcn_wave_barrier();
    if (own) F.dlump[(size_t)i * 36 + bsr_idx(r, c)] = F.val[(size_t)k0 * 36 + bsr_idx(r, c)] + 0.5 * (cs_[wv][r * 6 + c] + cs_[wv][c * 6 + r]);
}
// Ps_0 rounded to fp32 ONCE and stored twice (T = the transfer view of the keyframe level: rt_valf on Ps's own pattern by keyframe row, r_valf by level-1 row, rT_of_w = slot of
// block k there): exact transposes of each other, so the preconditioner stays symmetric.  Both in the level kernels' block layout.
__global__ __launch_bounds__(256) void mg_ps_f32_kernel(MgLevelDev T) {
    const int lane = threadIdx.x & 63;
    const int64_t k = wave_in_grid();
    if (k >= T.n_ps || lane >= 36) return;
    const int r = lane / 6, c = lane % 6;
    const float f = (float)T.ps_val[(size_t)k * 36 + lane];
    T.rt_valf[(size_t)k * 36 + bsr_idx(r, c)] = f;
    T.r_valf[(size_t)T.rT_of_w[k] * 36 + bsr_idx(c, r)] = f;
}
// z_i += s (Ps_0 x_1)_i and r.z += r.(s Ps_0 x_1), in cg_update's lane / workgroup mapping (same partial-sum slots, as mg_prolong0_kernel): lane = (keyframe, row pair j);
// rows 2j, 2j+1 of a block are 12 consecutive floats of the block layout
__global__ __launch_bounds__(CG_BLOCK) void mg_prolong0s_kernel(GraphDev G, MgLevelDev T, const double* __restrict__ x1, const double* __restrict__ rv, double* __restrict__ zv,
                                                                 double* __restrict__ part_rz, double scale, const int32_t* __restrict__ stop) {
    __shared__ double red[CG_BLOCK / 64];
    if (stop && *stop) return;
    const int64_t pairs = G.N * 3;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * CG_BLOCK) {
        const int64_t n = i / 3;
        const int j = (int)(i - n * 3);
        const int k0 = T.ps_rowptr[n], k1 = T.ps_rowptr[n + 1];
        if (k1 <= k0) continue;
        double a0 = 0.0, a1 = 0.0;
        for (int k = k0; k < k1; ++k) {
            const float4* vp = reinterpret_cast<const float4*>(T.rt_valf + (size_t)k * 36 + j * 12);
            const float4 v0 = vp[0], v1 = vp[1], v2 = vp[2];      // (row 2j, col 0) (row 2j+1, col 0) (2j, 1) (2j+1, 1) | cols 2, 3 | cols 4, 5
            const double2* xp = reinterpret_cast<const double2*>(x1 + (size_t)T.ps_col[k] * 6);
            const double2 xa = xp[0], xb = xp[1], xc = xp[2];
            a0 += (double)v0.x * xa.x + (double)v0.z * xa.y + (double)v1.x * xb.x + (double)v1.z * xb.y + (double)v2.x * xc.x + (double)v2.z * xc.y;
            a1 += (double)v0.y * xa.x + (double)v0.w * xa.y + (double)v1.y * xb.x + (double)v1.w * xb.y + (double)v2.y * xc.x + (double)v2.w * xc.y;
        }
        a0 *= scale; a1 *= scale;
        double2* zp = reinterpret_cast<double2*>(zv) + i;
        const double2 r = reinterpret_cast<const double2*>(rv)[i];
        double2 z = *zp;
        z.x += a0; z.y += a1;
        *zp = z;
        acc += r.x * a0 + r.y * a1;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part_rz[blockIdx.x] += s;
}

// ---- smoother safety: the damped block-Jacobi smoother needs w lambda_max(D^-1 A) < 2 or the cycle is not positive definite any more (the PCG then "converges" on a
// negative r.z: measured on a chain-like graph whose smoothed Galerkin level has lambda_max ~ 2.5).  lambda_max is estimated per level and per LM system by a few steps
// of the power method (a lower bound: hence the margins) and a level whose w lambda exceeds `limit` gets its Dinv = w D^-1 scaled down to w lambda = `target`.
__global__ void mg_power_init_kernel(double* __restrict__ v, double* __restrict__ w, int n6, int lo6, int hi6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n6) return;
    v[i] = (i >= lo6 && i < hi6) ? 1.0 + 0.5 * sin(0.7 * (double)i) : 0.0;          // fixed, not orthogonal to anything in particular; zero outside the rank's own rows (and they stay zero)
    w[i] = 0.0;
}
// one workgroup: lam = ||b|| / ||a||  (b = D^-1 A a after several un-normalised steps: lambda_max^8 ~ 10^3, far from overflow)
__global__ __launch_bounds__(1024) void mg_power_ratio_kernel(const double* __restrict__ a, const double* __restrict__ b, int lo6, int hi6, double* __restrict__ lam) {
    __shared__ double red[32];
    double sa = 0.0, sb = 0.0;
    for (int i = lo6 + threadIdx.x; i < hi6; i += blockDim.x) { sa += a[i] * a[i]; sb += b[i] * b[i]; }      // (the set-up's rows: outside them both vectors are zero)
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sa; red[16 + (threadIdx.x >> 6)] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int k = 0; k < 16; ++k) { ta += red[k]; tb += red[16 + k]; }
        *lam = ta > 0.0 ? sqrt(tb / ta) : 0.0;
    }
}
__global__ __launch_bounds__(256) void mg_rescale_dinv_kernel(MgLevelDev A, const double* __restrict__ lam, double omega, double limit, double target) {
    const double wl = omega * lam[0];
    if (!(wl > limit)) return;
    const double f = target / wl;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + (int64_t)A.su_row0 * 36;
    if (i < (int64_t)A.su_row1 * 36) A.Dinv[i] *= f;
}
// The whole level (one GPU, and levels every rank sets up completely).  A distributed level runs the same eight steps on the owners' rows with the halo of the iterate exchanged
// before every step and the two norms all-reduced (pgo_solver.hip: build_mg_ranks, through the pieces below) — the same estimate up to the order of the sums.  (First tried: every
// rank on its own diagonal part of D^-1 A, the estimates maximised — a lower bound as well, no exchanges; scripts/gpu_ranks_soak.py case 36 — 6 010 keyframes on 3 ranks by index
// ranges, two smoothed transitions — then missed a rescaling the whole-level estimate triggers and the PCG broke down on a preconditioner that was not positive definite.)
void launch_mg_level_power(const MgLevelDev& A, double omega, hipStream_t st) {
    const int n6 = A.n * 6;
    double* v = A.x; double* w = A.xt; double* lam = A.xf;              // the level's cycle vectors are free during the set-up
    MgLevelDev S = A;                                                   // the rows the set-up works on, as tiles: the cycle's share on a distributed level, else every tile
    const bool part = A.su_row0 != 0 || A.su_row1 != A.n;
    if (!part) { S.tile0 = 0; S.tiles_own = A.tiles; }
    hipLaunchKernelGGL(mg_power_init_kernel, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, st, v, w, n6, A.su_row0 * 6, A.su_row1 * 6);
    for (int it = 0; it < 8; ++it) {
        // w = D^-1 A v  =  (-1 / omega) (omega D^-1) (0 - A v)
        if (S.tiles_own > 0) hipLaunchKernelGGL(mg_smooth_step_kernel, dim3((unsigned)S.tiles_own), dim3(CG_BLOCK), 0, st, S, (const double*)nullptr, (const double*)v, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, w, -1.0 / omega, (const int32_t*)nullptr);
        double* tmp = v; v = w; w = tmp;
    }
    hipLaunchKernelGGL(mg_power_ratio_kernel, dim3(1), dim3(1024), 0, st, (const double*)w, (const double*)v, A.su_row0 * 6, A.su_row1 * 6, lam);      // (v: 8 steps, w: 7 steps)
}
void launch_mg_level_rescale(const MgLevelDev& A, const double* lam, double omega, hipStream_t st) {
    const int64_t cnt = (int64_t)(A.su_row1 - A.su_row0) * 36;
    if (cnt > 0) hipLaunchKernelGGL(mg_rescale_dinv_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, A, lam, omega, 1.75, 1.5);
}
void launch_mg_level_inverses(const MgLevelDev& A, double omega, int32_t* fail, hipStream_t st) {
    const int rows = A.su_row1 - A.su_row0;
    const int64_t blocks = A.su_blk1 - A.su_blk0;
    if (rows > 0) hipLaunchKernelGGL(mg_dinv_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, A, omega, fail, 0, (const double*)nullptr);
    if (blocks > 0) hipLaunchKernelGGL(mg_val_f32_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, st, A);
}
void launch_mg_transition_ps(const MgLevelDev& A, double prolong_scale, hipStream_t st) {
    const int cnt = A.su_ps1 - A.su_ps0;
    if (cnt > 0) hipLaunchKernelGGL(mg_ps_kernel<false>, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, A, prolong_scale);
}
void launch_mg_transition_w(const MgLevelDev& A, hipStream_t st) {
    const int cnt = A.su_w1 - A.su_w0;
    if (cnt <= 0) return;
    hipLaunchKernelGGL(mg_w_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, A);
    if (A.rt_valf) hipLaunchKernelGGL(mg_rt_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, A);
}
void launch_mg_transition_product(const MgLevelDev& A, const MgLevelDev& B, hipStream_t st) {
    const int64_t cnt = A.su_prod ? (int64_t)A.n_su_prod : B.nnzb;
    if (cnt > 0) hipLaunchKernelGGL(mg_psTw_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, A, B);
}
void launch_mg_level_galerkin(const MgLevelDev& A, const MgLevelDev& B, hipStream_t st) {
    const int64_t cnt = B.su_blk1 - B.su_blk0;
    if (cnt > 0) hipLaunchKernelGGL(mg_galerkin_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, A, B);
}
// ... step by step, for a distributed level: v = the fixed start vector on the rank's own rows (the formula of the whole-level run: the owners' rows of the same vector), one step
// w = D^-1 A v on its own tiles (the halo of v exchanged by the caller), this rank's part of the two norms and the level's failure flag, lambda from the all-reduced sums
void launch_mg_power_init(const MgLevelDev& A, hipStream_t st) {
    const int n6 = A.n * 6;
    hipLaunchKernelGGL(mg_power_init_kernel, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, st, A.x, A.xt, n6, A.su_row0 * 6, A.su_row1 * 6);
}
void launch_mg_power_step(const MgLevelDev& A, const double* v, double* w, double omega, hipStream_t st) {
    if (A.tiles_own > 0) hipLaunchKernelGGL(mg_smooth_step_kernel, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, (const double*)nullptr, v, (double*)nullptr, (const double*)nullptr, (const double*)nullptr, w, -1.0 / omega, (const int32_t*)nullptr);
}
__global__ __launch_bounds__(1024) void mg_power_sums_kernel(const double* __restrict__ a, const double* __restrict__ b, int lo6, int hi6, const int32_t* __restrict__ fail, double* __restrict__ out3) {
    __shared__ double red[32];
    double sa = 0.0, sb = 0.0;
    for (int i = lo6 + threadIdx.x; i < hi6; i += blockDim.x) { sa += a[i] * a[i]; sb += b[i] * b[i]; }
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sa; red[16 + (threadIdx.x >> 6)] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int k = 0; k < 16; ++k) { ta += red[k]; tb += red[16 + k]; }
        out3[0] = ta; out3[1] = tb; out3[2] = *fail != 0 ? 1.0 : 0.0;
    }
}
__global__ void mg_power_finish_kernel(const double* __restrict__ in3, int32_t* __restrict__ fail, double* __restrict__ lam) { if (in3[2] != 0.0) *fail = 1; *lam = in3[0] > 0.0 ? sqrt(in3[1] / in3[0]) : 0.0; }
void launch_mg_power_sums(const MgLevelDev& A, const double* a, const double* b, const int32_t* fail, double* out3, hipStream_t st) {
    hipLaunchKernelGGL(mg_power_sums_kernel, dim3(1), dim3(1024), 0, st, a, b, A.su_row0 * 6, A.su_row1 * 6, fail, out3);
}
void launch_mg_power_finish(const double* in3, int32_t* fail, double* lam, hipStream_t st) { hipLaunchKernelGGL(mg_power_finish_kernel, dim3(1), dim3(1), 0, st, in3, fail, lam); }

void launch_mg_galerkin0(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgDev& M, const MgLevelDev* levels, hipStream_t st, bool hoff_valid) {
    const int64_t cnt = M.g0_slots ? (int64_t)M.n_g0 : levels[0].nnzb;
    if (cnt <= 0) return;
    if (hoff_valid) hipLaunchKernelGGL((mg_galerkin0_kernel<true>), dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, G, L, Sc, C, M, levels[0]);
    else hipLaunchKernelGGL((mg_galerkin0_kernel<false>), dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, G, L, Sc, C, M, levels[0]);
}
void launch_mg_assemble(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, double omega, int32_t* fail, hipStream_t st, double prolong_scale, bool hoff_valid) {
    launch_mg_galerkin0(G, L, Sc, C, M, levels, st, hoff_valid);
    launch_mg_assemble_rest(M, levels, K, omega, fail, st, prolong_scale);
}
void launch_mg_dense_top(const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, hipStream_t st) {
    const MgLevelDev& T = levels[M.n_levels - 1];
    (void)hipMemsetAsync(K.Ac, 0, (size_t)K.nc * K.nc * sizeof(double), st);
    if (K.nc > 6 * K.n_agg) hipLaunchKernelGGL(coarse_pad_identity_kernel, dim3((unsigned)((K.nc - 6 * K.n_agg + 63) / 64)), dim3(64), 0, st, K);
    hipLaunchKernelGGL(mg_dense_scatter_kernel, dim3((unsigned)(((int64_t)T.n * 36 + 255) / 256)), dim3(256), 0, st, T, K);
}
void launch_mg_assemble_rest(const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, double omega, int32_t* fail, hipStream_t st, double prolong_scale, int first_level) {
    // level by level: the block-Jacobi inverse of level l is needed by a smoothed transition to level l+1 (Ps = (I - c Dinv A) P)
    for (int l = first_level; l < M.n_levels; ++l) {
        if (l > first_level) {
            const MgLevelDev& A = levels[l - 1];
            if (A.smoothed) { launch_mg_transition_ps(A, prolong_scale, st); launch_mg_transition_w(A, st); launch_mg_transition_product(A, levels[l], st); }
            else launch_mg_level_galerkin(A, levels[l], st);
        }
        if (l + 1 < M.n_levels) {
            launch_mg_level_inverses(levels[l], omega, fail, st);
            launch_mg_level_power(levels[l], omega, st);
            launch_mg_level_rescale(levels[l], levels[l].xf, omega, st);
        }
    }
    launch_mg_dense_top(M, levels, K, st);
}
// level 1 of a hierarchy with the smoothed keyframe transition: F = the keyframe level (set-up view), T = its transfer view (Ps's pattern in the explicit operator's fields)
void launch_mg_assemble_fine(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgLevelDev& F, const MgLevelDev& T, const MgLevelDev& L1, double omega, int32_t* fail, hipStream_t st,
                             double prolong_scale, bool hoff_valid, const double* pose8) {
    if (hoff_valid) hipLaunchKernelGGL((mg_fine_blocks_kernel<true>), dim3((unsigned)((F.nnzb + 3) / 4)), dim3(256), 0, st, G, L, Sc, C, F);
    else hipLaunchKernelGGL((mg_fine_blocks_kernel<false>), dim3((unsigned)((F.nnzb + 3) / 4)), dim3(256), 0, st, G, L, Sc, C, F);
    if (F.dlump && pose8) {      // filtered form: Ps_0 = (I - c D_f^-1 A_f) P_0 with the lumped diagonal
        hipLaunchKernelGGL(mg_fine_lump_kernel, dim3((unsigned)((F.n + 3) / 4)), dim3(256), 0, st, F, pose8);
        hipLaunchKernelGGL(mg_dinv_kernel, dim3((unsigned)((F.n + 255) / 256)), dim3(256), 0, st, F, omega, fail, 1, (const double*)F.dlump);
        hipLaunchKernelGGL(mg_ps_kernel<true>, dim3((unsigned)((F.n_ps + 3) / 4)), dim3(256), 0, st, F, prolong_scale);
    } else {
        hipLaunchKernelGGL(mg_dinv_kernel, dim3((unsigned)((F.n + 255) / 256)), dim3(256), 0, st, F, omega, fail, 1, (const double*)nullptr);
        hipLaunchKernelGGL(mg_ps_kernel<false>, dim3((unsigned)((F.n_ps + 3) / 4)), dim3(256), 0, st, F, prolong_scale);
    }
    hipLaunchKernelGGL(mg_w_kernel, dim3((unsigned)((F.n_w + 3) / 4)), dim3(256), 0, st, F);
    hipLaunchKernelGGL(mg_psTw_kernel, dim3((unsigned)((L1.nnzb + 3) / 4)), dim3(256), 0, st, F, L1);
    hipLaunchKernelGGL(mg_ps_f32_kernel, dim3((unsigned)((T.n_ps + 3) / 4)), dim3(256), 0, st, T);
}

// ---- the cycle ----
// r_1 = P_0^T r over the keyframes of each level-1 aggregate; x_1 = Dinv_1 r_1 (skipped when level 1 is the dense level)
__global__ __launch_bounds__(CG_BLOCK) void mg_restrict0_kernel(MgDev M, const double* __restrict__ rv, double* __restrict__ r_out, double* __restrict__ x_out,
                                                                 const double* __restrict__ Dinv, const int32_t* __restrict__ stop, const double* __restrict__ own = nullptr /* several ranks: owner weights */) {
    __shared__ double rb[CG_BLOCK];
    const int stopped = stop ? *stop : 0;        // requested together with the first data loads, tested when they are needed: no round trip of its own
    const int a = M.a0 + blockIdx.x * MG_TILE_ROWS + threadIdx.x / 6, k = threadIdx.x % 6;
    const bool live = a < M.a1;
    int m0 = 0, m1 = 0;
    if (live) { m0 = M.mem0_ptr[a]; m1 = M.mem0_ptr[a + 1]; }
    if (stopped) return;
    double s = 0.0;
    if (live) {
        if (own) for (int m = m0; m < m1; ++m) { const int i = M.mem0[m]; s += own[i] * mg_restrict_comp(rv + (size_t)i * 6, M.d0 + (size_t)i * 3, k); }
        else for (int m = m0; m < m1; ++m) { const int i = M.mem0[m]; s += mg_restrict_comp(rv + (size_t)i * 6, M.d0 + (size_t)i * 3, k); }
        r_out[(size_t)a * 6 + k] = s;
    }
    if (!x_out) return;
    rb[threadIdx.x] = s;
    __syncthreads();
    if (live) {
        const double* Dk = Dinv + (size_t)a * 36 + k * 6;
        const double* ra = rb + (threadIdx.x - k);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x += Dk[j] * ra[j];
        x_out[(size_t)a * 6 + k] = x;
    }
}
#ifdef PGO_MG_TIMELINE
// Development aid (variant build only: scripts/dev/mg_timeline.py): thread 0 of every workgroup of mg_down_kernel records the 100-MHz wall clock at its phase boundaries, after a
// full wait for the memory operations issued so far; [level][workgroup][8] ticks (level = MgLevelDev::pad3_, set at install).
__device__ unsigned long long pgo_mg_tl[4 * 1024 * 8];
#define MG_TL(slot) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0 && A.pad3_ < 4 && blockIdx.x < 1024) pgo_mg_tl[((size_t)A.pad3_ * 1024 + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
extern "C" int pgo_debug_mg_timeline(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pgo_mg_tl), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define MG_TL(slot) do {} while (0)
#endif
// t = r - A x on the rows of a tile of whole aggregates; r_next = P^T t; x_next = Dinv_next r_next (when the next level is a sparse one).
// Every kernel of the cycle starts on cold L2s (kernel boundaries invalidate them), so what it costs is its chain of DEPENDENT loads:
// tile_info -> rowptr -> col -> x is the only chain here; everything else a lane will need (its r entry, its row's offset d, the member
// range of its aggregate, the Dinv row of the next level) is requested up front, before the first barrier.
__global__ __launch_bounds__(CG_BLOCK) void mg_down_kernel(MgLevelDev A, double* __restrict__ r_next, double* __restrict__ x_next, const double* __restrict__ Dinv_next,
                                                            const int32_t* __restrict__ stop) {
    __shared__ double xch[CG_BLOCK * 7];
    __shared__ double tb[CG_BLOCK];
    __shared__ double cb[CG_BLOCK];
    MG_TL(0);
    const int stopped = stop ? *stop : 0;
    const int q6 = threadIdx.x / 6, c = threadIdx.x % 6;
    const int li = q6 & ((MG_TILE_ROWS >> A.seg_shift) - 1), sg = q6 >> (5 - A.seg_shift);      // row of the tile, lane group (0 unless the level's rows are split)
    const int tile_d = (int)blockIdx.x + A.tile0;
    const int4 ti = A.tile_info[tile_d];          // {a0, a1, i0, i1}
    const int2 rb = A.tile_rows[tile_d * MG_TILE_ROWS + li];   // this lane's row: its block range (independent of tile_info)
    MG_TL(1);
    if (stopped) return;
    const int a0 = ti.x, na = ti.y - ti.x, i0 = ti.z, i1 = ti.w;
    const int row = i0 + li;
    const bool rowlive = row < i1;
    const bool live = rowlive && sg == 0;
    const bool lagg = threadIdx.x < na * 6;
    const int a = a0 + li;
    double rv = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
    int m0 = 0, m1 = 0;
    double Dk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (live) { rv = A.r[(size_t)row * 6 + c]; d0 = A.d[(size_t)row * 3]; d1 = A.d[(size_t)row * 3 + 1]; d2 = A.d[(size_t)row * 3 + 2]; }
    if (lagg) {
        m0 = A.agg_ptr[a]; m1 = A.agg_ptr[a + 1];
        if (x_next) {
            const double2* Dp = reinterpret_cast<const double2*>(Dinv_next + (size_t)a * 36 + c * 6);
            const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2];
            Dk[0] = u0.x; Dk[1] = u0.y; Dk[2] = u1.x; Dk[3] = u1.y; Dk[4] = u2.x; Dk[5] = u2.y;
        }
    }
    MG_TL(2);
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    MG_ROW_PRODUCT(blockIdx.x, A.x);
    MG_TL(3);
    double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
    for (int q = 0; q < 6; ++q) mine[q] = acc[q];
    __syncthreads();
    if (live) tb[threadIdx.x] = rv - mg_gather_row(xch, li, c, A.seg_shift);
    __syncthreads();
    MG_TL(4);
    if (live) {                                       // the row's own contribution (P_row^T t)[c]
        const double d[3] = {d0, d1, d2};
        cb[threadIdx.x] = mg_restrict_comp(tb + (size_t)li * 6, d, c);
    }
    __syncthreads();
    double s = 0.0;
    if (lagg) {
        for (int m = m0; m < m1; ++m) s += cb[(m - i0) * 6 + c];
        r_next[(size_t)a * 6 + c] = s;
    }
    MG_TL(5);
    if (!x_next) return;
    tb[threadIdx.x] = s;
    __syncthreads();
    if (lagg) {
        const double* ra = tb + (threadIdx.x - c);
        double x = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) x += Dk[j] * ra[j];
        x_next[(size_t)a * 6 + c] = x;
    }
    MG_TL(6);
}
// dense level: one workgroup of 6 wavefronts per node — wavefront q takes row 6 a + q of the explicit inverse (one wavefront per node leaves
// the 58 MB of a 2688-wide inverse to 440 wavefronts: 21 us; one per row: 2640 wavefronts) — then x + s P y on the node's members below
__global__ __launch_bounds__(384) void mg_dense_solve_kernel(CoarseDev K, MgLevelDev Below, int has_below, double scale, const int32_t* __restrict__ stop) {
    __shared__ double ys[6];
    const int stopped = stop ? *stop : 0;
    const int a = blockIdx.x;
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double2* __restrict__ x = reinterpret_cast<const double2*>(K.rc);
    const float4* __restrict__ Ar = reinterpret_cast<const float4*>(K.Acf + (size_t)(a * 6 + q) * K.nc);     // the inverse in fp32: half the bytes of the dominant stream
    const int n4 = K.nc >> 2;
    double s = 0.0;
    if (lane < n4) { const float4 u = Ar[lane]; const double2 v = x[2 * lane], w = x[2 * lane + 1]; s = (double)u.x * v.x + (double)u.y * v.y + (double)u.z * w.x + (double)u.w * w.y; }   // first trip issued before the flag is needed
    if (stopped) return;
#pragma unroll 4      // (8: no change, measured)
    for (int j = lane + 64; j < n4; j += 64) { const float4 u = Ar[j]; const double2 v = x[2 * j], w = x[2 * j + 1]; s += (double)u.x * v.x + (double)u.y * v.y + (double)u.z * w.x + (double)u.w * w.y; }
    s = wave_sum(s);
    if (lane == 0) { ys[q] = s; K.yc[a * 6 + q] = s; }
    if (!has_below) return;
    __syncthreads();
    double y[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] = ys[k];
    const int c0 = Below.agg_ptr[a], c1 = Below.agg_ptr[a + 1];
    for (int idx = threadIdx.x; idx < (c1 - c0) * 6; idx += 384) {
        const int i = c0 + idx / 6, k = idx % 6;
        Below.xt[(size_t)i * 6 + k] = Below.x[(size_t)i * 6 + k] + scale * mg_prolong_comp(y, Below.d + (size_t)i * 3, k);
    }
}
// x = xt + Dinv (r - A xt) on a tile; then xt = x + s P x on the members (level below) of the tile's rows.  Loads hoisted as in mg_down.
// FINE (level 1 only): the prolongation to the KEYFRAMES is done here too — z_i += s P_i x_1[agg0(i)] over the keyframes of the tile's rows
// (mem0 lists) with this workgroup's share of r.(P x_1) going to its own partial-sum slot after the update kernel's (CgDev::extra_rz).
// EXPLICIT (xnext != null; levels with MgLevelDev::rt_valf): x = v + s R^T x_next with v = x_pre + Dinv (r - A x_pre) left in A.y by mg_sdown_kernel — the smoothed
// prolongation and the post-smoothing step in ONE row product over the fp32 blocks of R^T (W's pattern), no Dinv, no second pass over the level's own matrix.
template <bool FINE>
__global__ __launch_bounds__(CG_BLOCK) void mg_up_kernel(MgLevelDev A, MgLevelDev Below, int has_below, double scale, const int32_t* __restrict__ stop,
                                                          MgDev M, const double* __restrict__ rfine, double* __restrict__ zfine, double* __restrict__ part_extra,
                                                          const double* __restrict__ xnext = nullptr) {
    __shared__ double xch[CG_BLOCK * 7];
    __shared__ double tb[CG_BLOCK];
    __shared__ double xb[CG_BLOCK];
    const int stopped = stop ? *stop : 0;
    const int q6 = threadIdx.x / 6, c = threadIdx.x % 6;
    const int li = q6 & ((MG_TILE_ROWS >> A.seg_shift) - 1), sg = q6 >> (5 - A.seg_shift);
    double acc2 = 0.0;      // FINE: this workgroup's share of r.(P x_1), over all its tiles
    // FINE: the grid is capped at MAX_PARTIALS workgroups (one r.z partial slot each), a workgroup takes every gridDim-th tile; otherwise one tile per workgroup
    for (int tile = A.tile0 + (int)blockIdx.x; tile < A.tile0 + A.tiles_own; tile += gridDim.x) {
    const int4 ti = A.tile_info[tile];
    const bool expl = xnext != nullptr;
    const int2 rb = expl ? A.rt_rows[tile * MG_TILE_ROWS + li] : A.tile_rows[tile * MG_TILE_ROWS + li];
    if (stopped) return;
    const int i0 = ti.z, i1 = ti.w;
    const int row = i0 + li;
    const bool rowlive = row < i1;
    const bool live = rowlive && sg == 0;
    double rv = 0.0, xv = 0.0;
    double Dk[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int c0 = 0, c1 = 0;
    if (live) {
        rv = A.r[(size_t)row * 6 + c]; xv = expl ? A.y[(size_t)row * 6 + c] : A.xt[(size_t)row * 6 + c];
        if (!expl) {
            const double2* Dp = reinterpret_cast<const double2*>(A.Dinv + (size_t)row * 36 + c * 6);
            const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2];
            Dk[0] = u0.x; Dk[1] = u0.y; Dk[2] = u1.x; Dk[3] = u1.y; Dk[4] = u2.x; Dk[5] = u2.y;
        }
    }
    if (has_below) { c0 = Below.agg_ptr[i0]; c1 = Below.agg_ptr[i1]; }
    // the first trip of the prolongation loop: child row, its parent, its pre-smoothed x and offset d
    const int idx0 = threadIdx.x;
    int ch = 0, chk = 0, chp = 0; double chx = 0.0, chd[3] = {0.0, 0.0, 0.0};
    const bool first = has_below && idx0 < (c1 - c0) * 6;
    if (first) {
        ch = c0 + idx0 / 6; chk = idx0 % 6;
        chp = Below.parent[ch]; chx = Below.x[(size_t)ch * 6 + chk];
        chd[0] = Below.d[(size_t)ch * 3]; chd[1] = Below.d[(size_t)ch * 3 + 1]; chd[2] = Below.d[(size_t)ch * 3 + 2];
    }
    // FINE: everything the fine prolongation will need is requested now — keyframe, its aggregate's tile-local row, offset d, r and z of up to
    // four (keyframe, row pair) items per lane (<= 32 rows x 8 keyframes x 3 pairs per tile) — so that this chain runs beside the smoothing's
    constexpr int FT = FINE ? (MG_TILE_ROWS * 8 * 3 + CG_BLOCK - 1) / CG_BLOCK : 1;
    int fn[FT], fa[FT]; double fd[FT][3]; double2 fz[FT];
    int fcnt = 0;
    if (FINE) {
        const int e0 = M.mem0_ptr[i0];
        fcnt = (M.mem0_ptr[i1] - e0) * 3;
#pragma unroll
        for (int tt = 0; tt < FT; ++tt) {
            const int idx = threadIdx.x + tt * CG_BLOCK;
            fn[tt] = -1;
            if (idx < fcnt) {
                const int n = M.mem0[e0 + idx / 3], j = idx % 3;
                fn[tt] = n * 3 + j; fa[tt] = M.agg0[n] - i0;
                fd[tt][0] = M.d0[(size_t)n * 3]; fd[tt][1] = M.d0[(size_t)n * 3 + 1]; fd[tt][2] = M.d0[(size_t)n * 3 + 2];
                fz[tt] = reinterpret_cast<const double2*>(zfine)[(size_t)n * 3 + j];
            }
        }
    }
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (expl) { if (rowlive) { int kb, ke; mg_split_row(rb, sg, A.seg_shift, kb, ke); mg_row_accumulate_f32(kb, ke, A.w_col, A.rt_valf, xnext, c, acc); } }
    else MG_ROW_PRODUCT(tile, A.xt);
    double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
    for (int q = 0; q < 6; ++q) mine[q] = acc[q];
    __syncthreads();
    if (expl) {
        if (live) {
            const double x = xv + scale * mg_gather_row(xch, li, c, A.seg_shift);
            A.xf[(size_t)row * 6 + c] = x;
            xb[threadIdx.x] = x;
        }
    } else {
        if (live) tb[threadIdx.x] = rv - mg_gather_row(xch, li, c, A.seg_shift);
        __syncthreads();
        if (live) {
            const double* ta = tb + (threadIdx.x - c);
            double x = xv;
#pragma unroll
            for (int j = 0; j < 6; ++j) x += Dk[j] * ta[j];
            A.xf[(size_t)row * 6 + c] = x;
            xb[threadIdx.x] = x;
        }
    }
    if (FINE) {
        __syncthreads();
        // r.(P_0 x_1) = (P_0^T r).x_1 = r_1.x_1: the coarse part of r.z from this level's own vectors (the keyframes' r is not read again)
        if (live) acc2 += scale * rv * xb[threadIdx.x];
#pragma unroll
        for (int tt = 0; tt < FT; ++tt) {
            if (fn[tt] < 0) continue;
            const int j = fn[tt] % 3;
            const double* y = xb + (size_t)fa[tt] * 6;
            const double b0 = scale * mg_prolong_comp(y, fd[tt], 2 * j), b1 = scale * mg_prolong_comp(y, fd[tt], 2 * j + 1);
            reinterpret_cast<double2*>(zfine)[fn[tt]] = make_double2(fz[tt].x + b0, fz[tt].y + b1);
        }
        __syncthreads();      // the LDS buffers are reused by the next tile
        continue;
    }
    if (!has_below) return;
    __syncthreads();
    if (first) Below.xt[(size_t)ch * 6 + chk] = chx + scale * mg_prolong_comp(xb + (size_t)(chp - i0) * 6, chd, chk);
    for (int idx = threadIdx.x + CG_BLOCK; idx < (c1 - c0) * 6; idx += CG_BLOCK) {
        const int i = c0 + idx / 6, k = idx % 6;
        const double* y = xb + (size_t)(Below.parent[i] - i0) * 6;
        Below.xt[(size_t)i * 6 + k] = Below.x[(size_t)i * 6 + k] + scale * mg_prolong_comp(y, Below.d + (size_t)i * 3, k);
    }
    }
    if (FINE) {
        __shared__ double red[CG_BLOCK / 64];
        const double s2 = block_sum(acc2, red);
        if (threadIdx.x == 0) part_extra[blockIdx.x] = s2;
    }
}
// z_i += P_i y_{agg0(i)} and r.z += r.(P y), in cg_update's lane / workgroup mapping (same partial-sum slots)
__global__ __launch_bounds__(CG_BLOCK) void mg_prolong0_kernel(GraphDev G, MgDev M, const double* __restrict__ y1, const double* __restrict__ rv, double* __restrict__ zv,
                                                                double* __restrict__ part_rz, double scale, const int32_t* __restrict__ stop) {
    __shared__ double red[CG_BLOCK / 64];
    if (stop && *stop) return;
    const int64_t pairs = G.N * 3;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * CG_BLOCK) {
        const int64_t n = i / 3;
        const int j = (int)(i - n * 3);
        const int a = M.agg0[n];
        if (a < 0) continue;
        const double* y = y1 + (size_t)a * 6;
        const double* d = M.d0 + (size_t)n * 3;
        const double a0 = scale * mg_prolong_comp(y, d, 2 * j), a1 = scale * mg_prolong_comp(y, d, 2 * j + 1);
        double2* zp = reinterpret_cast<double2*>(zv) + i;
        const double2 r = reinterpret_cast<const double2*>(rv)[i];
        double2 z = *zp;
        z.x += a0; z.y += a1;
        *zp = z;
        const double w = G.own ? G.own[n] : 1.0;      // several ranks: a shared keyframe counts once, at its owner
        acc += w * (r.x * a0 + r.y * a1);
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part_rz[blockIdx.x] += s;
}

// cg_update_kernel + the multigrid's restriction to level 1.  A workgroup trip covers one run of MG_BLOCK0 keyframes; the level-1 aggregates never cross
// a run boundary, so r_1 = P_0^T r' of the run's aggregates is formed from the new residual while it is still in LDS (members in list order, the
// same sums as mg_restrict0_kernel) — and with it x_1 = w D_1^-1 r_1.  The slot table entries a lane needs are requested before the partial-sum
// re-reduction, together with the vector operands.
// SR: the single-reduction form of the update (cg_update_kernel<true>: p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s in place) with the same restriction.
#ifdef PGO_SR_MG_NO_BOUNDS      // A/B aid (variant build): 174 VGPRs, two waves per SIMD, no spills
#define PGO_SR_MG_WAVES 1
#else
#define PGO_SR_MG_WAVES 3
#endif
template <bool SR>
__global__ __launch_bounds__(CG_BLOCK, SR ? PGO_SR_MG_WAVES : 1) void cg_update_mg_kernel(GraphDev G, CgDev C, MgDev M, double* __restrict__ r1_out, double* __restrict__ x1_out, const double* __restrict__ Dinv1,
                                                                 int parity, int nparts_pq, int nparts, int first) {
    static_assert(CG_BLOCK / 3 == MG_BLOCK0, "one workgroup trip of the vector update = one run of the slot table");
    __shared__ double red[2 * (CG_BLOCK / 64) + 1];
    const double2* __restrict__ rin = reinterpret_cast<const double2*>(SR ? C.r : (parity ? C.r2 : C.r));
    double2* __restrict__ rout = reinterpret_cast<double2*>(SR ? C.r : (parity ? C.r : C.r2));
    const double2* __restrict__ pcur = reinterpret_cast<const double2*>(SR ? C.p : (parity ? C.p2 : C.p));
    const double2* __restrict__ qv = reinterpret_cast<const double2*>(C.q);
    double2* __restrict__ xv = reinterpret_cast<double2*>(C.x);
    double2* __restrict__ zv = reinterpret_cast<double2*>(C.z);
    double2* __restrict__ pout = reinterpret_cast<double2*>(C.p);      // (SR only)
    double2* __restrict__ sv = reinterpret_cast<double2*>(C.p2);       // (SR only) s = A p
    double2 u0 = make_double2(0.0, 0.0), s0 = u0;
    constexpr int KF = CG_BLOCK / 3;
    const int t = threadIdx.x;
    const int64_t pairs = G.N * 3;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    const int64_t trips = (pairs + stride - 1) / stride;
    // Everything a trip reads from global memory is requested together — residual / direction / solution / offsets, the block-Jacobi factors (two 16-B loads per lane, staged
    // through registers), the run's slot table and, once that has arrived, the Dinv rows of its aggregates — for the first trip before the partial-sum re-reduction:
    // one round trip (+ the table -> Dinv hop) and three barriers per trip instead of three round trips and seven barriers.
    const int c6 = t % 6;
    double2 r0, q0, p0, x0; double d0, d1, d2; float4 lf0, lf1; int4 tab0, tab1; double Dk0[6], Dk1[6];
    auto load_trip = [&](int64_t run) {
        const int64_t base = run * CG_BLOCK, i = base + t;
        r0 = make_double2(0.0, 0.0); q0 = r0; p0 = r0; x0 = r0; d0 = d1 = d2 = 0.0;
        if (SR) { u0 = r0; s0 = r0; }
        lf0 = make_float4(0.f, 0.f, 0.f, 0.f); lf1 = lf0; tab0 = make_int4(-1, -1, -1, 0); tab1 = tab0;
        if (base < pairs) {     // the run exists: its slots are served by all lanes, also those beyond the last keyframe
            tab0 = M.blk_tab[(size_t)run * MG_BLOCK0 + t / 6];
            tab1 = M.blk_tab[(size_t)run * MG_BLOCK0 + (t + CG_BLOCK) / 6];
        }
        if (i < pairs) {
            r0 = rin[i]; q0 = qv[i]; p0 = pcur[i]; x0 = xv[i];
            if (SR) { u0 = zv[i]; s0 = sv[i]; }
            const double* d = M.d0 + (size_t)(i / 3) * 3; d0 = d[0]; d1 = d[1]; d2 = d[2];
        }
        const int64_t first_node = base / 3;
        const float4* lp = reinterpret_cast<const float4*>(C.Lf + (size_t)first_node * LF_STRIDE);
        if (first_node + t / 6 < G.N) lf0 = lp[t];
        if (first_node + (t + CG_BLOCK) / 6 < G.N) lf1 = lp[t + CG_BLOCK];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) { Dk0[jj] = 0.0; Dk1[jj] = 0.0; }
        if (x1_out) {
            if (tab0.x >= 0) { const double2* Dp = reinterpret_cast<const double2*>(Dinv1 + (size_t)tab0.x * 36 + c6 * 6); const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2]; Dk0[0] = u0.x; Dk0[1] = u0.y; Dk0[2] = u1.x; Dk0[3] = u1.y; Dk0[4] = u2.x; Dk0[5] = u2.y; }
            if (tab1.x >= 0) { const double2* Dp = reinterpret_cast<const double2*>(Dinv1 + (size_t)tab1.x * 36 + c6 * 6); const double2 u0 = Dp[0], u1 = Dp[1], u2 = Dp[2]; Dk1[0] = u0.x; Dk1[1] = u0.y; Dk1[2] = u1.x; Dk1[3] = u1.y; Dk1[4] = u2.x; Dk1[5] = u2.y; }
        }
    };
    load_trip((int64_t)blockIdx.x);
    double alpha = 0.0, beta = 0.0;
    if (SR) { if (!sr_head(C, parity, first, nparts_pq, nparts, red, alpha, beta)) return; }
    else {
        double pq, rz;
        if (block_total2_done(C.flags, C.part_pq, nparts_pq, C.part_rz + parity * RZ_STRIDE, nparts + C.extra_rz, red, pq, rz)) return;
        if (!(pq > 0.0)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) C.flags[1] = 1;
            if (threadIdx.x == 0) C.part_rz[(parity ^ 1) * RZ_STRIDE + blockIdx.x] = 0.0;
            return;
        }
        alpha = rz / pq;
    }
    __shared__ double2 rnew[CG_BLOCK];
    __shared__ double2 btr[CG_BLOCK];
    __shared__ double rs[2 * CG_BLOCK];
    __shared__ __attribute__((aligned(16))) float lfs[KF * LF_STRIDE];
    static_assert(KF * LF_STRIDE / 4 == 2 * CG_BLOCK, "two 16-B loads per lane stage a trip's factors");
    double acc = 0.0;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t run = it * gridDim.x + blockIdx.x;
        const int64_t base = run * CG_BLOCK;
        const int64_t i = base + t;
        const bool live = i < pairs;
        double2 rr = make_double2(0.0, 0.0);
        if (live) {
            if (SR) {
                p0.x = u0.x + beta * p0.x; p0.y = u0.y + beta * p0.y;
                s0.x = q0.x + beta * s0.x; s0.y = q0.y + beta * s0.y;
                rr = make_double2(r0.x - alpha * s0.x, r0.y - alpha * s0.y);
                x0.x += alpha * p0.x; x0.y += alpha * p0.y;
                pout[i] = p0; sv[i] = s0;
            } else {
                rr = make_double2(r0.x - alpha * q0.x, r0.y - alpha * q0.y);
                x0.x += alpha * p0.x; x0.y += alpha * p0.y;
            }
            rout[i] = rr; xv[i] = x0;
        }
        reinterpret_cast<float4*>(lfs)[t] = lf0; reinterpret_cast<float4*>(lfs)[t + CG_BLOCK] = lf1;
        rnew[t] = rr;
        const double e0 = d0, e1 = d1, e2 = d2;
        const int4 tb0 = tab0, tb1 = tab1;
        __syncthreads();
        {
            const int j = t % 3;
            const double* r6 = reinterpret_cast<const double*>(rnew + (t - j));
            double2 b;
            if (j == 0) b = make_double2(r6[0] + 2.0 * (e1 * r6[5] - e2 * r6[4]), r6[1] + 2.0 * (e2 * r6[3] - e0 * r6[5]));
            else if (j == 1) b = make_double2(r6[2] + 2.0 * (e0 * r6[4] - e1 * r6[3]), r6[3]);
            else b = make_double2(r6[4], r6[5]);
            btr[t] = b;
            if (live) {
                const double2 z = lf_apply_pair(lfs + (t / 3) * LF_STRIDE, r6, j);
                zv[i] = z;
                acc += rr.x * z.x + rr.y * z.y;
            }
        }
        __syncthreads();
        // r_1 of the run's aggregates: slot u / 6, component u % 6 (two rounds of the workgroup cover the MG_BLOCK0 slots)
        double sum2[2];
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const int4 tb = round ? tb1 : tb0;
            double sum = 0.0;
            if (tb.x >= 0) {
                const double* col = reinterpret_cast<const double*>(btr) + c6;
                const uint32_t lo = (uint32_t)tb.y, hi = (uint32_t)tb.z;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const uint32_t kf = ((m < 4 ? lo >> (8 * m) : hi >> (8 * (m - 4))) & 0xffu);
                    if (kf != 0xffu) sum += col[kf * 6];
                }
                r1_out[(size_t)tb.x * 6 + c6] = sum;
            }
            sum2[round] = sum;
            rs[round * CG_BLOCK + t] = sum;
        }
        if (x1_out) {
            __syncthreads();
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const int4 tb = round ? tb1 : tb0;
                if (tb.x >= 0) {
                    const double* ra = rs + round * CG_BLOCK + (t - c6);
                    double x = 0.0;
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) x += (round ? Dk1[jj] : Dk0[jj]) * ra[jj];
                    x1_out[(size_t)tb.x * 6 + c6] = x;
                }
            }
        }
        (void)sum2;
        if (it + 1 < trips) { load_trip(run + gridDim.x); __syncthreads(); }      // (a prefetch during this trip's arithmetic costs 70 VGPRs: 196, two waves per SIMD); the LDS buffers are rewritten by the next trip
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) C.part_rz[(parity ^ 1) * RZ_STRIDE + blockIdx.x] = s;
}
void launch_cg_update_mg(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, int k, int n_pq_partials, hipStream_t st) {
    const int g = cg_grid(G);
    if (M.n_levels == 1) hipLaunchKernelGGL(cg_update_mg_kernel<false>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, M, K.rc, (double*)nullptr, (const double*)nullptr, k & 1, n_pq_partials, g, 0);
    else hipLaunchKernelGGL(cg_update_mg_kernel<false>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, M, levels[0].r, levels[0].x, (const double*)levels[0].Dinv, k & 1, n_pq_partials, g, 0);
}
void launch_cg_update_mg_sr(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, int k, int first, int n_pq_partials, hipStream_t st) {
    const int g = cg_grid(G);
    if (M.n_levels == 1) hipLaunchKernelGGL(cg_update_mg_kernel<true>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, M, K.rc, (double*)nullptr, (const double*)nullptr, k & 1, n_pq_partials, g, first);
    else hipLaunchKernelGGL(cg_update_mg_kernel<true>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, M, levels[0].r, levels[0].x, (const double*)levels[0].Dinv, k & 1, n_pq_partials, g, first);
}
void launch_mg_apply(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, const double* r, double* z, double* part_rz, double scale, bool inside_iteration, hipStream_t st,
                     bool restricted, double prolong_scale, const MgLevelDev* fine /* transfer view of the keyframe level: smoothed keyframe transition (never `restricted`, never fused) */,
                     const MgExchangeHook* hook, int* hook_rc) {
    const int32_t* stop = inside_iteration ? C.flags : nullptr;     // at PCG start the flag still belongs to the previous solve
    const int nl = M.n_levels;
    const unsigned g1 = (unsigned)((std::max(M.a1 - M.a0, 0) + MG_TILE_ROWS - 1) / MG_TILE_ROWS);      // (several ranks: the rank's own aggregates)
    // several ranks: the exchange a kernel's reads need is issued right before it (pgo_solver.hip); a failing exchange ends the cycle (the caller sees *hook_rc)
    bool hook_failed = false;
    auto xchg = [&](int point, int level) { if (hook && !hook_failed) { const int rc = hook->fn(hook->ctx, point, level); if (rc != 0) { hook_failed = true; if (hook_rc) *hook_rc = rc; } } };
    if (restricted) {}
    else if (fine) {      // r_1 = Ps_0^T r (and x_1 = Dinv_1 r_1): the restriction half of mg_sdown_kernel on the keyframe level's transfer view, which has no tiles of its own
        MgLevelDev T = *fine; T.tiles = 0; T.tiles_own = 0; T.r = const_cast<double*>(r);
        if (nl == 1) hipLaunchKernelGGL(mg_sdown_kernel, dim3((unsigned)T.rT_tiles), dim3(CG_BLOCK), 0, st, T, K.rc, (double*)nullptr, (const double*)nullptr, stop);
        else hipLaunchKernelGGL(mg_sdown_kernel, dim3((unsigned)T.rT_tiles), dim3(CG_BLOCK), 0, st, T, levels[0].r, levels[0].x, (const double*)levels[0].Dinv, stop);
    }
    else if (g1 == 0) {}
    else if (nl == 1) hipLaunchKernelGGL(mg_restrict0_kernel, dim3(g1), dim3(CG_BLOCK), 0, st, M, r, K.rc, (double*)nullptr, (const double*)nullptr, stop);
    else hipLaunchKernelGGL(mg_restrict0_kernel, dim3(g1), dim3(CG_BLOCK), 0, st, M, r, levels[0].r, levels[0].x, (const double*)levels[0].Dinv, stop);
    for (int l = 1; l < nl; ++l) {                     // sparse level l -> level l+1
        MgLevelDev A = levels[l - 1];
        xchg(0, l);
        if (hook_failed) return;
        if (A.smoothed && A.rt_valf) {      // explicit transfer operator: smoothing step and restriction in one launch (two kinds of workgroups)
            const unsigned g = (unsigned)(A.tiles_own + A.rT_tiles);
            if (g == 0) continue;
            if (l + 1 == nl) hipLaunchKernelGGL(mg_sdown_kernel, dim3(g), dim3(CG_BLOCK), 0, st, A, K.rc, (double*)nullptr, (const double*)nullptr, stop);
            else hipLaunchKernelGGL(mg_sdown_kernel, dim3(g), dim3(CG_BLOCK), 0, st, A, levels[l].r, levels[l].x, (const double*)levels[l].Dinv, stop);
            continue;
        }
        if (A.tiles_own == 0) continue;
        if (A.smoothed) {
            // smoothed prolongator: t = r - A x_pre, u = c Dinv t; the restriction kernel then forms P^T (t - A u) = Ps^T t
            hipLaunchKernelGGL(mg_smooth_step_kernel, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, (const double*)A.r, (const double*)A.x, A.t, (const double*)nullptr, (const double*)nullptr, A.u, prolong_scale, stop);
            A.r = A.t; A.x = A.u;
        }
        if (l + 1 == nl) hipLaunchKernelGGL(mg_down_kernel, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, K.rc, (double*)nullptr, (const double*)nullptr, stop);
        else hipLaunchKernelGGL(mg_down_kernel, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, levels[l].r, levels[l].x, (const double*)levels[l].Dinv, stop);
    }
    // the level below a kernel that prolongs: with a smoothed transition it must receive the bare correction e = P x (xt = 0 + P x), smoothed afterwards
    auto below_of = [&](int idx) { MgLevelDev B = levels[idx]; if (B.smoothed) B.x = const_cast<double*>(B.zero); return B; };
    auto expl_at = [&](int idx) { return idx >= 0 && levels[idx].smoothed && levels[idx].rt_valf != nullptr; };      // that level's up-sweep reads x_next itself: nothing is prolonged into it
    xchg(0, nl);
    if (hook_failed) return;
    hipLaunchKernelGGL(mg_dense_solve_kernel, dim3((unsigned)K.n_agg), dim3(384), 0, st, K, nl >= 2 ? below_of(nl - 2) : levels[0], nl >= 2 && !expl_at(nl - 2) ? 1 : 0, scale, stop);
    const bool fused = C.extra_rz > 0;     // level 1's kernel prolongs to the keyframes itself (the solver sets extra_rz = its tile count when that fits the partial-sum slots)
    const unsigned g0 = (unsigned)cg_grid(G);
    for (int l = nl - 1; l >= 1; --l) {
        MgLevelDev A = levels[l - 1];
        xchg(1, l);
        if (hook_failed) return;
        if (A.tiles_own == 0) continue;
        if (expl_at(l - 1)) {      // x = v + s R^T x_next: prolongation and post-smoothing in one launch
            const double* xn = l + 1 == nl ? (const double*)K.yc : (const double*)levels[l].xf;
            if (l == 1 && fused) hipLaunchKernelGGL(mg_up_kernel<true>, dim3((unsigned)(A.tiles_own < MAX_PARTIALS ? A.tiles_own : MAX_PARTIALS)), dim3(CG_BLOCK), 0, st, A, A, 0, scale, stop, M, r, z, part_rz + g0, xn);
            else hipLaunchKernelGGL(mg_up_kernel<false>, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, l >= 2 ? below_of(l - 2) : levels[0], l >= 2 && !expl_at(l - 2) ? 1 : 0, scale, stop, M, r, z, part_rz, xn);
            continue;
        }
        if (A.smoothed) {
            // y = x_pre + e - c Dinv (A e) = x_pre + Ps x_next ; the post-smoothing kernel then works on y
            hipLaunchKernelGGL(mg_smooth_step_kernel, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, (const double*)nullptr, (const double*)A.xt, (double*)nullptr, (const double*)A.x, (const double*)A.xt, A.y, prolong_scale, stop);
            A.xt = A.y;
        }
        if (l == 1 && fused) hipLaunchKernelGGL(mg_up_kernel<true>, dim3((unsigned)(A.tiles_own < MAX_PARTIALS ? A.tiles_own : MAX_PARTIALS)), dim3(CG_BLOCK), 0, st, A, A, 0, scale, stop, M, r, z, part_rz + g0, (const double*)nullptr);
        else hipLaunchKernelGGL(mg_up_kernel<false>, dim3((unsigned)A.tiles_own), dim3(CG_BLOCK), 0, st, A, l >= 2 ? below_of(l - 2) : levels[0], l >= 2 && !expl_at(l - 2) ? 1 : 0, scale, stop, M, r, z, part_rz, (const double*)nullptr);
    }
    xchg(2, 1);
    if (hook_failed) return;
    if (fine) hipLaunchKernelGGL(mg_prolong0s_kernel, dim3(g0), dim3(CG_BLOCK), 0, st, G, *fine, (const double*)(nl == 1 ? K.yc : levels[0].xf), r, z, part_rz, scale, stop);
    else if (!fused) hipLaunchKernelGGL(mg_prolong0_kernel, dim3(g0), dim3(CG_BLOCK), 0, st, G, M, (const double*)(nl == 1 ? K.yc : levels[0].xf), r, z, part_rz, scale, stop);
}
