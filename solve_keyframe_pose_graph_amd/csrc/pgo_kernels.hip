// pgo_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, wave64).  fp64 VALU, HBM-bound; no MFMA
// (6x6 blocks are not a dense contraction).  Layouts: pgo_internal.hpp.  Kernel inventory:
//   K1  k1_edges_kernel<J>     per-edge residual (+ two 6x6 Jacobian blocks), every edge of every class in ONE launch,
//                              one lane per edge, one wavefront per 64-edge tile, endpoint poses staged through LDS
//   K2  k2_node_kernel         per-keyframe segmented reduction  Hd = sum J^T J, g = sum J^T r   (deterministic, no atomics)
//       k2_edge_kernel         per-edge off-diagonal block J1^T J2 and switch couplings
//   K3  cg_spmv_kernel         block-CSR (6x6) SpMV of the Schur-reduced damped normal matrix
//   K4  cg_update/direction    fused PCG vector updates, block-Jacobi apply and dot products
//   K5  plus_kernel            manifold Plus + step norms; k1 (J = false) evaluates the candidate cost
#include <algorithm>

#include "pgo_internal.hpp"

namespace pgo {

// A wavefront's index in the grid / in its workgroup as a value the compiler KNOWS to be wave-uniform (an SGPR): everything indexed with it — list bounds, list entries, edge
// endpoints, offsets — is then read by scalar loads instead of 64 identical vector loads.  The one-wavefront-per-block set-up kernels (Galerkin products, Ps / W / Ps^T W, the two-level
// assembly) walk per-block lists whose every entry costs several such reads; as vector loads they queue in the CU's one texture-address unit (round 4: mg_galerkin0 637 us -> see DESIGN.md).
#ifndef PGO_NO_UNIFORM
__device__ __forceinline__ int wave_in_grid() { return __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)); }
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#else      // A/B aid (variant build): the plain expressions, which the compiler must treat as per-lane values
__device__ __forceinline__ int wave_in_grid() { return (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); }
__device__ __forceinline__ int wave_in_block() { return (int)(threadIdx.x >> 6); }
#endif

// ------------------------------------------------------------------------------------------------
// reductions (fixed-shape trees: results are bitwise reproducible run to run)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// This thread's share of a partial-sum array, p[threadIdx.x + k blockDim.x]: up to 12 entries per thread (2 048 partials on 192 lanes) REQUESTED TOGETHER — the loop form
// `for (i = tid; i < n; i += blockDim) v += p[i]` has a run-time trip count, is not unrolled, and waits for every load before it issues the next: four to eleven dependent
// round trips at the head of every PCG kernel, where the whole workgroup waits for alpha / beta (timeline build: 5 us of the matvec's 26).  Same summation order as the loop.
#ifndef PGO_SERIAL_HEAD
__device__ __forceinline__ double strided_share(const double* __restrict__ p, int n) {
    constexpr int MAXK = 12;
    double v[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { const int i = (int)threadIdx.x + k * (int)blockDim.x; v[k] = i < n ? p[i] : 0.0; }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) s += v[k];
    for (int i = (int)threadIdx.x + MAXK * (int)blockDim.x; i < n; i += blockDim.x) s += p[i];
    return s;
}
#else      // A/B aid (variant build): the loop form
__device__ __forceinline__ double strided_share(const double* __restrict__ p, int n) { double s = 0.0; for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i]; return s; }
#endif
// sum over the workgroup; valid in thread 0.  `buf` holds blockDim/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* buf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) buf[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) s += buf[i];
    return s;
}
// every thread gets sum(partials[0..n)); n <= a few thousand, read through L2
__device__ __forceinline__ double block_total(const double* __restrict__ partials, int n, double* buf) {
    double v = strided_share(partials, n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) buf[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += buf[i];
    return s;
}

// 6x6 blocks of the BSR matrix and of the preconditioner are stored PAIR-MAJOR: element (r, c) lives at (c/2)*12 + r*2 + (c&1),
// so the 6 lanes (rows) of one keyframe read consecutive 16-B words: every SpMV load is a contiguous 96-B run per keyframe.
__device__ __forceinline__ int pm(int r, int c) { return (c >> 1) * 12 + r * 2 + (c & 1); }

// two totals in one pass (one barrier pair instead of two): every PCG kernel starts by re-reducing two partial arrays
__device__ __forceinline__ void block_total2(const double* __restrict__ pa, int na, const double* __restrict__ pb, int nb, double* buf /*2 x nwaves*/, double& sa, double& sb) {
    double va = strided_share(pa, na), vb = strided_share(pb, nb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    va = wave_sum(va); vb = wave_sum(vb);
    __syncthreads();
    if (lane == 0) { buf[wave] = va; buf[nw + wave] = vb; }
    __syncthreads();
    sa = 0.0; sb = 0.0;
    for (int i = 0; i < nw; ++i) { sa += buf[i]; sb += buf[nw + i]; }
}

// three totals in one pass: the two-level method's matvec keeps the block-Jacobi part and the coarse part of r.z apart (below)
__device__ __forceinline__ void block_total3(const double* __restrict__ pa, int na, const double* __restrict__ pb, int nb, const double* __restrict__ pc, int nc, double* buf /*3 x nwaves*/,
                                             double& sa, double& sb, double& sc) {
    double va = strided_share(pa, na), vb = strided_share(pb, nb), vc = strided_share(pc, nc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    va = wave_sum(va); vb = wave_sum(vb); vc = wave_sum(vc);
    __syncthreads();
    if (lane == 0) { buf[wave] = va; buf[nw + wave] = vb; buf[2 * nw + wave] = vc; }
    __syncthreads();
    sa = 0.0; sb = 0.0; sc = 0.0;
    for (int i = 0; i < nw; ++i) { sa += buf[i]; sb += buf[nw + i]; sc += buf[2 * nw + i]; }
}

// The head of every PCG kernel: "has the PCG stopped?" and the re-reduction of two (three) partial-sum arrays.  Measured with the timeline build (scripts/dev/mf_timeline.py, C3):
// as two steps — flag -> barrier -> partial sums -> barrier — the head took 5 us of the matvec's 26, two dependent round trips to cold L2s.  Here ONE lane requests the flag
// BEFORE the partial sums are requested and hands it over through the reduction's own LDS buffer: one round trip, one barrier pair.  (The flag may be raised by workgroup 0 of
// a matvec while other workgroups of the same launch are starting: one lane reads, LDS broadcasts, nobody diverges around a barrier.)  buf: 2 (3) x nwaves + 1 doubles.
__device__ __forceinline__ bool block_total2_done(const int32_t* __restrict__ flags, const double* __restrict__ pa, int na, const double* __restrict__ pb, int nb, double* buf,
                                                  double& sa, double& sb) {
    int f = 0;
    if (threadIdx.x == 0) f = flags[0];
    double va = strided_share(pa, na), vb = strided_share(pb, nb);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    va = wave_sum(va); vb = wave_sum(vb);
    __syncthreads();
    if (lane == 0) { buf[wave] = va; buf[nw + wave] = vb; }
    if (threadIdx.x == 0) buf[2 * nw] = (double)f;
    __syncthreads();
    sa = 0.0; sb = 0.0;
    for (int i = 0; i < nw; ++i) { sa += buf[i]; sb += buf[nw + i]; }
    return buf[2 * nw] != 0.0;
}
__device__ __forceinline__ bool block_total3_done(const int32_t* __restrict__ flags, const double* __restrict__ pa, int na, const double* __restrict__ pb, int nb, const double* __restrict__ pc, int nc,
                                                  double* buf, double& sa, double& sb, double& sc) {
    int f = 0;
    if (threadIdx.x == 0) f = flags[0];
    double va = strided_share(pa, na), vb = strided_share(pb, nb), vc = strided_share(pc, nc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    va = wave_sum(va); vb = wave_sum(vb); vc = wave_sum(vc);
    __syncthreads();
    if (lane == 0) { buf[wave] = va; buf[nw + wave] = vb; buf[2 * nw + wave] = vc; }
    if (threadIdx.x == 0) buf[3 * nw] = (double)f;
    __syncthreads();
    sa = 0.0; sb = 0.0; sc = 0.0;
    for (int i = 0; i < nw; ++i) { sa += buf[i]; sb += buf[nw + i]; sc += buf[2 * nw + i]; }
    return buf[3 * nw] != 0.0;
}

__device__ __forceinline__ size_t tile_elem(int doubles_per_edge, int64_t e, int k) {
    return (size_t)(e >> 6) * (size_t)(doubles_per_edge * TILE) + (size_t)(k >> 1) * (2 * TILE) + (size_t)(e & 63) * 2 + (k & 1);
}

// ------------------------------------------------------------------------------------------------
// K1 — residual + Jacobian blocks for all edges, one launch
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Pose load_pose_global(const double* __restrict__ pose8, int32_t c) {
    const double2* g = reinterpret_cast<const double2*>(pose8 + (size_t)c * 8);
    const double2 a = g[0], b = g[1], d = g[2], e = g[3];
    return Pose{a.x, a.y, b.x, b.y, d.x, d.y, e.x};
}
__device__ __forceinline__ Pose load_pose_lds(const char* win, int rel) {
    const double2* g = reinterpret_cast<const double2*>(win + rel * WIN_STRIDE);
    const double2 a = g[0], b = g[1], d = g[2], e = g[3];
    return Pose{a.x, a.y, b.x, b.y, d.x, d.y, e.x};
}
// one wavefront copies `n` 64-B pose records starting at keyframe `lo` into its LDS window (80-B stride)
__device__ __forceinline__ void stage_window(const double* __restrict__ pose8, int lo, int n, char* win, int lane) {
    const double2* src = reinterpret_cast<const double2*>(pose8 + (size_t)lo * 8);
    for (int u = lane; u < n * 4; u += 64) {
        const double2 v = src[u];                                   // 16 B/lane, 1 KiB per wave-instruction, fully coalesced
        *reinterpret_cast<double2*>(win + (u >> 2) * WIN_STRIDE + (u & 3) * 16) = v;
    }
}

// K1's output is write-once / read-later streaming data.  Measured on MI355X: with plain stores a 194-MB output (C3) is absorbed by
// the 256-MiB Infinity Cache (36.0 us = 6.16 TB/s; non-temporal 39.7 us), while a 775-MB output (400k keyframes) runs 188 us plain
// vs 140 us = 6.35 TB/s non-temporal.  launch_k1 picks the variant by output size.
template <bool NT>
__device__ __forceinline__ void k1_store(double2* p, double a, double b) {
    if (NT) {
        __builtin_nontemporal_store(a, reinterpret_cast<double*>(p));
        __builtin_nontemporal_store(b, reinterpret_cast<double*>(p) + 1);
    } else {
        *p = make_double2(a, b);
    }
}

template <bool WANT_J, bool NT>
__global__ __launch_bounds__(K1_WAVES * 64) void k1_edges_kernel(EdgeClassDev rel, EdgeClassDev sw, const double* __restrict__ pose8,
                                                                  const double* __restrict__ swv, double* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) char lds_win[K1_WAVES * 2 * WIN_MAX * WIN_STRIDE];
    __shared__ double red[K1_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;     // (NOT wave_in_block(): with a wave-uniform tile class the compiler keeps both residual paths' registers apart — 160 -> 200 VGPRs, 2 waves/SIMD)
    const int tile_g = blockIdx.x * K1_WAVES + wave;
    const bool active = tile_g < rel.tiles + sw.tiles;
    const bool is_sw = tile_g >= rel.tiles;
    const int tile = is_sw ? tile_g - rel.tiles : tile_g;
    const EdgeClassDev& C = is_sw ? sw : rel;
    char* win1 = lds_win + (wave * 2 + 0) * WIN_MAX * WIN_STRIDE;
    char* win2 = lds_win + (wave * 2 + 1) * WIN_MAX * WIN_STRIDE;
    int4 w = make_int4(0, 0, 0, 0);
    if (active) {
        w = C.win[tile];
        if (w.y > 0) stage_window(pose8, w.x, w.y, win1, lane);
        if (w.w > 0) stage_window(pose8, w.z, w.w, win2, lane);
    }
    __syncthreads();
    double cost = 0.0;
    if (active) {
        const int64_t e = (int64_t)tile * TILE + lane;
        const bool valid = e < C.E;
        const int32_t c1 = C.c1[e], c2 = C.c2[e];          // (padding lanes replicate the last edge: computed, never stored or counted)
        const Pose P1 = w.y > 0 ? load_pose_lds(win1, c1 - w.x) : load_pose_global(pose8, c1);
        const Pose P2 = w.w > 0 ? load_pose_lds(win2, c2 - w.z) : load_pose_global(pose8, c2);
        const double* mp = C.meas + e;
        const size_t ep = (size_t)C.Epad;
        const Meas M{mp[0], mp[ep], mp[2 * ep], mp[3 * ep], mp[4 * ep], mp[5 * ep], mp[6 * ep], mp[7 * ep]};
        if (!is_sw) {
            double r[6], J1[36], J2[36];
            relpose_residual<WANT_J>(P1, P2, M, M.w, r, J1, J2);
            if (valid) {
#pragma unroll
                for (int i = 0; i < 6; ++i) cost += r[i] * r[i];
            }
            double2* out = reinterpret_cast<double2*>(C.J) + (size_t)tile * (REL_DOUBLES / 2 * TILE) + lane;
            if (!valid) {
#pragma unroll
                for (int i = 0; i < 6; ++i) r[i] = 0.0;
                if (WANT_J) {
#pragma unroll
                    for (int i = 0; i < 36; ++i) { J1[i] = 0.0; J2[i] = 0.0; }
                }
            }
            if (WANT_J) {   // cost-only evaluation never touches the J buffers: the linearisation must survive a rejected step
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) k1_store<NT>(out + kp * TILE, r[2 * kp], r[2 * kp + 1]);
#pragma unroll
                for (int kp = 0; kp < 18; ++kp) k1_store<NT>(out + (3 + kp) * TILE, J1[2 * kp], J1[2 * kp + 1]);
#pragma unroll
                for (int kp = 0; kp < 18; ++kp) k1_store<NT>(out + (21 + kp) * TILE, J2[2 * kp], J2[2 * kp + 1]);
            }
        } else {
            const double s = valid ? swv[C.swidx[e]] : 0.0;
            double r[7], Js[7], J1[36], J2[36];
            switch_residual<WANT_J>(P1, P2, M, s, r, J1, J2, Js);
            if (valid) {
#pragma unroll
                for (int i = 0; i < 7; ++i) cost += r[i] * r[i];
            } else {
#pragma unroll
                for (int i = 0; i < 7; ++i) { r[i] = 0.0; Js[i] = 0.0; }
            }
            double2* out = reinterpret_cast<double2*>(C.J) + (size_t)tile * (SW_DOUBLES / 2 * TILE) + lane;
            if (WANT_J) {
                k1_store<NT>(out + 0 * TILE, r[0], r[1]);
                k1_store<NT>(out + 1 * TILE, r[2], r[3]);
                k1_store<NT>(out + 2 * TILE, r[4], r[5]);
                k1_store<NT>(out + 3 * TILE, r[6], Js[0]);
                k1_store<NT>(out + 4 * TILE, Js[1], Js[2]);
                k1_store<NT>(out + 5 * TILE, Js[3], Js[4]);
                k1_store<NT>(out + 6 * TILE, Js[5], Js[6]);
#pragma unroll
                for (int kp = 0; kp < 18; ++kp) k1_store<NT>(out + (7 + kp) * TILE, J1[2 * kp], J1[2 * kp + 1]);
#pragma unroll
                for (int kp = 0; kp < 18; ++kp) k1_store<NT>(out + (25 + kp) * TILE, J2[2 * kp], J2[2 * kp + 1]);
            }
        }
    }
    const double s = block_sum(cost, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// regularisers: a handful of unary blocks, one lane each, single workgroup
template <bool WANT_J>
__global__ __launch_bounds__(256) void prior_kernel(const PriorDev* __restrict__ pr, int n, const double* __restrict__ pose8,
                                                    double* __restrict__ Jp, double* __restrict__ cost_out) {
    __shared__ double red[4];
    double cost = 0.0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const PriorDev P = pr[k];
        const Pose X = load_pose_global(pose8, P.node);
        double r[6], J1[36];
        prior_residual<WANT_J>(X, P.Rf, P.tf, P.qf, P.w, r, J1);
        for (int i = 0; i < 6; ++i) cost += r[i] * r[i];
        if (WANT_J) {
            for (int i = 0; i < 6; ++i) Jp[(size_t)k * PRIOR_DOUBLES + i] = r[i];
            for (int i = 0; i < 36; ++i) Jp[(size_t)k * PRIOR_DOUBLES + 6 + i] = J1[i];
        }
    }
    const double s = block_sum(cost, red);
    if (threadIdx.x == 0) cost_out[0] = s;
}

void launch_k1(const GraphDev& G, const double* pose8, const double* sw, bool want_jacobian, double* partials, int* n_partials, hipStream_t st) {
    const int tiles = G.rel.tiles + G.sw.tiles;
    const int grid = (tiles + K1_WAVES - 1) / K1_WAVES;
    *n_partials = grid;
    if (grid == 0) return;
    const double out_bytes = 8.0 * TILE * ((double)G.rel.tiles * REL_DOUBLES + (double)G.sw.tiles * SW_DOUBLES);
    const bool nt = out_bytes > 224.0e6;   // beyond what the Infinity Cache absorbs (see k1_store)
    if (!want_jacobian) hipLaunchKernelGGL((k1_edges_kernel<false, false>), dim3(grid), dim3(K1_WAVES * 64), 0, st, G.rel, G.sw, pose8, sw, partials);
    else if (nt) hipLaunchKernelGGL((k1_edges_kernel<true, true>), dim3(grid), dim3(K1_WAVES * 64), 0, st, G.rel, G.sw, pose8, sw, partials);
    else hipLaunchKernelGGL((k1_edges_kernel<true, false>), dim3(grid), dim3(K1_WAVES * 64), 0, st, G.rel, G.sw, pose8, sw, partials);
}
void launch_prior(const GraphDev& G, const double* pose8, bool want_jacobian, double* partial_cost, hipStream_t st) {
    if (want_jacobian) hipLaunchKernelGGL(prior_kernel<true>, dim3(1), dim3(256), 0, st, G.prior, G.n_prior, pose8, G.Jp, partial_cost);
    else hipLaunchKernelGGL(prior_kernel<false>, dim3(1), dim3(256), 0, st, G.prior, G.n_prior, pose8, G.Jp, partial_cost);
}

double k1_algorithmic_bytes(const GraphDev& G, bool want_jacobian) {
    // in = 56 N + 8 E_s + 72 E + 68 E_g ; out = 624 E_r + 688 E_s + 336 E_g   (cost-only: inputs only)
    const double N = (double)G.N, Er = (double)G.rel.E, Es = (double)G.sw.E, Eg = (double)G.n_prior;
    double b = 56.0 * N + 8.0 * Es + 72.0 * (Er + Es) + 68.0 * Eg;
    if (want_jacobian) b += 624.0 * Er + 688.0 * Es + 336.0 * Eg;
    return b;
}

// ------------------------------------------------------------------------------------------------
// K2 — normal-equation assembly
// ------------------------------------------------------------------------------------------------
// one lane per keyframe: walk its incident (edge, side) list in a fixed order and accumulate J^T J and J^T r
__global__ __launch_bounds__(256) void k2_node_kernel(GraphDev G, LinDev L) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= G.N) return;
    double H[21], g[6];
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
    const int64_t b = G.inc_rowptr[n], e = G.inc_rowptr[n + 1];
    const int64_t slot_sw = G.rel.Epad, slot_pr = G.rel.Epad + G.sw.Epad;
    for (int64_t k = b; k < e; ++k) {
        const int64_t ent = G.inc[k];
        const int64_t slot = ent >> 1;
        const int side = (int)(ent & 1);
        double r[6], J[36];
        if (slot < slot_sw) {
            const double2* base = reinterpret_cast<const double2*>(G.rel.J + tile_elem(REL_DOUBLES, slot, 0));
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) { const double2 v = base[kp * TILE]; r[2 * kp] = v.x; r[2 * kp + 1] = v.y; }
            const int o = side ? 21 : 3;
#pragma unroll
            for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(o + kp) * TILE]; J[2 * kp] = v.x; J[2 * kp + 1] = v.y; }
        } else if (slot < slot_pr) {
            const double2* base = reinterpret_cast<const double2*>(G.sw.J + tile_elem(SW_DOUBLES, slot - slot_sw, 0));
#pragma unroll
            for (int kp = 0; kp < 3; ++kp) { const double2 v = base[kp * TILE]; r[2 * kp] = v.x; r[2 * kp + 1] = v.y; }
            const int o = side ? 25 : 7;
#pragma unroll
            for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(o + kp) * TILE]; J[2 * kp] = v.x; J[2 * kp + 1] = v.y; }
        } else {
            const double* base = G.Jp + (size_t)(slot - slot_pr) * PRIOR_DOUBLES;
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = base[i];
#pragma unroll
            for (int i = 0; i < 36; ++i) J[i] = base[6 + i];
        }
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double ga = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) ga += J[i * 6 + a] * r[i];
            g[a] += ga;
#pragma unroll
            for (int c = a; c < 6; ++c) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) s += J[i * 6 + a] * J[i * 6 + c];
                H[idx++] += s;
            }
        }
    }
    double* Hd = L.Hd + (size_t)n * 36;
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = a; c < 6; ++c) { Hd[a * 6 + c] = H[idx]; Hd[c * 6 + a] = H[idx]; ++idx; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) L.g[(size_t)n * 6 + i] = g[i];
}

// one lane per edge: off-diagonal block J1^T J2 and, for switchable edges, [J1^T Js ; J2^T Js], Js^T Js, Js^T r
__global__ __launch_bounds__(256) void k2_edge_kernel(GraphDev G, LinDev L, int want_offdiag) {
    const int64_t nrel = G.rel.Epad;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + (want_offdiag ? 0 : nrel);
    if (gid >= nrel + G.sw.Epad) return;
    const bool is_sw = gid >= nrel;
    const int64_t e = is_sw ? gid - nrel : gid;
    const EdgeClassDev& C = is_sw ? G.sw : G.rel;
    if (e >= C.E) return;
    double J1[36], J2[36];
    const int D = is_sw ? SW_DOUBLES : REL_DOUBLES;
    const double2* base = reinterpret_cast<const double2*>(C.J + tile_elem(D, e, 0));
    const int o1 = is_sw ? 7 : 3, o2 = is_sw ? 25 : 21;
#pragma unroll
    for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(o1 + kp) * TILE]; J1[2 * kp] = v.x; J1[2 * kp + 1] = v.y; }
#pragma unroll
    for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(o2 + kp) * TILE]; J2[2 * kp] = v.x; J2[2 * kp + 1] = v.y; }
    if (want_offdiag) {
        double* H = L.Hoff + (size_t)gid * 36;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) s += J1[i * 6 + a] * J2[i * 6 + c];
                H[a * 6 + c] = s;
            }
        }
    }
    if (is_sw) {
        double r[7], Js[7];
        { double2 v;
          v = base[0 * TILE]; r[0] = v.x; r[1] = v.y;  v = base[1 * TILE]; r[2] = v.x; r[3] = v.y;  v = base[2 * TILE]; r[4] = v.x; r[5] = v.y;
          v = base[3 * TILE]; r[6] = v.x; Js[0] = v.y; v = base[4 * TILE]; Js[1] = v.x; Js[2] = v.y;
          v = base[5 * TILE]; Js[3] = v.x; Js[4] = v.y; v = base[6 * TILE]; Js[5] = v.x; Js[6] = v.y; }
        double hss = 0.0, gs = 0.0;
#pragma unroll
        for (int i = 0; i < 7; ++i) { hss += Js[i] * Js[i]; gs += Js[i] * r[i]; }
        double* c = L.c + (size_t)e * 12;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) { s1 += J1[i * 6 + a] * Js[i]; s2 += J2[i * 6 + a] * Js[i]; }
            c[a] = s1; c[6 + a] = s2;
        }
        L.hss[e] = hss; L.gs[e] = gs;
    }
}

// K2 on the matrix-free operator's tiles (round 2): one lane per matvec lane — an edge side, or BOTH sides of an edge whose keyframes share the tile —
// loads the side's 6 x 6 Jacobian block and r6 from K1's output (21 or 39 double2 in flight per lane before the first use: consecutive lanes are
// consecutive edges of a few interleaved streams, so the loads coalesce into 256-B runs) and writes the 21 + 6 numbers of J^T J (upper triangle) and
// J^T r into the side's LDS slot (two rounds of 14 and 13 outputs: 43 KB of LDS, so that a CU holds three workgroups in different phases; with all 27 at
// once, 83 KB and one workgroup per CU, the kernel measured 90 us on C3); then one lane per (keyframe, output) sums the keyframe's slots in incident-list
// order — the order of the lane-per-keyframe kernel above, so the sums are bit-identical to it — adds the regulariser and stores Hd (both triangles) and g.
constexpr int K2_OUT = 27, K2_HALF = 14;      // outputs per keyframe: 21 of the upper triangle + 6 of g; done in two rounds of <= 14 so that two to three workgroups share a CU's LDS
template <int LO, int HI>
__device__ __forceinline__ void k2_side_products(const double* J, const double* r, double* out /* LDS slot, K2_HALF doubles */) {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = a; c < 6; ++c) {
            if (idx >= LO && idx < HI) {
                double sum = 0.0;
#pragma unroll
                for (int i = 0; i < 6; ++i) sum += J[i * 6 + a] * J[i * 6 + c];
                out[idx - LO] = sum;
            }
            ++idx;
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        if (21 + a >= LO && 21 + a < HI) {
            double ga = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) ga += J[i * 6 + a] * r[i];
            out[21 + a - LO] = ga;
        }
    }
}
template <int LO, int HI>
__device__ __forceinline__ void k2_reduce_round(const GraphDev& G, const LinDev& L, const MfDev& F, const double* sl, int n0, int nn, int l) {
    constexpr int W = HI - LO;
    for (int t = l; t < nn * W; t += MF_BLOCK) {
        const int nl = t / W, j = LO + (t - nl * W);
        const int64_t node = (int64_t)n0 + nl;
        const ushort4 rg = F.node_rng[node];
        double acc = 0.0;
        for (int q = rg.x; q < rg.y; ++q) acc += sl[q * K2_HALF + (j - LO)];
        for (int q = rg.z; q < rg.w; ++q) acc += sl[q * K2_HALF + (j - LO)];
        // which output this is: j < 21 -> (a, c) of the upper triangle in row-major order, else g[j - 21]
        int a = 0, c = j;
        if (j < 21) { int left = j; while (left >= 6 - a) { left -= 6 - a; ++a; } c = a + left; }
        const int32_t pk = F.node_prior[node];
        if (pk >= 0) {
            const double* base = G.Jp + (size_t)pk * PRIOR_DOUBLES;
            double sum = 0.0;
            if (j < 21) { for (int ii = 0; ii < 6; ++ii) sum += base[6 + ii * 6 + a] * base[6 + ii * 6 + c]; }
            else { for (int ii = 0; ii < 6; ++ii) sum += base[6 + ii * 6 + (j - 21)] * base[ii]; }
            acc += sum;
        }
        if (j < 21) { L.Hd[(size_t)node * 36 + a * 6 + c] = acc; L.Hd[(size_t)node * 36 + c * 6 + a] = acc; }
        else L.g[(size_t)node * 6 + (j - 21)] = acc;
    }
}
__global__ __launch_bounds__(MF_BLOCK) void k2_tiles_kernel(GraphDev G, LinDev L, MfDev F) {
    __shared__ double sl[MF_SLOTS * K2_HALF];
    const int tile = blockIdx.x, l = threadIdx.x;
    const int64_t i0 = F.tile_inc0[tile], i1 = F.tile_inc0[tile + 1];
    const int32_t n0 = F.tile_node0[tile], n1 = F.tile_node0[tile + 1];
    const int sw0 = F.tile_sw0[tile] & 0xffff, pair1 = (int)((uint32_t)F.tile_sw0[tile] >> 16);
    const int64_t i = i0 + l;
    const bool have = i < i1;
    const bool pair = have && l < pair1;
    int slot_a = 0, slot_b = 0;
    double r[6], Ja[36], Jb[36];
    if (have) {
        const uint32_t ent = F.einc[i];
        const uint32_t slw = F.einc_slot[i];
        slot_a = (int)(slw & 511u); slot_b = (int)((slw >> 9) & 511u);
        const bool is_sw = l >= sw0;
        const int side = (int)(ent & 1u);
        const int64_t e = (int64_t)((ent & 0x7fffffffu) >> 1);
        const double2* base = reinterpret_cast<const double2*>((is_sw ? G.sw.J : G.rel.J) + tile_elem(is_sw ? SW_DOUBLES : REL_DOUBLES, e, 0));
        const int o1 = is_sw ? 7 : 3, o2 = is_sw ? 25 : 21;
#pragma unroll
        for (int kp = 0; kp < 3; ++kp) { const double2 v = base[kp * TILE]; r[2 * kp] = v.x; r[2 * kp + 1] = v.y; }
        const int oa = side ? o2 : o1;
#pragma unroll
        for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(oa + kp) * TILE]; Ja[2 * kp] = v.x; Ja[2 * kp + 1] = v.y; }
        if (pair) {
#pragma unroll
            for (int kp = 0; kp < 18; ++kp) { const double2 v = base[(o2 + kp) * TILE]; Jb[2 * kp] = v.x; Jb[2 * kp + 1] = v.y; }
        }
        k2_side_products<0, K2_HALF>(Ja, r, sl + slot_a * K2_HALF);
        if (pair) k2_side_products<0, K2_HALF>(Jb, r, sl + slot_b * K2_HALF);
    }
    __syncthreads();
    const int nn = n1 - n0;
    k2_reduce_round<0, K2_HALF>(G, L, F, sl, n0, nn, l);
    __syncthreads();
    if (have) {
        k2_side_products<K2_HALF, K2_OUT>(Ja, r, sl + slot_a * K2_HALF);
        if (pair) k2_side_products<K2_HALF, K2_OUT>(Jb, r, sl + slot_b * K2_HALF);
    }
    __syncthreads();
    k2_reduce_round<K2_HALF, K2_OUT>(G, L, F, sl, n0, nn, l);
}

void launch_k2(const GraphDev& G, const LinDev& L, bool want_offdiag, hipStream_t st, const MfDev* F) {
    if (F && F->tiles > 0) hipLaunchKernelGGL(k2_tiles_kernel, dim3((unsigned)F->tiles), dim3(MF_BLOCK), 0, st, G, L, *F);
    else if (G.N > 0) hipLaunchKernelGGL(k2_node_kernel, dim3((unsigned)((G.N + 255) / 256)), dim3(256), 0, st, G, L);
    // the matrix-free operator needs only the switch couplings: skip the relpose part of the edge range entirely
    const int64_t first = want_offdiag ? 0 : G.rel.Epad;
    const int64_t ne = G.rel.Epad + G.sw.Epad - first;
    if (ne > 0) hipLaunchKernelGGL(k2_edge_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, G, L, want_offdiag ? 1 : 0);
}

// J1^T J2 of every edge into L.Hoff (36 contiguous doubles per edge) for the multigrid's level-1 Galerkin product under the matrix-free solver, which does not
// form them otherwise: one edge-parallel pass, the Jacobian loads coalesce (the switch couplings are rewritten with the same values)
void launch_k2_offdiag(const GraphDev& G, const LinDev& L, hipStream_t st) {
    const int64_t ne = G.rel.Epad + G.sw.Epad;
    if (ne > 0) hipLaunchKernelGGL(k2_edge_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, G, L, 1);
}

// ------------------------------------------------------------------------------------------------
// Jacobi scaling and LM diagonal (Ceres: scale = 1/(1+sqrt(col norm^2)) fixed at iteration 0;
// D^2 = clamp(col norm^2 of the scaled Jacobian, min, max) / radius)
// ------------------------------------------------------------------------------------------------
__global__ void scale_init_kernel(GraphDev G, LinDev L, ScaleDev Sc, int jacobi_scaling) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t np = G.N * 6;
    if (i < np) {
        const int64_t n = i / 6; const int c = (int)(i % 6);
        Sc.scale_p[i] = jacobi_scaling ? 1.0 / (1.0 + sqrt(L.Hd[(size_t)n * 36 + c * 6 + c])) : 1.0;
    } else if (i < np + G.sw.E) {
        const int64_t e = i - np;
        Sc.scale_s[e] = jacobi_scaling ? 1.0 / (1.0 + sqrt(L.hss[e])) : 1.0;
    }
}
__global__ void lm_diag_kernel(GraphDev G, LinDev L, ScaleDev Sc, double mn, double mx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t np = G.N * 6;
    if (i < np) {
        const int64_t n = i / 6; const int c = (int)(i % 6);
        const double s = Sc.scale_p[i];
        Sc.diag_p[i] = fmin(fmax(s * s * L.Hd[(size_t)n * 36 + c * 6 + c], mn), mx);
    } else if (i < np + G.sw.E) {
        const int64_t e = i - np;
        const double s = Sc.scale_s[e];
        Sc.diag_s[e] = fmin(fmax(s * s * L.hss[e], mn), mx);
    }
}
void launch_scale_init(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, int jacobi_scaling, hipStream_t st) {
    const int64_t n = G.N * 6 + G.sw.E;
    if (n > 0) hipLaunchKernelGGL(scale_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, L, Sc, jacobi_scaling);
}
void launch_lm_diag(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, double mn, double mx, hipStream_t st) {
    const int64_t n = G.N * 6 + G.sw.E;
    if (n > 0) hipLaunchKernelGGL(lm_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, L, Sc, mn, mx);
}

// ------------------------------------------------------------------------------------------------
// LM system build: Schur-eliminate the switch variables edge by edge (each switch occurs in exactly one
// residual block, reference src/PoseGraphSLAM.cpp:1555, so H_ss is diagonal), add the LM damping in the
// unscaled space (lambda_j = D_j^2 / scale_j^2) and write the block-CSR rows.
// ------------------------------------------------------------------------------------------------
__global__ void sw_prepare_kernel(GraphDev G, LinDev L, ScaleDev Sc, double radius) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G.sw.E) return;
    const double s = Sc.scale_s[e];
    Sc.a_inv[e] = 1.0 / (L.hss[e] + Sc.diag_s[e] / (radius * s * s));
}

__global__ __launch_bounds__(256) void build_rows_kernel(GraphDev G, LinDev L, ScaleDev Sc, CgDev C, double radius, int add_lambda, double* __restrict__ lam_out) {
    const bool write_val = lam_out == nullptr;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= G.N) return;
    const bool free_node = G.node_free[n] != 0;
    // Multi-GPU: Hd and g are already summed over the ranks sharing the keyframe (exchanged after K2), so exactly one rank — the
    // keyframe's OWNER — may put them and the damping into the system; the others contribute only their local switch Schur terms.
    if (G.own) add_lambda = G.own[n] != 0.0 ? 1 : 0;
    double D[36], bv[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) D[i] = add_lambda ? L.Hd[(size_t)n * 36 + i] : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) bv[i] = add_lambda ? -L.g[(size_t)n * 6 + i] : 0.0;
    const int64_t b = G.inc_rowptr[n], e = G.inc_rowptr[n + 1];
    const int64_t slot_sw = G.rel.Epad, slot_pr = G.rel.Epad + G.sw.Epad;
    const int64_t row0 = G.bsr_rowptr[n];
    for (int64_t k = b; k < e; ++k) {
        const int64_t ent = G.inc[k];
        const int64_t slot = ent >> 1;
        const int side = (int)(ent & 1);
        if (slot >= slot_pr) break;   // regularisers come last in the list; they are already in Hd
        double B[36];
        const double* H = L.Hoff + (size_t)slot * 36;
        if (!write_val) {
#pragma unroll
            for (int i = 0; i < 36; ++i) B[i] = 0.0;
        } else if (side == 0) {
#pragma unroll
            for (int i = 0; i < 36; ++i) B[i] = H[i];
        } else {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = 0; c < 6; ++c) B[a * 6 + c] = H[c * 6 + a];
        }
        if (slot >= slot_sw) {
            const int64_t es = slot - slot_sw;
            const double ai = Sc.a_inv[es];
            const double* cc = L.c + (size_t)es * 12;
            double cs[6], co[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { cs[i] = cc[side * 6 + i]; co[i] = cc[(1 - side) * 6 + i]; }
            const double gsa = L.gs[es] * ai;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                bv[a] += cs[a] * gsa;
#pragma unroll
                for (int c = 0; c < 6; ++c) { D[a * 6 + c] -= cs[a] * cs[c] * ai; B[a * 6 + c] -= cs[a] * co[c] * ai; }
            }
        }
        if (!write_val) continue;
        double* out = C.val + (size_t)(row0 + 1 + (k - b)) * 36;
        if (!free_node) {
#pragma unroll
            for (int i = 0; i < 36; ++i) B[i] = 0.0;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < 6; ++c) out[pm(c, a)] = B[a * 6 + c];   // BSR blocks: COLUMN-pair-major (lane = column)
    }
    double lam[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (add_lambda) {
#pragma unroll
        for (int c = 0; c < 6; ++c) { const double s = Sc.scale_p[(size_t)n * 6 + c]; lam[c] = Sc.diag_p[(size_t)n * 6 + c] / (radius * s * s); D[c * 6 + c] += lam[c]; }
    }
    if (lam_out) {
#pragma unroll
        for (int c = 0; c < 6; ++c) lam_out[(size_t)n * 6 + c] = free_node ? lam[c] : (add_lambda ? 1.0 : 0.0);   // fixed keyframes: identity row
    }
    if (!free_node) {
#pragma unroll
        for (int i = 0; i < 36; ++i) D[i] = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) { D[c * 6 + c] = add_lambda ? 1.0 : 0.0; bv[c] = 0.0; }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) { if (write_val) C.val[(size_t)row0 * 36 + pm(c, a)] = D[a * 6 + c]; C.Dtot[(size_t)n * 36 + a * 6 + c] = D[a * 6 + c]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) C.b[(size_t)n * 6 + i] = bv[i];
}

// Block-Jacobi preconditioner in FACTORED form: the lower Cholesky factor L of every (Schur-reduced, damped) 6x6 diagonal block, packed
// (21 entries, diagonal stored as 1/L_ii) and rounded to fp32 — 96 B per keyframe instead of 288 B for an fp64 inverse.  M = L~ L~^T
// is symmetric positive definite whatever the rounding did, so PCG stays valid; the arithmetic applying it is fp64.
constexpr int LF_STRIDE = 24;   // floats per keyframe (21 used): 6 x 16-B loads
__device__ __forceinline__ int lf_idx(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

__global__ __launch_bounds__(256) void invert_rows_kernel(GraphDev G, CgDev C, int32_t* fail) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= G.N) return;
    double D[36], Lm[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) D[i] = C.Dtot[(size_t)n * 36 + i];
    if (!G.node_free[n]) {
#pragma unroll
        for (int i = 0; i < 36; ++i) D[i] = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) D[c * 6 + c] = 1.0;
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = D[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= Lm[j * 6 + k] * Lm[j * 6 + k];
        ok = ok && (d > 0.0);
        d = sqrt(d);
        Lm[j * 6 + j] = d;
        const double di = 1.0 / d;
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i > j) {
            double sacc = D[i * 6 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) sacc -= Lm[i * 6 + k] * Lm[j * 6 + k];
            Lm[i * 6 + j] = sacc * di;
        }
    }
    if (!ok) atomicOr(fail, 1);
    float out[LF_STRIDE];
#pragma unroll
    for (int i = 0; i < LF_STRIDE; ++i) out[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) if (j <= i) out[lf_idx(i, j)] = (float)(i == j ? 1.0 / Lm[i * 6 + i] : Lm[i * 6 + j]);
    float4* dst = reinterpret_cast<float4*>(C.Lf + (size_t)n * LF_STRIDE);
#pragma unroll
    for (int k = 0; k < LF_STRIDE / 4; ++k) dst[k] = make_float4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}

// z_r = (L L^T)^-1 a, component r, from the packed factor in LDS (diagonal inverted) and the keyframe's 6-vector a
__device__ __forceinline__ double lf_apply_row(const float* __restrict__ Lf, const double* __restrict__ a, int r) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sacc = a[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) if (j < i) sacc -= (double)Lf[lf_idx(i, j)] * y[j];
        y[i] = sacc * (double)Lf[lf_idx(i, i)];
    }
    double zz[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sacc = y[i];
#pragma unroll
        for (int j = 0; j < 6; ++j) if (j > i) sacc -= (double)Lf[lf_idx(j, i)] * zz[j];
        zz[i] = sacc * (double)Lf[lf_idx(i, i)];
    }
    double out = zz[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) out = (r == i) ? zz[i] : out;
    return out;
}
// the workgroup's CG_BLOCK/6 keyframes' factors -> LDS: exactly one coalesced 16-B load per lane
template <int KEYFRAMES>
__device__ __forceinline__ void lf_stage(const float* __restrict__ Lf_global, int64_t first_node, int64_t n_nodes_total, float* lds) {
    for (int f4 = threadIdx.x; f4 < KEYFRAMES * 6; f4 += blockDim.x) {   // 6 float4 per keyframe, consecutive lanes -> consecutive 16 B
        const int64_t node = first_node + f4 / 6;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node < n_nodes_total) v = reinterpret_cast<const float4*>(Lf_global + (size_t)first_node * LF_STRIDE)[f4];
        reinterpret_cast<float4*>(lds)[f4] = v;
    }
}
// both rows (2j, 2j+1) of z = (L L^T)^-1 a for the lane owning row pair j
__device__ __forceinline__ double2 lf_apply_pair(const float* __restrict__ Lf, const double* __restrict__ a, int j) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sacc = a[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < i) sacc -= (double)Lf[lf_idx(i, k)] * y[k];
        y[i] = sacc * (double)Lf[lf_idx(i, i)];
    }
    double zz[6];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sacc = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k > i) sacc -= (double)Lf[lf_idx(k, i)] * zz[k];
        zz[i] = sacc * (double)Lf[lf_idx(i, i)];
    }
    return j == 0 ? make_double2(zz[0], zz[1]) : j == 1 ? make_double2(zz[2], zz[3]) : make_double2(zz[4], zz[5]);
}

void launch_build_rows(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, double radius, int add_lambda, double* lam_out, hipStream_t st) {
    if (G.sw.E > 0) hipLaunchKernelGGL(sw_prepare_kernel, dim3((unsigned)((G.sw.E + 255) / 256)), dim3(256), 0, st, G, L, Sc, radius);
    if (G.N > 0) hipLaunchKernelGGL(build_rows_kernel, dim3((unsigned)((G.N + 255) / 256)), dim3(256), 0, st, G, L, Sc, C, radius, add_lambda, lam_out);
}
void launch_invert_rows(const GraphDev& G, const CgDev& C, int32_t* fail_flag, hipStream_t st) {
    if (G.N > 0) hipLaunchKernelGGL(invert_rows_kernel, dim3((unsigned)((G.N + 255) / 256)), dim3(256), 0, st, G, C, fail_flag);
}

// ------------------------------------------------------------------------------------------------
// K3/K4 — preconditioned conjugate gradients on the block-CSR system, block-Jacobi preconditioner.
// TWO kernels per iteration:
//   cg_spmv_kernel   beta = rz_k / rz_{k-1};  p_k = z + beta p_{k-1} (own row written to the ping-pong buffer, neighbours'
//                    rows recomputed on the fly from z and p_{k-1});  q = A p_k;  partial p.q
//   cg_update_kernel alpha = rz_k / p.q;  x += alpha p_k;  r' = r - alpha q;  z = Minv r';  partial r'.z
// One thread per (keyframe, row): 192-thread workgroups = 32 keyframes x 6 rows; blocks are pair-major so each load
// instruction reads a contiguous 96-B run per keyframe.  Dot products: per-workgroup partials (grid capped at MAX_PARTIALS),
// re-reduced in a fixed order by every workgroup of the consuming kernel -> no atomics, bitwise reproducible.
// ------------------------------------------------------------------------------------------------
constexpr int CG_BLOCK = 192;   // 32 keyframes x 6 lanes = 3 wavefronts

// workgroup-uniform read of the convergence flag (it may be raised by workgroup 0 of cg_spmv while other workgroups of the
// same launch are starting: one lane reads, LDS broadcasts, nobody diverges around a barrier)
__device__ __forceinline__ bool cg_done(const CgDev& C) {
    __shared__ int s_done;
    if (threadIdx.x == 0) s_done = C.flags[0];
    __syncthreads();
    return s_done != 0;
}

// One lane per (keyframe, COLUMN c): it streams column c of every block of the block-row (3 x 16 B, contiguous 96-B runs across the
// 6 lanes of a keyframe), needs ONE entry of the input vector per block (8-B gather instead of 48 B), and accumulates a private
// 6-vector of row partials.  The 6 lanes of a keyframe reduce once per row through LDS.  U blocks per step with every load issued
// before the first use (col -> gather is a dependent chain; the block columns do not depend on col at all).
template <int U, bool FUSED>
__device__ __forceinline__ void spmv_chunk(const int32_t* __restrict__ colp, const double* __restrict__ valp, int c, const double* __restrict__ z,
                                           const double* __restrict__ pprev, double beta, double* acc) {
    int32_t col[U];
    double2 v[U][3];
    double zz[U], pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) col[u] = colp[u];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const double2* vp = reinterpret_cast<const double2*>(valp + (size_t)u * 36) + c;
        v[u][0] = vp[0]; v[u][1] = vp[6]; v[u][2] = vp[12];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        zz[u] = z[(size_t)col[u] * 6 + c];
        pp[u] = FUSED ? pprev[(size_t)col[u] * 6 + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const double x = FUSED ? zz[u] + beta * pp[u] : zz[u];
        acc[0] += v[u][0].x * x; acc[1] += v[u][0].y * x; acc[2] += v[u][1].x * x; acc[3] += v[u][1].y * x; acc[4] += v[u][2].x * x; acc[5] += v[u][2].y * x;
    }
}

template <bool FUSED>
__device__ __forceinline__ void bsr_row_accumulate(const GraphDev& G, const double* __restrict__ val, const double* __restrict__ z,
                                                   const double* __restrict__ pprev, double beta, int64_t n, int c, double* acc) {
    const int64_t b = G.bsr_rowptr[n], e = G.bsr_rowptr[n + 1];
    int64_t k = b;
    for (; k + 4 <= e; k += 4) spmv_chunk<4, FUSED>(G.bsr_col + k, val + (size_t)k * 36, c, z, pprev, beta, acc);
    if (k + 2 <= e) { spmv_chunk<2, FUSED>(G.bsr_col + k, val + (size_t)k * 36, c, z, pprev, beta, acc); k += 2; }
    if (k < e) spmv_chunk<1, FUSED>(G.bsr_col + k, val + (size_t)k * 36, c, z, pprev, beta, acc);
}

// q = A (z + beta p_prev), p_cur = z + beta p_prev
__global__ __launch_bounds__(CG_BLOCK) void cg_spmv_kernel(GraphDev G, CgDev C, int parity, int first, int nparts, double tol2) {
    __shared__ double red[2 * (CG_BLOCK / 64) + 1];
    __shared__ double xch[CG_BLOCK * 7];   // [keyframe-in-group][c][r], padded to 7 to spread LDS banks
    double beta = 0.0;
    if (first) { if (cg_done(C)) return; }
    else {
        double rz_new, rz_old;
        if (block_total2_done(C.flags, C.part_rz + parity * RZ_STRIDE, nparts + C.extra_rz, C.part_rz + (parity ^ 1) * RZ_STRIDE, nparts + C.extra_rz, red, rz_new, rz_old)) return;
        const bool breakdown = C.flags[1] != 0;
        // convergence on the preconditioned residual norm: every workgroup evaluates the same numbers -> uniform exit
        if (breakdown || !(rz_new > C.scal[3] * C.scal[0])) {
            // (r.z clearly negative: the preconditioner is not positive definite — a breakdown, never convergence)
            if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[0] = 1; if (!breakdown) { C.scal[1] = rz_new; if (!(rz_new >= -C.scal[3] * C.scal[0])) C.flags[1] = 1; } }
            return;
        }
        beta = rz_new / rz_old;
        if (blockIdx.x == 0 && threadIdx.x == 0) C.scal[1] = rz_new;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) C.flags[2] += 1;
    const double* __restrict__ pprev = parity ? C.p : C.p2;
    double* __restrict__ pcur = parity ? C.p2 : C.p;
    const double* __restrict__ z = C.z;
    const int64_t rows = G.N * 6;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    double pq = 0.0;
    // all threads of the workgroup run the same number of trips (barriers inside)
    const int64_t trips = (rows + stride - 1) / stride;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t i = it * stride + (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x;
        const bool live = i < rows;
        const int64_t n = live ? i / 6 : 0; const int c = live ? (int)(i - n * 6) : 0;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (live) bsr_row_accumulate<true>(G, C.val, z, pprev, beta, n, c, acc);
        __syncthreads();
        double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
        for (int r = 0; r < 6; ++r) mine[r] = acc[r];
        __syncthreads();
        if (live) {
            // lane (n, c) now plays row r = c: sum the c-th partial of the keyframe's 6 lanes
            const double* grp = xch + (size_t)(threadIdx.x - c) * 7;
            double q = 0.0;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) q += grp[cc * 7 + c];
            const double pi = z[i] + beta * pprev[i];
            pcur[i] = pi;
            C.q[i] = q;
            pq += q * pi;
        }
    }
    const double s = block_sum(pq, red);
    if (threadIdx.x == 0) C.part_pq[blockIdx.x] = s;
}

__global__ __launch_bounds__(CG_BLOCK) void apply_operator_kernel(GraphDev G, CgDev C, const double* __restrict__ x, double* __restrict__ y) {
    __shared__ double xch[CG_BLOCK * 7];
    const int64_t rows = G.N * 6;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    const int64_t trips = (rows + stride - 1) / stride;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t i = it * stride + (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x;
        const bool live = i < rows;
        const int64_t n = live ? i / 6 : 0; const int c = live ? (int)(i - n * 6) : 0;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (live) bsr_row_accumulate<false>(G, C.val, x, nullptr, 0.0, n, c, acc);
        __syncthreads();
        double* mine = xch + (size_t)threadIdx.x * 7;
#pragma unroll
        for (int r = 0; r < 6; ++r) mine[r] = acc[r];
        __syncthreads();
        if (live) {
            const double* grp = xch + (size_t)(threadIdx.x - c) * 7;
            double q = 0.0;
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) q += grp[cc * 7 + c];
            y[i] = q;
        }
    }
}

// cold: x = 0, r = b.   warm (after a rejected step: same H, larger damping): x keeps the previous solution, r = b - A x (A x is in q).
// z = M^-1 r, partial r.z -> part_rz[0]; the convergence reference stays ||b||_{M^-1} (partials -> part_pq) in both cases.
__global__ __launch_bounds__(CG_BLOCK) void cg_init_kernel(GraphDev G, CgDev C, int warm) {
    __shared__ double red[CG_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) float lfs[CG_BLOCK / 6 * LF_STRIDE];
    __shared__ double av[CG_BLOCK], bv[CG_BLOCK];
    const int64_t rows = G.N * 6;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    const int64_t trips = (rows + stride - 1) / stride;
    double rz = 0.0, bb = 0.0;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t base = it * stride + (int64_t)blockIdx.x * CG_BLOCK;
        const int64_t i = base + threadIdx.x;
        const bool live = i < rows;
        const double bi = live ? C.b[i] : 0.0;
        const double ri = live ? (warm ? bi - C.q[i] : bi) : 0.0;
        __syncthreads();
        lf_stage<CG_BLOCK / 6>(C.Lf, base / 6, G.N, lfs);
        av[threadIdx.x] = ri; bv[threadIdx.x] = bi;
        __syncthreads();
        if (live) {
            const int r = (int)(i % 6);
            const float* Lf = lfs + (threadIdx.x / 6) * LF_STRIDE;
            const double z = lf_apply_row(Lf, av + (threadIdx.x - r), r);
            const double zb = warm ? lf_apply_row(Lf, bv + (threadIdx.x - r), r) : z;
            if (!warm) C.x[i] = 0.0;
            C.r[i] = ri; C.z[i] = z; C.p[i] = 0.0; C.p2[i] = 0.0;   // p buffers zeroed: iteration 0 multiplies them by beta = 0
            const double w = G.own ? G.own[i / 6] : 1.0;
            rz += w * ri * z; bb += w * bi * zb;
        }
    }
    const double s = block_sum(rz, red);
    const double sb = block_sum(bb, red);
    if (threadIdx.x == 0) { C.part_rz[blockIdx.x] = s; C.part_pq[blockIdx.x] = sb; }
}
// nparts: r.z partial slots in use (the start-up kernels' — or, two-level method in its fused form, the update kernel's, the tail zeroed by the caller);
// nparts_bb: slots of part_pq the start-up kernel (cg_init) filled with the partials of b.D^-1 b — ITS grid, whatever nparts is.  (Round 3 read `nparts` slots of both:
// with aggregates whose size does not divide 64 the fused update kernel has more workgroups than cg_init, and the reference norm of the stopping test picked up
// whatever the slots behind cg_init's held — stale p.q partials of an earlier PCG, or what the allocation contained: the box-dependent C2 result of GPUTEST_r03.)
__global__ void cg_scalars_init_kernel(CgDev C, int nparts, int nparts_bb, double tol2) {
    __shared__ double red[4];
    const double rz0 = block_total(C.part_rz, nparts + C.extra_rz, red);
    const double bb = block_total(C.part_pq, nparts_bb, red);
    if (threadIdx.x == 0) {
        C.scal[0] = bb; C.scal[1] = rz0; C.scal[2] = 0.0; C.scal[3] = tol2;
        // r0.z0 < 0: the preconditioner is not positive definite; NaN anywhere: both are breakdowns, never "converged at iteration 0"
        C.flags[0] = (bb > 0.0 && rz0 > 0.0) ? 0 : 1; C.flags[1] = (bb >= 0.0 && rz0 >= 0.0) ? 0 : 1; C.flags[2] = 0;
    }
}

// The head of the single-reduction (Chronopoulos-Gear) update on one GPU: delta = u.w (the matvec's partials) and gamma = r.u (the previous update's) are re-reduced
// TOGETHER — the iteration's only reduction point — then the convergence test the classic form runs at the head of its matvec, beta = gamma / gamma_prev,
// alpha = gamma / (delta - beta gamma / alpha_prev).  Scalars in C.scal[8 + 2 parity] (gamma), [9 + 2 parity] (alpha) of the iteration with that parity (as the multi-rank
// kernel keeps them).  Returns false when the workgroup has nothing to do (stopped, converged, broken down).
// split_coarse (the two-level method's fused iteration): the slots behind the update kernel's `nparts` hold the coarse part rc.Ac^-1 rc of r.u (coarse_solve_dot_kernel); a
// convergence that the block-Jacobi part alone does not confirm is a breakdown of the fp32-rounded coarse inverse, never a converged step (the classic form runs the same
// test at the head of its matvec: mf_spmv_kernel<true, true>); coarse_only: one aggregate per keyframe, u is the coarse term alone.  red: 3 x waves + 1 doubles then.
__device__ __forceinline__ bool sr_head(const CgDev& C, int parity, int first, int nparts_pq, int nparts, double* red, double& alpha, double& beta, bool split_coarse = false, bool coarse_only = false) {
    double delta, gamma;
    bool coarse_negative = false;
    if (split_coarse) {
        double g_bj, g_c;
        if (block_total3_done(C.flags, C.part_pq, nparts_pq, C.part_rz + parity * RZ_STRIDE, nparts, C.part_rz + parity * RZ_STRIDE + nparts, C.extra_rz, red, delta, g_bj, g_c)) return false;
        gamma = g_bj + g_c;
        coarse_negative = !coarse_only && !(gamma > C.scal[3] * C.scal[0]) && g_bj > C.scal[3] * C.scal[0];
    } else if (block_total2_done(C.flags, C.part_pq, nparts_pq, C.part_rz + parity * RZ_STRIDE, nparts + C.extra_rz, red, delta, gamma)) return false;
    const bool breakdown = C.flags[1] != 0;
    if (breakdown || !(gamma > C.scal[3] * C.scal[0])) {      // converged (or broken down): the state stays that of the last completed update
        if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[0] = 1; if (!breakdown) { C.scal[1] = gamma; if (coarse_negative || !(gamma >= -C.scal[3] * C.scal[0])) C.flags[1] = 1; } }
        return false;
    }
    beta = 0.0;
    double den = delta;
    if (!first) {
        const double gamma_prev = C.scal[8 + 2 * (parity ^ 1)], alpha_prev = C.scal[9 + 2 * (parity ^ 1)];
        beta = gamma / gamma_prev;
        den = delta - beta * gamma / alpha_prev;
    }
    if (!(den > 0.0)) {      // not positive definite along the direction (or NaN): x is left untouched
        if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[1] = 1; C.flags[0] = 1; }
        return false;
    }
    alpha = gamma / den;
    if (blockIdx.x == 0 && threadIdx.x == 0) { C.scal[8 + 2 * parity] = gamma; C.scal[9 + 2 * parity] = alpha; C.scal[1] = gamma; C.flags[2] += 1; }
    return true;
}

// classic form (SR = false):   alpha = rz/pq ; x += alpha p ; r' = r - alpha q ; z = Minv r' ; partial r'.z -> part_rz[parity^1]
// single-reduction form (SR):  p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = Minv r ; partial r.u -> part_rz[parity^1]     (u in C.z, w in C.q, s in C.p2, r in place)
template <bool SR>
__global__ __launch_bounds__(CG_BLOCK) void cg_update_kernel(GraphDev G, CgDev C, int parity, int nparts_pq, int nparts, int first) {
    __shared__ double red[2 * (CG_BLOCK / 64) + 1];
    const double2* __restrict__ rin = reinterpret_cast<const double2*>(SR ? C.r : (parity ? C.r2 : C.r));
    double2* __restrict__ rout = reinterpret_cast<double2*>(SR ? C.r : (parity ? C.r : C.r2));
    const double2* __restrict__ pcur = reinterpret_cast<const double2*>(SR ? C.p : (parity ? C.p2 : C.p));
    const double2* __restrict__ qv = reinterpret_cast<const double2*>(C.q);
    double2* __restrict__ xv = reinterpret_cast<double2*>(C.x);
    double2* __restrict__ zv = reinterpret_cast<double2*>(C.z);
    double2* __restrict__ pout = reinterpret_cast<double2*>(C.p);      // (SR only)
    double2* __restrict__ sv = reinterpret_cast<double2*>(C.p2);       // (SR only) s = A p
    // one lane per (keyframe, ROW PAIR): every vector access is 16 B/lane (8-B accesses reach only ~0.6x of the streaming rate);
    // a workgroup covers CG_BLOCK/3 = 64 keyframes per trip
    constexpr int KF = CG_BLOCK / 3;
    const int64_t pairs = G.N * 3;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    const int64_t trips = (pairs + stride - 1) / stride;
    // the first trip's operands do not depend on alpha: issue their loads BEFORE the partial-sum re-reduction so that its ~2 us of
    // dependent L2 round trips overlap with the streaming loads.  The block-Jacobi factors of the trip's keyframes (two 16-B loads per lane) travel
    // with them — staged through registers, not behind a barrier of their own — and a second trip's operands are requested while the first
    // trip's z is being formed: per trip ONE round trip and one barrier on the critical path.
    double2 u0 = make_double2(0.0, 0.0), s0 = u0;      // (SR) the preconditioned residual u and s = A p of the trip
    auto load_trip = [&](int64_t base, double2& r_, double2& q_, double2& p_, double2& x_, float4& l0, float4& l1) {
        const int64_t i = base + threadIdx.x;
        r_ = make_double2(0.0, 0.0); q_ = r_; p_ = r_; x_ = r_; l0 = make_float4(0.f, 0.f, 0.f, 0.f); l1 = l0;
        if (SR) { u0 = r_; s0 = r_; }
        if (i < pairs) { r_ = rin[i]; q_ = qv[i]; p_ = pcur[i]; x_ = xv[i]; if (SR) { u0 = zv[i]; s0 = sv[i]; } }
        const int64_t first_node = base / 3;
        const float4* lp = reinterpret_cast<const float4*>(C.Lf + (size_t)first_node * LF_STRIDE);
        if (first_node + threadIdx.x / 6 < G.N) l0 = lp[threadIdx.x];
        if (first_node + (threadIdx.x + CG_BLOCK) / 6 < G.N) l1 = lp[threadIdx.x + CG_BLOCK];
    };
    double2 r0, q0, p0, x0; float4 lf0, lf1;
    load_trip((int64_t)blockIdx.x * CG_BLOCK, r0, q0, p0, x0, lf0, lf1);
    double alpha = 0.0, beta = 0.0;
    // (after the first trip's loads are in flight: the flag's and the partial sums' round trip overlaps with theirs)
    if (SR) { if (!sr_head(C, parity, first, nparts_pq, nparts, red, alpha, beta)) return; }
    else {
        double pq, rz;   // pq partials are produced by the matvec kernel (its own grid size)
        if (block_total2_done(C.flags, C.part_pq, nparts_pq, C.part_rz + parity * RZ_STRIDE, nparts + C.extra_rz, red, pq, rz)) return;
        if (!(pq > 0.0)) {   // breakdown: matrix not positive definite along p (or NaN); x is left untouched, the next spmv raises done
            if (blockIdx.x == 0 && threadIdx.x == 0) C.flags[1] = 1;
            if (threadIdx.x == 0) C.part_rz[(parity ^ 1) * RZ_STRIDE + blockIdx.x] = 0.0;
            return;
        }
        alpha = rz / pq;
    }
    __shared__ double2 rnew[CG_BLOCK];
    __shared__ __attribute__((aligned(16))) float lfs[KF * LF_STRIDE];
    static_assert(KF * LF_STRIDE / 4 == 2 * CG_BLOCK, "two 16-B loads per lane stage a trip's factors");
    double acc = 0.0;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t base = it * stride + (int64_t)blockIdx.x * CG_BLOCK;
        const int64_t i = base + threadIdx.x;
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
        reinterpret_cast<float4*>(lfs)[threadIdx.x] = lf0; reinterpret_cast<float4*>(lfs)[threadIdx.x + CG_BLOCK] = lf1;
        rnew[threadIdx.x] = rr;
        __syncthreads();
        if (it + 1 < trips) load_trip(base + stride, r0, q0, p0, x0, lf0, lf1);
        if (live) {
            const int j = (int)(i % 3);
            const double2 z = lf_apply_pair(lfs + (threadIdx.x / 3) * LF_STRIDE, reinterpret_cast<const double*>(rnew + (threadIdx.x - j)), j);
            zv[i] = z;
            const double w = G.own ? G.own[i / 3] : 1.0;
            acc += w * (rr.x * z.x + rr.y * z.y);
        }
        if (it + 1 < trips) __syncthreads();      // the LDS buffers are rewritten by the next trip
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) C.part_rz[(parity ^ 1) * RZ_STRIDE + blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU PCG: Chronopoulos-Gear form of the same block-Jacobi PCG.  Both dot products of an iteration — gamma = r.u (u = M^-1 r,
// owner-weighted partials left by the previous update) and delta = u.(A u) = sum over ranks of u.(A_r u) (rank-local partials) — are
// available right after the matvec, so they ride on the ONE all-reduce that sums the shared rows of w = A u:
//     w = A u ; [gamma, delta, shared rows of w] summed over ranks ;
//     beta = gamma/gamma_prev ; alpha = gamma / (delta - beta gamma / alpha_prev) ; p = u + beta p ; s = w + beta s ;
//     x += alpha p ; r -= alpha s ; u = M^-1 r
// Same iterates as the standard recurrence in exact arithmetic; measured on the C3-structured 20k system: same iteration counts and
// true residuals down to 1e-12 (scripts/research/precond_probe.py::probe10).  Scalars live in C.scal[8..15]:
//     [8 + 2*parity] gamma, [9 + 2*parity] alpha of the iteration with that parity ; [12] delta, [13] gamma as exchanged.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CG_BLOCK) void cgcg_dots_kernel(GraphDev G, CgDev C) {
    __shared__ double red[CG_BLOCK / 64];
    const int64_t pairs = G.N * 3;     // 16 B per lane (8-B accesses reach only ~0.6x of the streaming rate)
    const double2* __restrict__ zv = reinterpret_cast<const double2*>(C.z);
    const double2* __restrict__ qv = reinterpret_cast<const double2*>(C.q);
    double d = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * CG_BLOCK) { const double2 a = zv[i], b = qv[i]; d += a.x * b.x + a.y * b.y; }
    const double s = block_sum(d, red);
    if (threadIdx.x == 0) C.part_pq[blockIdx.x] = s;
}
// out[0] = sum pa, out[1] = sum pb — or zeros once the PCG has stopped (a stopped PCG keeps its state; the exchanged scalars are scratch)
__global__ void cg_reduce2_live_kernel(CgDev C, const double* __restrict__ pa, int na, const double* __restrict__ pb, int nb, double* __restrict__ out) {
    __shared__ double red[8];
    const bool stopped = C.flags[0] != 0;
    double a = 0.0, b = 0.0;
    if (!stopped) block_total2(pa, na, pb, nb, red, a, b);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}
// xq / sh_of / two (fused exchange): the summed rows of w for shared keyframes are read straight from the exchange buffer (sh_of[keyframe] = its position there, or -1)
// and [delta, gamma] from its tail — no unpack kernel, no copies; null: w complete in C.q, scalars in C.scal[12..13]
__global__ __launch_bounds__(CG_BLOCK) void cgcg_update_kernel(GraphDev G, CgDev C, int parity, int first, const double* __restrict__ xq = nullptr, const int32_t* __restrict__ sh_of = nullptr,
                                                               const double* __restrict__ two = nullptr) {
    if (cg_done(C)) return;
    const double delta = two ? two[0] : C.scal[12], gamma = two ? two[1] : C.scal[13];
    double beta = 0.0, den = delta;
    const bool breakdown = C.flags[1] != 0;
    if (breakdown || !(gamma > C.scal[3] * C.scal[0])) {   // converged (or broken down): the state stays that of the last completed update
        if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[0] = 1; if (!breakdown) { C.scal[1] = gamma; if (!(gamma >= -C.scal[3] * C.scal[0])) C.flags[1] = 1; } }
        return;
    }
    if (!first) {
        const double gamma_prev = C.scal[8 + 2 * (parity ^ 1)], alpha_prev = C.scal[9 + 2 * (parity ^ 1)];
        beta = gamma / gamma_prev;
        den = delta - beta * gamma / alpha_prev;
    }
    if (!(den > 0.0)) {   // not positive definite along the direction (or NaN)
        if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[1] = 1; C.flags[0] = 1; }
        return;
    }
    const double alpha = gamma / den;
    if (blockIdx.x == 0 && threadIdx.x == 0) { C.scal[8 + 2 * parity] = gamma; C.scal[9 + 2 * parity] = alpha; C.scal[1] = gamma; C.flags[2] += 1; }
    __shared__ double red[CG_BLOCK / 64];
    constexpr int KF = CG_BLOCK / 3;
    __shared__ double2 rnew[CG_BLOCK];
    __shared__ __attribute__((aligned(16))) float lfs[KF * LF_STRIDE];
    double2* __restrict__ rv = reinterpret_cast<double2*>(C.r);
    double2* __restrict__ pv = reinterpret_cast<double2*>(C.p);
    double2* __restrict__ sv = reinterpret_cast<double2*>(C.p2);    // the standard form's second direction buffer holds s = A p here
    const double2* __restrict__ wv = reinterpret_cast<const double2*>(C.q);
    double2* __restrict__ xv = reinterpret_cast<double2*>(C.x);
    double2* __restrict__ uv = reinterpret_cast<double2*>(C.z);
    const int64_t pairs = G.N * 3;
    const int64_t stride = (int64_t)gridDim.x * CG_BLOCK;
    const int64_t trips = (pairs + stride - 1) / stride;
    double acc = 0.0;
    for (int64_t it = 0; it < trips; ++it) {
        const int64_t base = it * stride + (int64_t)blockIdx.x * CG_BLOCK;
        const int64_t i = base + threadIdx.x;
        const bool live = i < pairs;
        double2 rr = make_double2(0.0, 0.0);
        if (live) {
            const double2 u = uv[i];
            double2 w = wv[i];
            if (xq) { const int32_t pos = sh_of[i / 3]; if (pos >= 0) w = reinterpret_cast<const double2*>(xq)[(size_t)pos * 3 + (i % 3)]; }
            double2 pp = pv[i], ss = sv[i], xx = xv[i];
            rr = rv[i];
            pp.x = u.x + beta * pp.x; pp.y = u.y + beta * pp.y;
            ss.x = w.x + beta * ss.x; ss.y = w.y + beta * ss.y;
            xx.x += alpha * pp.x; xx.y += alpha * pp.y;
            rr.x -= alpha * ss.x; rr.y -= alpha * ss.y;
            pv[i] = pp; sv[i] = ss; xv[i] = xx; rv[i] = rr;
        }
        __syncthreads();
        lf_stage<KF>(C.Lf, base / 3, G.N, lfs);
        rnew[threadIdx.x] = rr;
        __syncthreads();
        if (live) {
            const int j = (int)(i % 3);
            const double2 u = lf_apply_pair(lfs + (threadIdx.x / 3) * LF_STRIDE, reinterpret_cast<const double*>(rnew + (threadIdx.x - j)), j);
            uv[i] = u;
            const double wgt = G.own ? G.own[i / 3] : 1.0;
            acc += wgt * (rr.x * u.x + rr.y * u.y);
        }
    }
    const double sum = block_sum(acc, red);
    if (threadIdx.x == 0) C.part_rz[blockIdx.x] = sum;
}
// bb = b.M^-1 b (already summed over ranks at src[0]) -> scal[0]; tolerance, flags
__global__ void cgcg_scalars_init_kernel(CgDev C, const double* __restrict__ bb_src, double tol2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double bb = bb_src[0];
        C.scal[0] = bb; C.scal[1] = 0.0; C.scal[2] = 0.0; C.scal[3] = tol2;
        C.flags[0] = bb > 0.0 ? 0 : 1; C.flags[1] = bb >= 0.0 ? 0 : 1; C.flags[2] = 0;
    }
}

// grid capped at MAX_PARTIALS workgroups (grid-stride beyond).  Measured on C3 (3125 workgroups of work): the ragged 1024-workgroup
// grid (68 us / PCG iteration) beats both a balanced 782 x 4 trips (76 us) and 384-thread workgroups with 1563 partials (80 us).
static inline int cg_grid(const GraphDev& G) {
    const int64_t pairs = G.N * 3;       // cg_update: one lane per row pair -> one trip per workgroup up to MAX_PARTIALS * CG_BLOCK / 3 keyframes
    int64_t g = (pairs + CG_BLOCK - 1) / CG_BLOCK;
    if (g > CG_MAX_GRID) g = CG_MAX_GRID;
    if (g < 1) g = 1;
    return (int)g;
}
// resume a stopped PCG with a tighter tolerance: new squared tolerance, convergence flag cleared (a breakdown stays)
__global__ void cg_set_tolerance_kernel(CgDev C, double tol2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { C.scal[3] = tol2; if (!C.flags[1]) C.flags[0] = 0; }
}
void launch_cg_set_tolerance(const CgDev& C, double tol2, hipStream_t st) { hipLaunchKernelGGL(cg_set_tolerance_kernel, dim3(1), dim3(64), 0, st, C, tol2); }
// the host's view of a PCG chunk: {stopped, breakdown, iterations} and {reference norm, r.z, -} written straight into a pinned host slot (one launch at the
// floor instead of two blit copies)
__global__ void cg_poll_kernel(CgDev C, int32_t* __restrict__ hflags, double* __restrict__ hscal) {
    if (threadIdx.x < 3) { hflags[threadIdx.x] = C.flags[threadIdx.x]; hscal[threadIdx.x] = C.scal[threadIdx.x]; }
}
void launch_cg_poll(const CgDev& C, int32_t* host_flags, double* host_scal, hipStream_t st) { hipLaunchKernelGGL(cg_poll_kernel, dim3(1), dim3(64), 0, st, C, host_flags, host_scal); }

void launch_cg_init(const GraphDev& G, const CgDev& C, int warm, double tol2, hipStream_t st) {
    const int g = cg_grid(G);
    hipLaunchKernelGGL(cg_init_kernel, dim3(g), dim3(CG_BLOCK), 0, st, G, C, warm);
    hipLaunchKernelGGL(cg_scalars_init_kernel, dim3(1), dim3(256), 0, st, C, g, g, tol2);
}
void launch_cg_init_scalars(const CgDev& C, int nparts, int nparts_bb, double tol2, hipStream_t st) { hipLaunchKernelGGL(cg_scalars_init_kernel, dim3(1), dim3(256), 0, st, C, nparts, nparts_bb, tol2); }
int launch_cg_init_vectors(const GraphDev& G, const CgDev& C, int warm, hipStream_t st) {
    const int g = cg_grid(G);
    hipLaunchKernelGGL(cg_init_kernel, dim3(g), dim3(CG_BLOCK), 0, st, G, C, warm);
    return g;
}
void launch_cg_spmv(const GraphDev& G, const CgDev& C, int k, double tol2, hipStream_t st) {
    const int g = cg_grid(G);
    hipLaunchKernelGGL(cg_spmv_kernel, dim3(g), dim3(CG_BLOCK), 0, st, G, C, k & 1, k == 0 ? 1 : 0, g, tol2);
}
void launch_cg_update(const GraphDev& G, const CgDev& C, int k, int n_pq_partials, hipStream_t st) {
    const int g = cg_grid(G);
    hipLaunchKernelGGL(cg_update_kernel<false>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, k & 1, n_pq_partials, g, 0);
}
// single-reduction form: `first` = the PCG's first update (p = s = 0: also when a PCG that stopped before its first update is resumed)
void launch_cg_update_sr(const GraphDev& G, const CgDev& C, int k, int first, int n_pq_partials, hipStream_t st) {
    const int g = cg_grid(G);
    hipLaunchKernelGGL(cg_update_kernel<true>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, k & 1, n_pq_partials, g, first);
}
void launch_cgcg_dots(const GraphDev& G, const CgDev& C, hipStream_t st) { hipLaunchKernelGGL(cgcg_dots_kernel, dim3(cg_grid(G)), dim3(CG_BLOCK), 0, st, G, C); }
void launch_cg_reduce2_live(const CgDev& C, const double* pa, int na, const double* pb, int nb, double* out, hipStream_t st) {
    hipLaunchKernelGGL(cg_reduce2_live_kernel, dim3(1), dim3(256), 0, st, C, pa, na, pb, nb, out);
}
void launch_cgcg_update(const GraphDev& G, const CgDev& C, int k, int first, hipStream_t st, const double* xq, const int32_t* sh_of, const double* two) {
    hipLaunchKernelGGL(cgcg_update_kernel, dim3(cg_grid(G)), dim3(CG_BLOCK), 0, st, G, C, k & 1, first, xq, sh_of, two);
}
void launch_cgcg_scalars_init(const CgDev& C, const double* bb_src, double tol2, hipStream_t st) { hipLaunchKernelGGL(cgcg_scalars_init_kernel, dim3(1), dim3(64), 0, st, C, bb_src, tol2); }
int cg_grid_size(const GraphDev& G) { return cg_grid(G); }
void launch_apply_operator(const GraphDev& G, const CgDev& C, const double* x, double* y, hipStream_t st) { hipLaunchKernelGGL(apply_operator_kernel, dim3(cg_grid(G)), dim3(CG_BLOCK), 0, st, G, C, x, y); }

// ------------------------------------------------------------------------------------------------
// K3 (matrix-free) — the PCG matvec without an assembled matrix.  Per edge-side one compact record of 22 doubles
// (q2, q1 (x) q_o, a', dt, w|s — the switch term's r6 follows from them: pgo_device_math.hpp) instead of a 288-B block; the diagonal blocks need no storage at all
// (sum_e J_i^T J_i p_i falls out of the per-edge products).  One 128-B record per lane (C3: 420k lanes, 54 MB) instead of a 288-B block per edge side (600k x 288 B).
//   phase A  one lane per edge-side of the workgroup's keyframes: y_e = J_side^T (I - k k^T)(J1 p1 + J2 p2)  -> LDS
//   phase B  one lane per (keyframe, row): sum of the keyframe's edge-sides in list order (deterministic) + damping + regulariser
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mf_compact_kernel(GraphDev G, MfDev F, const double* __restrict__ pose8, const double* __restrict__ swv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.ninc) return;
    const uint32_t ent = F.einc[i];
    const bool is_sw = (ent >> 31) != 0;
    const int64_t e = (ent & 0x7fffffffu) >> 1;
    const EdgeClassDev& C = is_sw ? G.sw : G.rel;
    const Pose P1 = load_pose_global(pose8, C.c1[e]);
    const Pose P2 = load_pose_global(pose8, C.c2[e]);
    const double* mp = C.meas + e;
    const size_t ep = (size_t)C.Epad;
    const Meas M{mp[0], mp[ep], mp[2 * ep], mp[3 * ep], mp[4 * ep], mp[5 * ep], mp[6 * ep], mp[7 * ep]};
    const double ws = is_sw ? swv[C.swidx[e]] : M.w;
    double rec[COMPACT_DOUBLES];
    edge_compact(P1, P2, M, ws, is_sw, rec);
    for (int pl = 0; pl < MF_PLANES; ++pl) F.rec[(size_t)pl * F.ninc_pad + i] = make_double2(rec[2 * pl], rec[2 * pl + 1]);
}

// component r of B_i y_a, the pending coarse correction of keyframe i (dtheta_i = dtheta_a ; dt_i = dt_a - 2 d_i x dtheta_a); 0 for fixed keyframes
__device__ __forceinline__ double coarse_pending(const CoarseDev& K, const uint8_t* __restrict__ node_free, int64_t node, int r) {
    // every load is issued unconditionally (no load depends on another one's value): one round trip
    const double fr = node_free[node] ? 1.0 : 0.0;
    const double* y = K.yc + (size_t)(node / K.m) * 6;
    const double* d = K.d + (size_t)node * 3;
    const int c = r < 3 ? r : r - 3, c1 = c == 2 ? 0 : c + 1, c2 = c == 0 ? 2 : c - 1;
    const double yr = y[r], ya = y[c1], yb = y[c2], da = d[c1], db = d[c2];
    return fr * (r < 3 ? yr : yr - 2.0 * (da * yb - db * ya));
}

// COARSE (two-level preconditioner, fused form): the preconditioned residual is z = z_bj + P y with z_bj in C.z (block-Jacobi part, written by
// cg_update_restrict_kernel) and y = Ac^-1 P^T r in K.yc (coarse_solve_dot_kernel): the prolongation happens HERE, where z is consumed, instead
// of in a kernel of its own; `pending` = 0 right after the PCG start, when C.z is already complete.  K.m == 1: z = y alone (direct inverse).
#ifdef PGO_MF_TIMELINE
// Development aid (variant build only: scripts/dev/mf_timeline.py): thread 0 of every workgroup records the 100-MHz wall clock at the phase boundaries of the matvec — after a
// full wait for the memory operations issued so far — so that one can see where a tile's ~12 us go.  [workgroup][16] ticks; slots 0-1 kernel entry / after the
// re-reduction; per tile (first two tiles): bounds, phase 0 + barrier, record loads done, far gathers done, products + barrier, phase-B loads done, tile end.
__device__ unsigned long long pgo_mf_tl[MF_MAX_GRID * 16];
#define MF_TL(slot) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0 && (slot) < 16) pgo_mf_tl[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
extern "C" int pgo_debug_mf_timeline(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pgo_mf_tl), (size_t)n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define MF_TL(slot) do {} while (0)
#endif

template <bool FUSED, bool COARSE = false>
#ifndef PGO_MF_WAVES      // A/B aid (variant builds, scripts/dev/ab_variant.py): minimum waves per SIMD the matvec is compiled for (1 = the compiler's choice: 122 VGPRs, 4 waves)
#define PGO_MF_WAVES 1
#endif
__global__ __launch_bounds__(MF_BLOCK, PGO_MF_WAVES) void mf_spmv_kernel(GraphDev G, MfDev F, ScaleDev Sc, CgDev C, const double* __restrict__ xin, double* __restrict__ yout,
                                                           int parity, int first, int nparts, double tol2, CoarseDev K = CoarseDev{}, int pending = 0) {
    __shared__ double contrib[MF_SLOTS * 7];
    __shared__ double pwin[MF_BLOCK];
    __shared__ double red[3 * (MF_BLOCK / 64) + 1];
    const int l = threadIdx.x;
    MF_TL(0);
    double beta = 0.0;
    const double* __restrict__ pprev = FUSED ? (parity ? C.p : C.p2) : xin;
    const double* __restrict__ z = FUSED ? C.z : xin;
    // the first tile's bounds and phase-0 operands do not depend on beta: request them BEFORE the partial-sum re-reduction so that its
    // dependent L2 round trips overlap with theirs (two registers; the records stay behind the reduction — hoisting them spills)
    double v0_first = 0.0, v1_first = 0.0;
    // (the first tile's bounds too: uniform loads whose round trip — 1.2 us behind the re-reduction in the timeline build — then runs beside it)
    int64_t i0_first = 0, i1_first = 0; int32_t n0_first = 0, n1_first = 0, sw_first = 0;
    {
        const int tile = blockIdx.x;
        if (tile < F.tiles) {
            i0_first = F.tile_inc0[tile]; i1_first = F.tile_inc0[tile + 1]; sw_first = F.tile_sw0[tile];
            const int32_t n0 = F.tile_node0[tile], n1 = F.tile_node0[tile + 1];
            n0_first = n0; n1_first = n1;
            if (l < (n1 - n0) * 6) {
                const size_t vi = (size_t)n0 * 6 + l;
                v0_first = z[vi];
                if (FUSED) v1_first = pprev[vi];
                if (COARSE && pending) { const double cp = coarse_pending(K, G.node_free, (int64_t)n0 + l / 6, l % 6); v0_first = K.m == 1 ? cp : v0_first + cp; }
            }
        }
    }
    if (!FUSED && first == 3) { if (cg_done(C)) return; }      // the single-reduction PCG's matvec w = A u: nothing but "has the PCG stopped?" stands before the tiles
    if (FUSED) {
        if (first) { if (cg_done(C)) return; }
        else {
            double rz_new, rz_old;
            bool coarse_negative = false;
            bool stopped;
            if (COARSE) {
                // Two-level method: r.z = r.D^-1 r (the update kernel's partials) + rc.Ac^-1 rc (the dense solve's partials, behind them).  With the exact coarse inverse the
                // second term is >= 0, so r.z <= tol^2 b.D^-1 b implies the same for the block-Jacobi part alone.  The inverse is applied rounded to fp32, and at large
                // trust-region radii (coarse condition numbers beyond 2^24) it need not be positive definite any more: a negative coarse term can pull r.z below the
                // tolerance — or through zero — while the residual is still large.  So convergence that the block-Jacobi part alone does not confirm is a BREAKDOWN
                // (lm_step then finishes the system with plain block-Jacobi from the current iterate), never a converged step.  K.m == 1: z is the coarse term alone.
                double rz_bj, rz_c;
                stopped = block_total3_done(C.flags, C.part_rz + parity * RZ_STRIDE, nparts, C.part_rz + parity * RZ_STRIDE + nparts, C.extra_rz, C.part_rz + (parity ^ 1) * RZ_STRIDE, nparts + C.extra_rz, red, rz_bj, rz_c, rz_old);
                rz_new = rz_bj + rz_c;
                coarse_negative = K.m != 1 && !(rz_new > C.scal[3] * C.scal[0]) && rz_bj > C.scal[3] * C.scal[0];
            } else stopped = block_total2_done(C.flags, C.part_rz + parity * RZ_STRIDE, nparts + C.extra_rz, C.part_rz + (parity ^ 1) * RZ_STRIDE, nparts + C.extra_rz, red, rz_new, rz_old);
            if (stopped) return;
            const bool breakdown = C.flags[1] != 0;
            if (breakdown || !(rz_new > C.scal[3] * C.scal[0])) {
                if (blockIdx.x == 0 && threadIdx.x == 0) { C.flags[0] = 1; if (!breakdown) { C.scal[1] = rz_new; if (coarse_negative || !(rz_new >= -C.scal[3] * C.scal[0])) C.flags[1] = 1; } }
                return;
            }
            beta = rz_new / rz_old;
            if (blockIdx.x == 0 && threadIdx.x == 0) C.scal[1] = rz_new;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) C.flags[2] += 1;
    }
    MF_TL(1);
    double* __restrict__ pcur = parity ? C.p2 : C.p;
    double pq = 0.0;
#ifdef PGO_MF_TIMELINE
    int tl_base = 2;
#endif
    for (int tile = blockIdx.x; tile < F.tiles; tile += gridDim.x) {
        const bool is_first = tile == (int)blockIdx.x;
        const int64_t i0 = is_first ? i0_first : F.tile_inc0[tile], i1 = is_first ? i1_first : F.tile_inc0[tile + 1];
        const int32_t n0 = is_first ? n0_first : F.tile_node0[tile], n1 = is_first ? n1_first : F.tile_node0[tile + 1];
        const int32_t swp = is_first ? sw_first : F.tile_sw0[tile];
        const int sw0 = swp & 0xffff, pair1 = (int)((uint32_t)swp >> 16);
        const int64_t i = i0 + l;
        const int nn = n1 - n0;
        MF_TL(tl_base + 0);
        // phase 0: the tile's own keyframes' input vector p = z + beta p_prev, once, into LDS (every edge side of a keyframe needs it,
        // and odometry neighbours are inside the same window): only far endpoints of loop closures gather from global memory
        if (l < nn * 6) {
            double v;
            if (COARSE) {
                const size_t vi = (size_t)n0 * 6 + l;
                double zf = v0_first;
                if (tile != (int)blockIdx.x) {
                    zf = z[vi];
                    if (pending) { const double cp = coarse_pending(K, G.node_free, (int64_t)n0 + l / 6, l % 6); zf = K.m == 1 ? cp : zf + cp; }
                }
                v = zf + beta * (tile == (int)blockIdx.x ? v1_first : pprev[vi]);
            } else if (tile == (int)blockIdx.x) v = FUSED ? v0_first + beta * v1_first : v0_first;
            else {
                const size_t vi = (size_t)n0 * 6 + l;
                v = z[vi];
                if (FUSED) v += beta * pprev[vi];
            }
            pwin[l] = v;
        }
        __syncthreads();
        MF_TL(tl_base + 1);
        if (i < i1) {
            const uint32_t ent = F.einc[i];
            const int32_t other = F.einc_other[i];      // (requesting the first tile's index words before the head, so that its far gathers go out with the records: measured, no change)
            const uint32_t sl = F.einc_slot[i];
            const bool is_sw = l >= sw0;
            const int side = (int)(ent & 1u);
            const int ownl = (int)(sl >> 18), slot_a = (int)(sl & 511u), slot_b = (int)((sl >> 9) & 511u);
            double rec[COMPACT_DOUBLES];
#pragma unroll
            for (int pl = 0; pl < 8; ++pl) { const double2 v = F.rec[(size_t)pl * F.ninc_pad + i]; rec[2 * pl] = v.x; rec[2 * pl + 1] = v.y; }
            double kscale = 0.0;
            if (is_sw) {   // tail wavefronts of the tile only
                // r6 = [dt ; 2 (q2* (x) b).vec] is a function of the first seven planes: formed here exactly as edge_compact forms it (same expressions, same bits) instead of
                // being stored and fetched by three more 16-B loads per lane (block-Jacobi iteration 36.8 -> 35.9 us, 9.6 MB less per matvec on C3)
                {
                    const double q2c[4] = {-rec[0], -rec[1], -rec[2], rec[3]};
                    double dq[4];
                    quat_mul(q2c, rec + 4, dq);
                    rec[16] = rec[11]; rec[17] = rec[12]; rec[18] = rec[13];
                    rec[19] = 2.0 * dq[0]; rec[20] = 2.0 * dq[1]; rec[21] = 2.0 * dq[2];
                }
                kscale = sqrt(Sc.a_inv[(ent & 0x7fffffffu) >> 1]);
            } else {
#pragma unroll
                for (int k = 16; k < COMPACT_DOUBLES; ++k) rec[k] = 0.0;
            }
            MF_TL(tl_base + 2);
            double po[6], pt[6];
            {
                const double* a = pwin + ownl * 6;
#pragma unroll
                for (int c = 0; c < 6; ++c) po[c] = a[c];
            }
            if (other >= n0 && other < n1) {
                const double* b = pwin + (other - n0) * 6;
#pragma unroll
                for (int c = 0; c < 6; ++c) pt[c] = b[c];
            } else {
                const double2* b = reinterpret_cast<const double2*>(z + (size_t)other * 6);
                const double2 b0 = b[0], b1 = b[1], b2 = b[2];
                pt[0] = b0.x; pt[1] = b0.y; pt[2] = b1.x; pt[3] = b1.y; pt[4] = b2.x; pt[5] = b2.y;
                if (COARSE && pending) {
                    if (K.m == 1) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) pt[c] = 0.0;
                    }
                    {   // unconditional loads (a fixed keyframe's correction is multiplied by 0): nothing here waits for another load
                        const double fr = G.node_free[other] ? 1.0 : 0.0;
                        const double* yy = K.yc + (size_t)(other / K.m) * 6;
                        const double* dd = K.d + (size_t)other * 3;
                        const double y0 = yy[0], y1 = yy[1], y2 = yy[2], y3 = yy[3], y4 = yy[4], y5 = yy[5], d0 = dd[0], d1 = dd[1], d2 = dd[2];
                        pt[0] += fr * y0; pt[1] += fr * y1; pt[2] += fr * y2;
                        pt[3] += fr * (y3 - 2.0 * (d1 * y2 - d2 * y1));
                        pt[4] += fr * (y4 - 2.0 * (d2 * y0 - d0 * y2));
                        pt[5] += fr * (y5 - 2.0 * (d0 * y1 - d1 * y0));
                    }
                }
                if (FUSED) {
                    const double2* c = reinterpret_cast<const double2*>(pprev + (size_t)other * 6);
                    const double2 c0 = c[0], c1 = c[1], c2 = c[2];
                    pt[0] += beta * c0.x; pt[1] += beta * c0.y; pt[2] += beta * c1.x; pt[3] += beta * c1.y; pt[4] += beta * c2.x; pt[5] += beta * c2.y;
                }
            }
            MF_TL(tl_base + 3);
            double y[6];
            if (l < pair1) {       // both keyframes of the edge are in this tile: the record is read once, the shared part computed once
                double y2[6];
                compact_apply_both(rec, po, pt, 0.0, y, y2);
#pragma unroll
                for (int r = 0; r < 6; ++r) contrib[slot_b * 7 + r] = y2[r];
            } else compact_apply(rec, side, po, pt, kscale, y);
#pragma unroll
            for (int r = 0; r < 6; ++r) contrib[slot_a * 7 + r] = y[r];
        }
        __syncthreads();
        MF_TL(tl_base + 4);
        if (l < nn * 6) {
            const int nl = l / 6, r = l - nl * 6;
            const int64_t node = (int64_t)n0 + nl;
            const size_t vi = (size_t)node * 6 + r;
            const double pr = pwin[l];
            // (all four requested together: behind the free flag the slot ranges and the regulariser index were a second dependent round trip)
            const double lam = F.lam[vi];
            const uint8_t is_free = G.node_free[node];
            const ushort4 rg = F.node_rng[node];
            const int32_t pk = F.node_prior[node];
            double acc = lam * pr;
            MF_TL(tl_base + 5);
            if (is_free) {
                for (int j = rg.x; j < rg.y; ++j) acc += contrib[j * 7 + r];    // relative-pose sides, then switchable sides: the same
                for (int j = rg.z; j < rg.w; ++j) acc += contrib[j * 7 + r];    // fixed order as the incident list -> deterministic
                if (pk >= 0) {   // regulariser: J^T J p of a unary block (a handful per graph)
                    const double* Jp = G.Jp + (size_t)pk * PRIOR_DOUBLES + 6;
                    double pn[6];
                    for (int c = 0; c < 6; ++c) pn[c] = pwin[nl * 6 + c];
                    double s = 0.0;
                    for (int ii = 0; ii < 6; ++ii) { double t = 0.0; for (int c = 0; c < 6; ++c) t += Jp[ii * 6 + c] * pn[c]; s += Jp[ii * 6 + r] * t; }
                    acc += s;
                }
            }
            if (FUSED) { pcur[vi] = pr; C.q[vi] = acc; pq += acc * pr; }
            else { yout[vi] = acc; pq += acc * pr; }
        }
        __syncthreads();
        MF_TL(tl_base + 6);
#ifdef PGO_MF_TIMELINE
        tl_base += 7;
#endif
    }
    if (FUSED || first >= 2) {     // plain y = A x with first == 2 / 3: the partial sums of x.y as well (the multi-rank PCG's u.(A_r u); the one-GPU single-reduction PCG's u.w)
        const double s = block_sum(pq, red);
        if (threadIdx.x == 0) C.part_pq[blockIdx.x] = s;
    }
}

// one workgroup per tile up to MF_MAX_GRID (its p.q partials are consumed by cg_update only); beyond that an even number of trips
static inline int mf_grid(const MfDev& F) { const int g = F.tiles < MF_MAX_GRID ? F.tiles : MF_MAX_GRID; return g < 1 ? 1 : g; }
int mf_grid_size(const MfDev& F) { return mf_grid(F); }
void launch_mf_compact(const GraphDev& G, const MfDev& F, const double* pose8, const double* sw, hipStream_t st) {
    if (F.ninc > 0) hipLaunchKernelGGL(mf_compact_kernel, dim3((unsigned)((F.ninc + 255) / 256)), dim3(256), 0, st, G, F, pose8, sw);
}
void launch_mf_spmv(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, int k, double tol2, hipStream_t st) {
    const int g = mf_grid(F);
    // the partial-sum count consumed here is the one cg_init / cg_update produced (cg_grid); the one produced is mf_grid
    hipLaunchKernelGGL((mf_spmv_kernel<true, false>), dim3(g), dim3(MF_BLOCK), 0, st, G, F, Sc, C, (const double*)nullptr, (double*)nullptr, k & 1, k == 0 ? 1 : 0, cg_grid(G), tol2);
}
// two-level preconditioner, fused form (3 kernels per iteration): nparts = r.z partial slots of the update kernel (coarse_update_grid)
void launch_mf_spmv_coarse(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, int k, double tol2, int nparts, int pending, hipStream_t st) {
    hipLaunchKernelGGL((mf_spmv_kernel<true, true>), dim3(mf_grid(F)), dim3(MF_BLOCK), 0, st, G, F, Sc, C, (const double*)nullptr, (double*)nullptr, k & 1, k == 0 ? 1 : 0, nparts, tol2, K, pending);
}
// y = A x and the per-workgroup partial sums of x.y in C.part_pq[0 .. mf_grid_size): the multi-rank PCG's matvec and its u.(A_r u) in one kernel
void launch_mf_apply_dot(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const double* x, double* y, hipStream_t st) {
    hipLaunchKernelGGL((mf_spmv_kernel<false, false>), dim3(mf_grid(F)), dim3(MF_BLOCK), 0, st, G, F, Sc, C, x, y, 0, 2, 0, 0.0);
}
// the single-reduction PCG on one GPU: w = A u (u in C.z, w in C.q), partial sums of u.w in C.part_pq[0 .. mf_grid_size), nothing once the PCG has stopped
void launch_mf_apply_dot_live(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, hipStream_t st) {
    hipLaunchKernelGGL((mf_spmv_kernel<false, false>), dim3(mf_grid(F)), dim3(MF_BLOCK), 0, st, G, F, Sc, C, (const double*)C.z, C.q, 0, 3, 0, 0.0);
}
void launch_mf_apply(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const double* x, double* y, hipStream_t st) {
    hipLaunchKernelGGL((mf_spmv_kernel<false, false>), dim3(mf_grid(F)), dim3(MF_BLOCK), 0, st, G, F, Sc, C, x, y, 0, 1, 0, 0.0);
}

// ------------------------------------------------------------------------------------------------
// step finish: switch back-substitution  ds = -(gs + c1.d1 + c2.d2) a_inv  and
// model_cost_change = - sum (J d)^T (r + J d / 2)    (trust_region_minimizer.cc)
// one lane per edge, tiles as in K1
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void model_change_kernel(GraphDev G, LinDev L, ScaleDev Sc, const double* __restrict__ dp, double* __restrict__ ds_out, double* __restrict__ partials) {
    __shared__ double red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile_g = blockIdx.x * 4 + wave;
    const int ntile = G.rel.tiles + G.sw.tiles;
    double mc = 0.0;
    if (tile_g < ntile) {
        const bool is_sw = tile_g >= G.rel.tiles;
        const int tile = is_sw ? tile_g - G.rel.tiles : tile_g;
        const EdgeClassDev& C = is_sw ? G.sw : G.rel;
        const int64_t e = (int64_t)tile * TILE + lane;
        if (e < C.E) {
            const int32_t c1 = C.c1[e], c2 = C.c2[e];
            double d1[6], d2[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { d1[i] = dp[(size_t)c1 * 6 + i]; d2[i] = dp[(size_t)c2 * 6 + i]; }
            const int D = is_sw ? SW_DOUBLES : REL_DOUBLES;
            const double2* base = reinterpret_cast<const double2*>(C.J + tile_elem(D, e, 0));
            const int o1 = is_sw ? 7 : 3, o2 = is_sw ? 25 : 21;
            double u[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) u[i] = 0.0;
#pragma unroll
            for (int row = 0; row < 6; ++row) {
#pragma unroll
                for (int kp = 0; kp < 3; ++kp) {
                    const double2 a = base[(o1 + row * 3 + kp) * TILE], b = base[(o2 + row * 3 + kp) * TILE];
                    u[row] += a.x * d1[2 * kp] + a.y * d1[2 * kp + 1] + b.x * d2[2 * kp] + b.y * d2[2 * kp + 1];
                }
            }
            double r[7];
            { double2 v; v = base[0]; r[0] = v.x; r[1] = v.y; v = base[TILE]; r[2] = v.x; r[3] = v.y; v = base[2 * TILE]; r[4] = v.x; r[5] = v.y; }
            r[6] = 0.0;
            if (is_sw) {
                double Js[7];
                { double2 v; v = base[3 * TILE]; r[6] = v.x; Js[0] = v.y; v = base[4 * TILE]; Js[1] = v.x; Js[2] = v.y;
                  v = base[5 * TILE]; Js[3] = v.x; Js[4] = v.y; v = base[6 * TILE]; Js[5] = v.x; Js[6] = v.y; }
                const double* cc = L.c + (size_t)e * 12;
                double acc = L.gs[e];
#pragma unroll
                for (int i = 0; i < 6; ++i) acc += cc[i] * d1[i] + cc[6 + i] * d2[i];
                const double ds = -acc * Sc.a_inv[e];
                ds_out[e] = ds;
#pragma unroll
                for (int i = 0; i < 7; ++i) u[i] += Js[i] * ds;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) mc += u[i] * (r[i] + 0.5 * u[i]);
        }
    }
    // regularisers: handled by the last workgroup's first lanes
    if (blockIdx.x == gridDim.x - 1) {
        for (int k = threadIdx.x; k < G.n_prior; k += blockDim.x) {
            const double* base = G.Jp + (size_t)k * PRIOR_DOUBLES;
            const int32_t node = G.prior[k].node;
            for (int row = 0; row < 6; ++row) {
                double u = 0.0;
                for (int c = 0; c < 6; ++c) u += base[6 + row * 6 + c] * dp[(size_t)node * 6 + c];
                mc += u * (base[row] + 0.5 * u);
            }
        }
    }
    const double s = block_sum(mc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
void launch_model_change(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const double* delta_p, double* delta_s, double* partials, int* n_partials, hipStream_t st) {
    const int tiles = G.rel.tiles + G.sw.tiles;
    int grid = (tiles + 3) / 4;
    if (grid < 1) grid = 1;
    *n_partials = grid;
    hipLaunchKernelGGL(model_change_kernel, dim3(grid), dim3(256), 0, st, G, L, Sc, delta_p, delta_s, partials);
}

// ------------------------------------------------------------------------------------------------
// K5 — candidate point x (+) delta and the ambient step norm || x - x_cand ||^2 (Ceres' step_norm)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plus_kernel(GraphDev G, const double* __restrict__ pose8, const double* __restrict__ sw, const double* __restrict__ dp,
                                                   const double* __restrict__ ds, double* __restrict__ pose8_out, double* __restrict__ sw_out,
                                                   double* __restrict__ part_step2, double* __restrict__ part_sw_step2) {
    __shared__ double red[4];
    double s2 = 0.0, s2sw = 0.0;
    const int64_t total = G.N + G.sw.E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < G.N) {
            const double2* g = reinterpret_cast<const double2*>(pose8 + (size_t)i * 8);
            const double2 a = g[0], b = g[1], c = g[2], d = g[3];
            double q[4] = {a.x, a.y, b.x, b.y}, t[3] = {c.x, c.y, d.x}, qn[4], tn[3];
            if (G.node_free[i]) {
                const double* di = dp + (size_t)i * 6;
                quat_plus(q, di, qn);
                tn[0] = t[0] + di[3]; tn[1] = t[1] + di[4]; tn[2] = t[2] + di[5];
                double d2 = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) d2 += (q[k] - qn[k]) * (q[k] - qn[k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) d2 += (t[k] - tn[k]) * (t[k] - tn[k]);
                s2 += G.own ? G.own[i] * d2 : d2;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) qn[k] = q[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) tn[k] = t[k];
            }
            double2* o = reinterpret_cast<double2*>(pose8_out + (size_t)i * 8);
            o[0] = make_double2(qn[0], qn[1]); o[1] = make_double2(qn[2], qn[3]); o[2] = make_double2(tn[0], tn[1]); o[3] = make_double2(tn[2], 0.0);
        } else {
            const int64_t e = i - G.N;
            const int32_t si = G.sw.swidx[e];
            const double d = ds[e];
            sw_out[si] = sw[si] + d;
            s2sw += d * d;
        }
    }
    const double a = block_sum(s2, red);
    const double b = block_sum(s2sw, red);
    if (threadIdx.x == 0) { part_step2[blockIdx.x] = a; part_sw_step2[blockIdx.x] = b; }
}
// parity hook: the manifold step alone on caller-supplied keyframes (reference layout)
__global__ void manifold_plus_kernel(int64_t n, const double* __restrict__ quat, const double* __restrict__ t, const double* __restrict__ delta,
                                     double* __restrict__ quat_out, double* __restrict__ t_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double q[4] = {quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]}, qn[4];
    quat_plus(q, delta + 6 * i, qn);
#pragma unroll
    for (int k = 0; k < 4; ++k) quat_out[4 * i + k] = qn[k];
    if (t && t_out) for (int k = 0; k < 3; ++k) t_out[3 * i + k] = t[3 * i + k] + delta[6 * i + 3 + k];
}
void launch_manifold_plus(int64_t n, const double* quat, const double* t, const double* delta, double* quat_out, double* t_out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(manifold_plus_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, quat, t, delta, quat_out, t_out);
}

static inline int capped_grid(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g > MAX_PARTIALS) g = MAX_PARTIALS;
    if (g < 1) g = 1;
    return (int)g;
}
void launch_plus(const GraphDev& G, const double* pose8, const double* sw, const double* delta_p, const double* delta_s, double* pose8_out, double* sw_out,
                 double* part_step2, double* part_sw_step2, int* n_partials, hipStream_t st) {
    const int g = capped_grid(G.N + G.sw.E, 256);
    *n_partials = g;
    hipLaunchKernelGGL(plus_kernel, dim3(g), dim3(256), 0, st, G, pose8, sw, delta_p, delta_s, pose8_out, sw_out, part_step2, part_sw_step2);
}

// ||x||^2 over the free parameters (ambient: 4 + 3 per keyframe, 1 per switch) and the projected-gradient max norm
// || Plus(x, -g) - x ||_inf  (trust_region_minimizer.cc)
__global__ __launch_bounds__(256) void state_norms_kernel(GraphDev G, LinDev L, const double* __restrict__ pose8, const double* __restrict__ sw,
                                                          double* __restrict__ part_x2, double* __restrict__ part_sw_x2, double* __restrict__ part_gmax) {
    __shared__ double red[4];
    double x2 = 0.0, x2sw = 0.0, gm = 0.0;
    const int64_t total = G.N + G.sw.E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < G.N) {
            if (!G.node_free[i]) continue;
            const double2* g = reinterpret_cast<const double2*>(pose8 + (size_t)i * 8);
            const double2 a = g[0], b = g[1], c = g[2], d = g[3];
            const double q[4] = {a.x, a.y, b.x, b.y};
            const double n2 = a.x * a.x + a.y * a.y + b.x * b.x + b.y * b.y + c.x * c.x + c.y * c.y + d.x * d.x;
            x2 += G.own ? G.own[i] * n2 : n2;
            const double* gi = L.g + (size_t)i * 6;
            const double ng[3] = {-gi[0], -gi[1], -gi[2]};
            double qn[4];
            quat_plus(q, ng, qn);
#pragma unroll
            for (int k = 0; k < 4; ++k) gm = fmax(gm, fabs(qn[k] - q[k]));
#pragma unroll
            for (int k = 0; k < 3; ++k) gm = fmax(gm, fabs(gi[3 + k]));
        } else {
            const int64_t e = i - G.N;
            const double s = sw[G.sw.swidx[e]];
            x2sw += s * s;
            gm = fmax(gm, fabs(L.gs[e]));
        }
    }
    const double a = block_sum(x2, red);
    const double b = block_sum(x2sw, red);
    // max over the workgroup
    gm = wave_max(gm);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmax(m, red[i]);
        part_x2[blockIdx.x] = a; part_sw_x2[blockIdx.x] = b; part_gmax[blockIdx.x] = m;
    }
}
void launch_state_norms(const GraphDev& G, const LinDev& L, const double* pose8, const double* sw, double* part_xnorm2, double* part_sw_xnorm2, double* part_gmax,
                        int* n_partials, hipStream_t st) {
    const int g = capped_grid(G.N + G.sw.E, 256);
    *n_partials = g;
    hipLaunchKernelGGL(state_norms_kernel, dim3(g), dim3(256), 0, st, G, L, pose8, sw, part_xnorm2, part_sw_xnorm2, part_gmax);
}

__global__ void reduce_kernel(const double* __restrict__ partials, int n, int op, double* __restrict__ out) {
    __shared__ double red[4];
    double v = 0.0;
    if (op == 0) { for (int i = threadIdx.x; i < n; i += blockDim.x) v += partials[i]; v = wave_sum(v); }
    else { for (int i = threadIdx.x; i < n; i += blockDim.x) v = fmax(v, partials[i]); v = wave_max(v); }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = red[0];
        for (int i = 1; i < 4; ++i) s = op == 0 ? s + red[i] : fmax(s, red[i]);
        out[0] = s;
    }
}
void launch_reduce(const double* partials, int n, int op, double* out, hipStream_t st) { hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), 0, st, partials, n, op, out); }

// ------------------------------------------------------------------------------------------------
// layout conversion at the C-ABI boundary (reference layout: quat[4N] xyzw + t[3N], src/PoseGraphSLAM.h:153-175)
// ------------------------------------------------------------------------------------------------
__global__ void pack_pose_kernel(const double* __restrict__ quat, const double* __restrict__ t, double* __restrict__ pose8, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double2* o = reinterpret_cast<double2*>(pose8 + (size_t)i * 8);
    o[0] = make_double2(quat[4 * i], quat[4 * i + 1]); o[1] = make_double2(quat[4 * i + 2], quat[4 * i + 3]);
    o[2] = make_double2(t[3 * i], t[3 * i + 1]); o[3] = make_double2(t[3 * i + 2], 0.0);
}
__global__ void unpack_pose_kernel(const double* __restrict__ pose8, double* __restrict__ quat, double* __restrict__ t, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double* p = pose8 + (size_t)i * 8;
    quat[4 * i] = p[0]; quat[4 * i + 1] = p[1]; quat[4 * i + 2] = p[2]; quat[4 * i + 3] = p[3];
    t[3 * i] = p[4]; t[3 * i + 1] = p[5]; t[3 * i + 2] = p[6];
}
void launch_pack_pose(const double* quat, const double* t, double* pose8, int64_t N, hipStream_t st) {
    if (N > 0) hipLaunchKernelGGL(pack_pose_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, quat, t, pose8, N);
}
void launch_unpack_pose(const double* pose8, double* quat, double* t, int64_t N, hipStream_t st) {
    if (N > 0) hipLaunchKernelGGL(unpack_pose_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, pose8, quat, t, N);
}

// parity hooks: K1's tiled output -> plain row-major blocks
__global__ void unpack_k1_kernel(GraphDev G, int kind, int64_t first, int64_t count, double* r, double* J1, double* J2, double* Js) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t e = first + i;
    if (kind == 2) {
        const double* base = G.Jp + (size_t)e * PRIOR_DOUBLES;
        if (r) for (int k = 0; k < 6; ++k) r[i * 6 + k] = base[k];
        if (J1) for (int k = 0; k < 36; ++k) J1[i * 36 + k] = base[6 + k];
        return;
    }
    const bool is_sw = kind == 1;
    const EdgeClassDev& C = is_sw ? G.sw : G.rel;
    const int D = is_sw ? SW_DOUBLES : REL_DOUBLES;
    const int nr = is_sw ? 7 : 6, o1 = is_sw ? 14 : 6, o2 = is_sw ? 50 : 42;
    if (r) for (int k = 0; k < nr; ++k) r[i * nr + k] = C.J[tile_elem(D, e, k)];
    if (J1) for (int k = 0; k < 36; ++k) J1[i * 36 + k] = C.J[tile_elem(D, e, o1 + k)];
    if (J2) for (int k = 0; k < 36; ++k) J2[i * 36 + k] = C.J[tile_elem(D, e, o2 + k)];
    if (Js && is_sw) for (int k = 0; k < 7; ++k) Js[i * 7 + k] = C.J[tile_elem(D, e, 7 + k)];
}
void launch_unpack_k1(const GraphDev& G, int kind, int64_t first, int64_t count, double* r, double* J1, double* J2, double* Js, hipStream_t st) {
    if (count > 0) hipLaunchKernelGGL(unpack_k1_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, G, kind, first, count, r, J1, J2, Js);
}

// ---- K0: graph construction from the device-resident raw VIO pose array ----
// K0a: one lane per odometry edge (u = c1, u-f = c2): record = (q_obs, t_obs, 0.9^f exp(-yaw^2/6)) of u_M_umf = w_M_u^-1 w_M_umf
// (reference src/PoseGraphSLAM.cpp:1597-1606).  Reads 2 x 128 B (neighbouring lanes share poses through L1/L2), writes 64 B.
__global__ void __launch_bounds__(256) vio_odometry_kernel(int64_t n, const int32_t* __restrict__ c1, const int32_t* __restrict__ c2,
                                                           const double* __restrict__ vio, int yaw_weight, double* __restrict__ meas8) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int u = c1[e], m = c2[e];
    double Mu[16], Mm[16];
    const double2* pu = reinterpret_cast<const double2*>(vio + (size_t)u * 16);
    const double2* pm = reinterpret_cast<const double2*>(vio + (size_t)m * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const double2 a = pu[k], b = pm[k]; Mu[2 * k] = a.x; Mu[2 * k + 1] = a.y; Mm[2 * k] = b.x; Mm[2 * k + 1] = b.y; }
    double out[8];
    vio_odometry_record(Mu, Mm, u - m, yaw_weight != 0, out);
    double2* po = reinterpret_cast<double2*>(meas8 + (size_t)e * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) po[k] = make_double2(out[2 * k], out[2 * k + 1]);
}
void launch_vio_odometry(int64_t n, const int32_t* c1, const int32_t* c2, const double* vio, int yaw_weight, double* meas8, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(vio_odometry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, c1, c2, vio, yaw_weight, meas8);
}

// K0b: initial guesses (reference :1770-1778): pose[u] = left[left_of_node[u - u_begin]] * w_M_u as (xyzw, t); negative selector = skip
__global__ void __launch_bounds__(256) vio_initial_guess_kernel(int64_t u_begin, int64_t count, const double* __restrict__ left, const int32_t* __restrict__ left_of_node,
                                                                const double* __restrict__ vio, double* __restrict__ quat, double* __restrict__ t) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int sel = left_of_node[i];
    if (sel < 0) return;
    const int64_t u = u_begin + i;
    double L[16], Mu[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { L[k] = left[(size_t)sel * 16 + k]; Mu[k] = vio[(size_t)u * 16 + k]; }
    double q[4], tt[3];
    vio_left_compose(L, Mu, q, tt);
#pragma unroll
    for (int k = 0; k < 4; ++k) quat[i * 4 + k] = q[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[i * 3 + k] = tt[k];
}
void launch_vio_initial_guess(int64_t u_begin, int64_t count, const double* left, const int32_t* left_of_node, const double* vio, double* quat, double* t, hipStream_t st) {
    if (count > 0) hipLaunchKernelGGL(vio_initial_guess_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, u_begin, count, left, left_of_node, vio, quat, t);
}

// ---- several ranks: neighbour exchanges (round 6) ----
// A rank sends another rank exactly the rows that one reads and does not own (plans: pgo_mg_host.hpp).  Send and receive buffers hold one row of K = k1 + k2 doubles per listed
// node, segment by segment in peer order.  The keyframes' own exchange SUMS: a keyframe touched by several ranks holds a partial row on each; every rank adds the parts in ascending
// rank order (its own row where its rank comes), so that all of them end up with the same bits.
__global__ void gather_rows_kernel(double* __restrict__ buf, const double* __restrict__ a1, int k1, const double* __restrict__ a2, int k2, int64_t n, const int32_t* __restrict__ idx,
                                   const int32_t* __restrict__ stop) {
    const int K = k1 + k2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int64_t j = i / K; const int c = (int)(i - j * K);
    const bool stopped = stop && *stop;      // (a stopped PCG still takes part in every exchange: it sends zeros)
    const int64_t node = idx[j];
    buf[i] = stopped ? 0.0 : (c < k1 ? a1[(size_t)node * k1 + c] : a2[(size_t)node * k2 + (c - k1)]);
}
__global__ void scatter_rows_kernel(const double* __restrict__ buf, double* __restrict__ a1, int k1, double* __restrict__ a2, int k2, int64_t n, const int32_t* __restrict__ idx,
                                    const int32_t* __restrict__ stop) {
    const int K = k1 + k2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K || (stop && *stop)) return;   // a stopped PCG keeps its state bit for bit (it may be resumed with a tighter tolerance)
    const int64_t j = i / K; const int c = (int)(i - j * K);
    const int64_t node = idx[j];
    if (c < k1) a1[(size_t)node * k1 + c] = buf[i]; else a2[(size_t)node * k2 + (c - k1)] = buf[i];
}
__global__ void sum_rows_kernel(const double* __restrict__ buf, double* __restrict__ a1, int k1, double* __restrict__ a2, int k2, int64_t n_sh, const int32_t* __restrict__ sh_loc,
                                const int32_t* __restrict__ sum_ptr, const int32_t* __restrict__ sum_src, const int32_t* __restrict__ stop) {
    const int K = k1 + k2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sh * K || (stop && *stop)) return;
    const int64_t j = i / K; const int c = (int)(i - j * K);
    double* mine = c < k1 ? a1 + (size_t)sh_loc[j] * k1 + c : a2 + (size_t)sh_loc[j] * k2 + (c - k1);
    double s = 0.0;
    for (int e = sum_ptr[j]; e < sum_ptr[j + 1]; ++e) { const int32_t src = sum_src[e]; s += src < 0 ? *mine : buf[(size_t)src * K + c]; }
    *mine = s;
}
// scatter of a level's residual rows with the pre-smoothed x of the same rows formed on the spot: x = (omega D^-1) r is pointwise, and every rank holds every level's Dinv
// (the set-up has brought the halo rows' Dinv in: pgo_solver.hip, build_mg_ranks) — so only r travels (half the bytes of sending both); the same six products in the same order as the kernel that owns the row
__global__ void scatter_rows_dinv_kernel(const double* __restrict__ buf, double* __restrict__ r, double* __restrict__ x, const double* __restrict__ Dinv, int64_t n,
                                         const int32_t* __restrict__ idx, const int32_t* __restrict__ stop) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 6 || (stop && *stop)) return;
    const int64_t j = i / 6; const int c = (int)(i - j * 6);
    const int64_t node = idx[j];
    const double* rv = buf + j * 6;
    const double* Dk = Dinv + (size_t)node * 36 + c * 6;
    double xv = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) xv += Dk[k] * rv[k];
    r[(size_t)node * 6 + c] = rv[c];
    x[(size_t)node * 6 + c] = xv;
}
void launch_scatter_rows_dinv(const double* buf, double* r, double* x, const double* Dinv, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(scatter_rows_dinv_kernel, dim3((unsigned)((n * 6 + 255) / 256)), dim3(256), 0, st, buf, r, x, Dinv, n, idx, stop);
}
void launch_gather_rows(double* buf, const double* a1, int k1, const double* a2, int k2, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n * (k1 + k2) + 255) / 256)), dim3(256), 0, st, buf, a1, k1, a2, k2, n, idx, stop);
}
void launch_scatter_rows(const double* buf, double* a1, int k1, double* a2, int k2, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((n * (k1 + k2) + 255) / 256)), dim3(256), 0, st, buf, a1, k1, a2, k2, n, idx, stop);
}
void launch_sum_rows(const double* buf, double* a1, int k1, double* a2, int k2, int64_t n_sh, const int32_t* sh_loc, const int32_t* sum_ptr, const int32_t* sum_src, const int32_t* stop, hipStream_t st) {
    if (n_sh > 0) hipLaunchKernelGGL(sum_rows_kernel, dim3((unsigned)((n_sh * (k1 + k2) + 255) / 256)), dim3(256), 0, st, buf, a1, k1, a2, k2, n_sh, sh_loc, sum_ptr, sum_src, stop);
}
// in-process communicator: the ranks are handles of ONE process (threads), so a "collective" is a kernel that reads the peers' device buffers directly — on one GPU, or over
// xGMI with peer access between the GPUs of one process
__global__ void local_reduce_kernel(double* __restrict__ out, LocalPeers P, int64_t n, int op) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = P.src[0][i];
    for (int q = 1; q < P.n; ++q) { const double v = P.src[q][i]; s = op == 0 ? s + v : (v > s ? v : s); }      // rank order: the same bits on every rank
    out[i] = s;
}
__global__ void local_copy_kernel(double* __restrict__ recv, LocalPeers P) {
    const int q = blockIdx.y;
    const int64_t n = P.cnt[q];
    const double* __restrict__ src = P.src[q];
    double* __restrict__ dst = recv + P.off[q];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_local_reduce(double* out, const LocalPeers& P, int64_t n, int op, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(local_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, P, n, op);
}
void launch_local_copy(double* recv, const LocalPeers& P, hipStream_t st) {
    int64_t mx = 0;
    for (int q = 0; q < P.n; ++q) mx = std::max(mx, P.cnt[q]);
    if (mx > 0) hipLaunchKernelGGL(local_copy_kernel, dim3((unsigned)std::min<int64_t>((mx + 255) / 256, 256), (unsigned)P.n), dim3(256), 0, st, recv, P);
}
__global__ void scatter_owned_pose_kernel(const double* __restrict__ quat, const double* __restrict__ t, int64_t n, const int32_t* __restrict__ l2g,
                                          const double* __restrict__ own, double* __restrict__ gquat, double* __restrict__ gt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || own[i] == 0.0) return;
    const int64_t g = l2g[i];
    for (int k = 0; k < 4; ++k) gquat[g * 4 + k] = quat[i * 4 + k];
    for (int k = 0; k < 3; ++k) gt[g * 3 + k] = t[i * 3 + k];
}
void launch_scatter_owned_pose(const double* quat, const double* t, int64_t n, const int32_t* l2g, const double* own, double* gquat, double* gt, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(scatter_owned_pose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, quat, t, n, l2g, own, gquat, gt);
}

// ------------------------------------------------------------------------------------------------
// Two-level preconditioner: the coarse space of rigid-body modes of keyframe aggregates (CoarseDev, pgo_internal.hpp)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_bcast0(double v) { return __shfl(v, 0, 64); }

// one wavefront per aggregate: centroid of its free keyframes, then d_i = t_i - centroid for every member
__global__ __launch_bounds__(256) void coarse_geometry_kernel(GraphDev G, CoarseDev K, const double* __restrict__ pose8) {
    const int a = wave_in_grid();
    const int lane = threadIdx.x & 63;
    if (a >= K.n_agg) return;
    const int64_t i0 = (int64_t)a * K.m, i1 = i0 + K.m < G.N ? i0 + K.m : G.N;
    double sx = 0.0, sy = 0.0, sz = 0.0, cnt = 0.0;
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        if (!G.node_free[i]) continue;
        const double* t = pose8 + (size_t)i * 8 + 4;
        sx += t[0]; sy += t[1]; sz += t[2]; cnt += 1.0;
    }
    sx = wave_bcast0(wave_sum(sx)); sy = wave_bcast0(wave_sum(sy)); sz = wave_bcast0(wave_sum(sz)); cnt = wave_bcast0(wave_sum(cnt));
    const double inv = cnt > 0.0 ? 1.0 / cnt : 0.0;
    const double cx = sx * inv, cy = sy * inv, cz = sz * inv;
    if (lane == 0) { K.cen[a * 3] = cx; K.cen[a * 3 + 1] = cy; K.cen[a * 3 + 2] = cz; }
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        const double* t = pose8 + (size_t)i * 8 + 4;
        K.d[i * 3] = t[0] - cx; K.d[i * 3 + 1] = t[1] - cy; K.d[i * 3 + 2] = t[2] - cz;
    }
}
void launch_coarse_geometry(const GraphDev& G, const CoarseDev& K, const double* pose8, hipStream_t st) {
    hipLaunchKernelGGL(coarse_geometry_kernel, dim3((unsigned)((K.n_agg + 3) / 4)), dim3(256), 0, st, G, K, pose8);
}

// entry (r, c) of B_i^T H B_j with B = [[I, 0], [X, I]], X = -2 [d]x ; H (6x6 row-major) read from LDS
__device__ __forceinline__ double coarse_xentry(const double* d, int q, int c) {   // X[q][c] = -2 skew(d)[q][c]
    // skew(d) = [[0, -dz, dy], [dz, 0, -dx], [-dy, dx, 0]]
    if (q == c) return 0.0;
    const int k = 3 - q - c;                        // the remaining axis
    const double sgn = ((c - q + 3) % 3 == 1) ? -1.0 : 1.0;   // (q,c) = (0,1),(1,2),(2,0) -> -d_k ; reversed -> +d_k
    return -2.0 * sgn * d[k];
}
__device__ __forceinline__ double coarse_t(const double* H, const double* dj, int p, int c) {   // T = H B_j
    double t = H[p * 6 + c];
    if (c < 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) t += H[p * 6 + 3 + q] * coarse_xentry(dj, q, c);
    }
    return t;
}
__device__ __forceinline__ double coarse_entry(const double* H, const double* di, const double* dj, int r, int c) {
    double v = coarse_t(H, dj, r, c);
    if (r < 3) {
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) v += coarse_xentry(di, pp, r) * coarse_t(H, dj, 3 + pp, c);
    }
    return v;
}

// The value a lane owns of one contribution's fine block of the keyframe system — kind 0: the reduced diagonal block C.Dtot; an edge: J1^T J2 - c1 c2^T / a (transposed for kinds 2 / 4), from Hoff where the solver has formed it, else recomputed from K1's Jacobians.
template <bool HOFF>
__device__ __forceinline__ double fine_block_value(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, int64_t ent, int lane, int r, int c, bool own) {
    const int kind = (int)(ent & 7);
    const int64_t idx = ent >> 3;
    double h = 0.0;
    if (!own) return h;
    if (kind == 0) return C.Dtot[(size_t)idx * 36 + lane];
    const bool is_sw = kind >= 3;
    const bool transposed = kind == 2 || kind == 4;
    if (HOFF) h = L.Hoff[(size_t)((is_sw ? G.rel.Epad : 0) + idx) * 36 + (transposed ? c * 6 + r : r * 6 + c)];
    else {
        const EdgeClassDev& E = is_sw ? G.sw : G.rel;
        const int D = is_sw ? SW_DOUBLES : REL_DOUBLES;
        const int o1 = is_sw ? 14 : 6, o2 = is_sw ? 50 : 42;
        const int oa = transposed ? o2 : o1, ob = transposed ? o1 : o2;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) h += E.J[tile_elem(D, idx, oa + kk * 6 + r)] * E.J[tile_elem(D, idx, ob + kk * 6 + c)];
    }
    if (is_sw) {
        const double* cc = L.c + (size_t)idx * 12;
        h -= cc[(transposed ? 6 : 0) + r] * cc[(transposed ? 0 : 6) + c] * Sc.a_inv[idx];
    }
    return h;
}
// Ac = P^T A P: one wavefront per coarse block (a <= b), lane l < 36 owns entry (l / 6, l % 6); the block's contributions are summed in
// list order (deterministic, no atomics).  Fine blocks come from the data the LM iteration already has: the reduced diagonal blocks
// (C.Dtot: J^T J + damping - switch Schur terms + regularisers) and, per edge, J1^T J2 - c1 c2^T / a recomputed from K1's Jacobians.
// The list (a diagonal block of 8 keyframes with 5 odometry neighbours each holds ~40 entries) is taken four entries at a time, every hop of their chains
// (entry -> Jacobians; entry -> endpoint -> offset, by one lane per (contribution, component) into LDS) issued for the whole chunk before anything waits.
__global__ __launch_bounds__(256) void coarse_assemble_kernel(GraphDev G, LinDev L, ScaleDev Sc, CgDev C, CoarseDev K) {
    constexpr int CH = 4;
    __shared__ double Hs[4][CH][36];
    __shared__ double ds[4][CH][6];
    const int wv = wave_in_block(), lane = threadIdx.x & 63;
    const int blk = wave_in_grid();
    if (blk >= K.n_blk) return;                       // whole wavefronts leave together; no workgroup barrier below
    const int a = K.blk_ab[blk * 2], b = K.blk_ab[blk * 2 + 1];
    const int r = lane / 6, c = lane - r * 6;
    const bool own = lane < 36;
    const int du = lane / 6, dm = lane - du * 6;
    double acc = 0.0;
    const int64_t kend = K.blk_ptr[blk + 1];
    for (int64_t k0 = K.blk_ptr[blk]; k0 < kend; k0 += CH) {
        const int n = (int)(kend - k0 < CH ? kend - k0 : CH);
        if (du < n) {       // lanes 0 .. 6 n - 1: component dm of contribution du's pair of offsets (dm < 3: the row keyframe)
            const int64_t e = K.contrib[k0 + du];
            const int kind = (int)(e & 7);
            int64_t node = e >> 3;
            if (kind != 0) {
                const bool is_sw = kind >= 3, first = (dm < 3) != (kind == 2 || kind == 4);
                const int32_t* cp = is_sw ? (first ? G.sw.c1 : G.sw.c2) : (first ? G.rel.c1 : G.rel.c2);
                node = cp[node];
            }
            ds[wv][du][dm] = K.d[(size_t)node * 3 + (dm < 3 ? dm : dm - 3)];
        }
        double h[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) h[u] = fine_block_value<false>(G, L, Sc, C, K.contrib[k0 + u < kend ? k0 + u : kend - 1], lane, r, c, own);
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
    if (!own) return;
    if (a == b && K.agg_free[a] == 0) acc = r == c ? 1.0 : 0.0;        // an aggregate without free keyframes: identity block
    K.Ac[(size_t)(a * 6 + r) * K.nc + b * 6 + c] = acc;
    if (a != b) K.Ac[(size_t)(b * 6 + c) * K.nc + a * 6 + r] = acc;
}
__global__ void coarse_pad_identity_kernel(CoarseDev K);
void launch_coarse_assemble(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, hipStream_t st) {
    (void)hipMemsetAsync(K.Ac, 0, (size_t)K.nc * K.nc * sizeof(double), st);
    if (K.nc > 6 * K.n_agg) hipLaunchKernelGGL(coarse_pad_identity_kernel, dim3((unsigned)((K.nc - 6 * K.n_agg + 63) / 64)), dim3(64), 0, st, K);
    if (K.n_blk > 0) hipLaunchKernelGGL(coarse_assemble_kernel, dim3((unsigned)((K.n_blk + 3) / 4)), dim3(256), 0, st, G, L, Sc, C, K);
}

// the computed inverse is symmetric only up to rounding; PCG needs an exactly symmetric preconditioner: mirror one triangle
// One workgroup per 32 x 32 tile on or above the diagonal (row-major; the inversion maintains rows <= columns): the tile is read along its rows, its fp32 image written,
// and the mirror tile below the diagonal written from the LDS transpose — every global access runs along rows (the one-thread-per-element version read the
// mirror elements with stride n: 110 us at n = 2 304, this one 12.6 us).
__global__ __launch_bounds__(256) void coarse_symmetrize_kernel(CoarseDev K) {
    __shared__ double tile[32][33];
    const int n = K.nc, R = blockIdx.y, Cc = blockIdx.x;
    if (Cc < R) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 8 rows of 32 columns per pass
    for (int rr = ty; rr < 32; rr += 8) {
        const size_t g = (size_t)(R * 32 + rr) * n + Cc * 32 + tx;
        double v = K.Ac[g];
        if (R == Cc && rr > tx) v = 0.0;                         // below the diagonal of a diagonal tile: taken from the mirror element below
        tile[rr][tx] = v;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        double up = tile[rr][tx];
        if (R == Cc && rr > tx) up = tile[tx][rr];
        const size_t g = (size_t)(R * 32 + rr) * n + Cc * 32 + tx;
        if (R == Cc) K.Ac[g] = up;
        if (K.Acf) K.Acf[g] = (float)up;                        // both mirror images round the same fp64 value: the fp32 copy is exactly symmetric too
        if (R != Cc) {
            const double lo = tile[tx][rr];                     // element (row Cc*32 + rr, column R*32 + tx) = upper element (R*32 + tx, Cc*32 + rr)
            const size_t gl = (size_t)(Cc * 32 + rr) * n + R * 32 + tx;
            K.Ac[gl] = lo;
            if (K.Acf) K.Acf[gl] = (float)lo;
        }
    }
}
// one row of the fp32 inverse times the fp64 vector, lanes of one wavefront striding over it (n a multiple of 64: rows are 16-B aligned)
__device__ __forceinline__ double dense_row_dot(const float* __restrict__ Arow, const double* __restrict__ x, int n, int lane) {
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(Arow);
    const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x);
    double s = 0.0;
    const int n4 = n >> 2;
#pragma unroll 4
    for (int j = lane; j < n4; j += 64) { const float4 a = A4[j]; const double2 u = x2[2 * j], v = x2[2 * j + 1]; s += (double)a.x * u.x + (double)a.y * u.y + (double)a.z * v.x + (double)a.w * v.y; }
    return s;
}
void launch_coarse_symmetrize(const CoarseDev& K, hipStream_t st) {
    const unsigned t = (unsigned)(K.nc / 32);         // nc is a multiple of 64
    hipLaunchKernelGGL(coarse_symmetrize_kernel, dim3(t, t), dim3(256), 0, st, K);
}

// rc = P^T r: one wavefront per aggregate;  B_i^T r_i = [r_theta + 2 d_i x r_t ; r_t]
__global__ __launch_bounds__(256) void coarse_restrict_kernel(GraphDev G, CoarseDev K, const double* __restrict__ rv, const int32_t* __restrict__ stop) {
    if (stop && *stop) return;   // a stopped PCG keeps z and its partial sums (it may be resumed)
    const int a = wave_in_grid();
    const int lane = threadIdx.x & 63;
    if (a >= K.n_agg) return;
    const int64_t i0 = (int64_t)a * K.m, i1 = i0 + K.m < G.N ? i0 + K.m : G.N;
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t i = i0 + lane; i < i1; i += 64) {
        if (!G.node_free[i]) continue;
        const double* r = rv + (size_t)i * 6;
        const double* d = K.d + (size_t)i * 3;
        s[0] += r[0] + 2.0 * (d[1] * r[5] - d[2] * r[4]);
        s[1] += r[1] + 2.0 * (d[2] * r[3] - d[0] * r[5]);
        s[2] += r[2] + 2.0 * (d[0] * r[4] - d[1] * r[3]);
        s[3] += r[3]; s[4] += r[4]; s[5] += r[5];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = wave_sum(s[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) K.rc[a * 6 + k] = s[k];
    }
}
// yc = Ac^-1 rc: one wavefront per row of the dense inverse
__global__ __launch_bounds__(256) void coarse_solve_kernel(CoarseDev K, const int32_t* __restrict__ stop) {
    if (stop && *stop) return;
    const int row = wave_in_grid();
    const int lane = threadIdx.x & 63;
    if (row >= K.nc) return;
    const double s = wave_sum(dense_row_dot(K.Acf + (size_t)row * K.nc, K.rc, K.nc, lane));
    if (lane == 0) K.yc[row] = s;
}
// z_i += B_i y_a  (dtheta_i = dtheta_a ; dt_i = dt_a - 2 d_i x dtheta_a) and r.z += r.(P y): the cg_update lane / workgroup mapping, so
// that workgroup b adds its partial to the same slot b the update kernel wrote
__global__ __launch_bounds__(CG_BLOCK) void coarse_prolong_kernel(GraphDev G, CoarseDev K, const double* __restrict__ rv, double* __restrict__ zv, double* __restrict__ part_rz,
                                                                   const int32_t* __restrict__ stop) {
    __shared__ double red[CG_BLOCK / 64];
    if (stop && *stop) return;
    const int64_t pairs = G.N * 3;
    const bool direct = K.m == 1;      // one aggregate per keyframe: P = I and Ac^-1 is the inverse of the system itself, z = Ac^-1 r replaces the block-Jacobi z
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CG_BLOCK + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * CG_BLOCK) {
        const int64_t n = i / 3;
        const int j = (int)(i - n * 3);
        if (!G.node_free[n]) continue;
        const double* y = K.yc + (size_t)(n / K.m) * 6;
        const double* d = K.d + (size_t)n * 3;
        const double add[6] = {y[0], y[1], y[2], y[3] - 2.0 * (d[1] * y[2] - d[2] * y[1]), y[4] - 2.0 * (d[2] * y[0] - d[0] * y[2]), y[5] - 2.0 * (d[0] * y[1] - d[1] * y[0])};
        const double a0 = j == 0 ? add[0] : (j == 1 ? add[2] : add[4]), a1 = j == 0 ? add[1] : (j == 1 ? add[3] : add[5]);
        double2* zp = reinterpret_cast<double2*>(zv) + i;
        const double2 r = reinterpret_cast<const double2*>(rv)[i];
        double2 z = *zp;
        if (direct) { z.x = a0; z.y = a1; } else { z.x += a0; z.y += a1; }
        *zp = z;
        const double w = G.own ? G.own[n] : 1.0;
        acc += w * (r.x * a0 + r.y * a1);
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) { if (direct) part_rz[blockIdx.x] = s; else part_rz[blockIdx.x] += s; }
}
// ---- fused form of the two-level iteration (matrix-free operator, aggregates of <= 64 keyframes): THREE kernels instead of five ----
//   mf_spmv_kernel<true, true>    prolongs the pending coarse correction while it forms p (above)
//   cg_update_restrict_kernel     cg_update_kernel + rc = P^T r' : a workgroup trip covers `kft` keyframes = WHOLE aggregates (kft = (64 / m) m)
//   coarse_solve_dot_kernel       y = Ac^-1 rc and the partial sums of rc.y = r'.(P y), the coarse part of r.z, behind the update kernel's partials
// SR: the single-reduction form (cg_update_kernel<true>) of the same kernel.  The preconditioned residual it needs in full, u = z_bj + P y (y = Ac^-1 rc of the previous dense
// solve, still "pending": the classic form folds it into the next matvec's direction update), is completed here per lane from K.yc and the keyframe's offset.
template <bool SR>
__global__ __launch_bounds__(CG_BLOCK) void cg_update_restrict_kernel(GraphDev G, CgDev C, CoarseDev K, int parity, int nparts_pq, int nparts, int kft, int first, int pending) {
    __shared__ double red[3 * (CG_BLOCK / 64) + 1];
    const double2* __restrict__ rin = reinterpret_cast<const double2*>(SR ? C.r : (parity ? C.r2 : C.r));
    double2* __restrict__ rout = reinterpret_cast<double2*>(SR ? C.r : (parity ? C.r : C.r2));
    const double2* __restrict__ pcur = reinterpret_cast<const double2*>(SR ? C.p : (parity ? C.p2 : C.p));
    const double2* __restrict__ qv = reinterpret_cast<const double2*>(C.q);
    double2* __restrict__ xv = reinterpret_cast<double2*>(C.x);
    double2* __restrict__ zv = reinterpret_cast<double2*>(C.z);
    double2* __restrict__ pout = reinterpret_cast<double2*>(C.p);      // (SR only)
    double2* __restrict__ sv = reinterpret_cast<double2*>(C.p2);       // (SR only) s = A p
    double2 u0 = make_double2(0.0, 0.0), s0 = u0;
    // the lane's row pair of the pending coarse correction P y: dtheta_i = dtheta_a ; dt_i = dt_a - 2 d_i x dtheta_a (0 for fixed keyframes)
    auto pending_pair = [&](int64_t node, int j, double e0, double e1, double e2, double free_) -> double2 {
        const double* y = K.yc + (size_t)(node / K.m) * 6;
        const double y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3], y4 = y[4], y5 = y[5];
        double2 c;
        if (j == 0) c = make_double2(y0, y1);
        else if (j == 1) c = make_double2(y2, y3 - 2.0 * (e1 * y2 - e2 * y1));
        else c = make_double2(y4 - 2.0 * (e2 * y0 - e0 * y2), y5 - 2.0 * (e0 * y1 - e1 * y0));
        c.x *= free_; c.y *= free_;
        return c;
    };
    const int t = threadIdx.x;
    const int64_t groups = (G.N + kft - 1) / kft;
    const int64_t kf_first = (int64_t)blockIdx.x * kft + t / 3;
    const bool live_first = t < kft * 3 && kf_first < G.N;
    const int64_t i_first = (int64_t)blockIdx.x * kft * 3 + t;
    double2 r0 = make_double2(0.0, 0.0), q0 = r0, p0 = r0, x0 = r0;
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, fr = 0.0;      // the lane's keyframe: offset from its aggregate's centroid, free flag
    const bool direct = K.m == 1;
    auto load_u = [&](int64_t i, int64_t node, int j) {      // (SR) u = z_bj + P y (direct: the coarse term alone), s
        u0 = zv[i]; s0 = sv[i];
        if (pending) { const double2 c = pending_pair(node, j, d0, d1, d2, fr); if (direct) u0 = c; else { u0.x += c.x; u0.y += c.y; } }
    };
    if (live_first) {
        r0 = rin[i_first]; q0 = qv[i_first]; p0 = pcur[i_first]; x0 = xv[i_first];
        const double* d = K.d + (size_t)kf_first * 3; d0 = d[0]; d1 = d[1]; d2 = d[2]; fr = G.node_free[kf_first] ? 1.0 : 0.0;
        if (SR) load_u(i_first, kf_first, t % 3);
    }
    double alpha = 0.0, beta = 0.0;
    if (SR) { if (!sr_head(C, parity, first, nparts_pq, nparts, red, alpha, beta, true, direct)) return; }
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
    __shared__ double2 btr[CG_BLOCK];       // B_i^T r'_i, row pair j of keyframe t / 3 at [t]
    __shared__ double2 rnew[CG_BLOCK];
    __shared__ __attribute__((aligned(16))) float lfs[(CG_BLOCK / 3) * LF_STRIDE];
    double acc = 0.0;
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const int64_t base = g * kft;                       // first keyframe of the group: a multiple of m
        const int64_t i = base * 3 + t;
        const bool live = t < kft * 3 && base + t / 3 < G.N;
        double2 rr = make_double2(0.0, 0.0);
        if (live) {
            if (g != (int64_t)blockIdx.x) {
                r0 = rin[i]; q0 = qv[i]; p0 = pcur[i]; x0 = xv[i];
                const double* d = K.d + (size_t)(base + t / 3) * 3; d0 = d[0]; d1 = d[1]; d2 = d[2]; fr = G.node_free[base + t / 3] ? 1.0 : 0.0;
                if (SR) load_u(i, base + t / 3, t % 3);
            }
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
        __syncthreads();
        for (int f4 = t; f4 < kft * 6; f4 += CG_BLOCK) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + f4 / 6 < G.N) v = reinterpret_cast<const float4*>(C.Lf + (size_t)base * LF_STRIDE)[f4];
            reinterpret_cast<float4*>(lfs)[f4] = v;
        }
        rnew[t] = rr;
        __syncthreads();
        if (live) {
            const int j = t % 3;
            const double2 z = lf_apply_pair(lfs + (t / 3) * LF_STRIDE, reinterpret_cast<const double*>(rnew + (t - j)), j);
            zv[i] = z;
            if (!direct) acc += rr.x * z.x + rr.y * z.y;
        }
        // rc = P^T r' for the aggregates of this group: B_i^T r'_i = [r_theta + 2 d_i x r_t ; r_t] per keyframe (its three lanes, operands from LDS),
        // then a fixed-order sum over each aggregate's keyframes (deterministic)
        {
            double2 b = make_double2(0.0, 0.0);
            if (live) {
                const int j = t % 3;
                const double* r6 = reinterpret_cast<const double*>(rnew + (t - j));
                if (j == 0) b = make_double2(r6[0] + 2.0 * (d1 * r6[5] - d2 * r6[4]), r6[1] + 2.0 * (d2 * r6[3] - d0 * r6[5]));
                else if (j == 1) b = make_double2(r6[2] + 2.0 * (d0 * r6[4] - d1 * r6[3]), r6[3]);
                else b = make_double2(r6[4], r6[5]);
                b.x *= fr; b.y *= fr;
            }
            btr[t] = b;
        }
        __syncthreads();
        const int64_t left = G.N - base < kft ? G.N - base : kft;
        const int n_ag = (int)((left + K.m - 1) / K.m);
        for (int u = t; u < n_ag * 6; u += CG_BLOCK) {
            const int la = u / 6, c = u - la * 6;
            const double* col = reinterpret_cast<const double*>(btr) + (size_t)la * K.m * 6 + c;
            const int cnt = (int)(left - (int64_t)la * K.m < K.m ? left - (int64_t)la * K.m : K.m);
            double sum = 0.0;
            for (int j = 0; j < cnt; ++j) sum += col[j * 6];
            K.rc[(size_t)(base / K.m + la) * 6 + c] = sum;
        }
        __syncthreads();
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) C.part_rz[(parity ^ 1) * RZ_STRIDE + blockIdx.x] = s;
}
constexpr int CSOLVE_ROWS = 8;      // rows of the dense inverse per workgroup (one wavefront each)
__global__ __launch_bounds__(CSOLVE_ROWS * 64) void coarse_solve_dot_kernel(CoarseDev K, const int32_t* __restrict__ stop, double* __restrict__ part) {
    __shared__ double ws[CSOLVE_ROWS];
    if (stop && *stop) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * CSOLVE_ROWS + wv;
    double s = 0.0, rc_row = 0.0;
    if (row < K.nc) {
        rc_row = K.rc[row];
        s = wave_sum(dense_row_dot(K.Acf + (size_t)row * K.nc, K.rc, K.nc, lane));
        if (lane == 0) K.yc[row] = s;
    }
    if (lane == 0) ws[wv] = s * rc_row;
    __syncthreads();
    if (threadIdx.x == 0) { double tsum = 0.0; for (int i = 0; i < CSOLVE_ROWS; ++i) tsum += ws[i]; part[blockIdx.x] = tsum; }
}
int coarse_group_keyframes(const CoarseDev& K) { return K.m >= 1 && K.m <= CG_BLOCK / 3 ? (CG_BLOCK / 3 / K.m) * K.m : 0; }   // 0: aggregates too large for the fused form
int coarse_update_grid(const GraphDev& G, const CoarseDev& K) {
    const int kft = coarse_group_keyframes(K);
    int64_t g = kft > 0 ? (G.N + kft - 1) / kft : 1;
    if (g > MAX_PARTIALS) g = MAX_PARTIALS;
    return (int)(g < 1 ? 1 : g);
}
int coarse_solve_grid(const CoarseDev& K) { return (K.nc + CSOLVE_ROWS - 1) / CSOLVE_ROWS; }
void launch_cg_update_restrict(const GraphDev& G, const CgDev& C, const CoarseDev& K, int k, int n_pq_partials, hipStream_t st) {
    const int g = coarse_update_grid(G, K);
    hipLaunchKernelGGL(cg_update_restrict_kernel<false>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, K, k & 1, n_pq_partials, g, coarse_group_keyframes(K), 0, 0);
}
// single-reduction form of the two-level method's fused iteration: w = A (z_bj + P y) with the partials of u.w (stops with the PCG), then the update that re-reduces both dot products
void launch_mf_apply_dot_live_coarse(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, int pending, hipStream_t st) {
    hipLaunchKernelGGL((mf_spmv_kernel<false, true>), dim3(mf_grid(F)), dim3(MF_BLOCK), 0, st, G, F, Sc, C, (const double*)C.z, C.q, 0, 3, 0, 0.0, K, pending);
}
void launch_cg_update_restrict_sr(const GraphDev& G, const CgDev& C, const CoarseDev& K, int k, int first, int pending, int n_pq_partials, hipStream_t st) {
    const int g = coarse_update_grid(G, K);
    hipLaunchKernelGGL(cg_update_restrict_kernel<true>, dim3(g), dim3(CG_BLOCK), 0, st, G, C, K, k & 1, n_pq_partials, g, coarse_group_keyframes(K), first, pending);
}
void launch_coarse_solve_dot(const CoarseDev& K, const int32_t* stop, double* part, hipStream_t st) {
    hipLaunchKernelGGL(coarse_solve_dot_kernel, dim3(coarse_solve_grid(K)), dim3(CSOLVE_ROWS * 64), 0, st, K, stop, part);
}
void launch_coarse_apply(const GraphDev& G, const CgDev& C, const CoarseDev& K, const double* r, double* z, double* part_rz, bool inside_iteration, hipStream_t st) {
    const int32_t* stop = inside_iteration ? C.flags : nullptr;     // at PCG start the flag still belongs to the previous solve
    hipLaunchKernelGGL(coarse_restrict_kernel, dim3((unsigned)((K.n_agg + 3) / 4)), dim3(256), 0, st, G, K, r, stop);
    hipLaunchKernelGGL(coarse_solve_kernel, dim3((unsigned)((K.nc + 3) / 4)), dim3(256), 0, st, K, stop);
    hipLaunchKernelGGL(coarse_prolong_kernel, dim3(cg_grid(G)), dim3(CG_BLOCK), 0, st, G, K, r, z, part_rz, stop);
}

// ------------------------------------------------------------------------------------------------
// Dense inverse of the (symmetric positive definite) coarse operator, in place: blocked Gauss-Jordan without pivoting, block size 32,
// working on the (row-major) UPPER triangle only.  After the blocks before k0 have been swept the full Gauss-Jordan matrix M satisfies
// M_ij = M_ji when i and j are on the same side of k0 and M_ij = -M_ji otherwise, so with
//     u_x = the old coupling of index x to the pivot block (row x of the column panel for x < k0, column x of the row panel for x > k),
//     v_x = P u_x,  P = A_kk^-1,  s_x = -1 for x < k0 and +1 beyond the pivot block,
// one step is:  panels  A[x][k] = -v_x (x < k0),  A[k][x] = v_x (x > k),  A_kk = P ;  trailing update  A_ij -= u_i . (s_j v_j)  for i <= j.
// Two launches per block: `gj_panels` (V^T = P U^T on the fp64 MFMA, one wavefront per 16 indices; writes the new panels and U^T, s V^T in
// k-major scratch) and `gj_update` (64 x 64 tile per workgroup on or above the diagonal, 32 x 32 per wavefront as 2 x 2 MFMA 16x16x4 tiles
// over K = 32, operands straight from the L2-resident scratch).  The workgroup that owns the NEXT pivot block inverts it in LDS right after
// updating it, so the pivot inverse never sits on the critical path.  Rows and columns of the pivot block carry u = v = 0 and so pass
// through the update unchanged.  n is a multiple of 64 (padded with an identity block).  n^3 flops on half the matrix per step; the matrix
// (<= 75 MB) streams from the Infinity Cache.  Written here instead of calling rocSOLVER: the first rocBLAS handle of a process costs
// seconds to minutes of library loading on a cold box.
// v_mfma_f64_16x16x4_f64 operand maps: A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], D[row = (l >> 4) + 4 reg][col = l & 15].
// ------------------------------------------------------------------------------------------------
constexpr int GJ_NB = 32;
typedef double gj_d4 __attribute__((ext_vector_type(4)));

// 32 x 32 Gauss-Jordan of the symmetric block in `a` (LDS), called by all 256 threads of a workgroup; writes the symmetrised inverse to
// Pinv.  The 32 pivots are a sequential chain, so the elimination runs in the registers of ONE wavefront without barriers: lane (br, bc) of
// an 8 x 8 grid owns the 4 x 4 block (4 br.., 4 bc..); per pivot it fetches its 4 entries of the pivot row and of the pivot column with
// cross-lane shuffles (8 instead of the 17 LDS reads + 2 barriers of a workgroup-wide version: 15 -> 5 us).
__device__ inline void gj_block_inverse(double (*a)[GJ_NB + 1], double* __restrict__ Pinv, int32_t* __restrict__ fail) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int br = tid >> 3, bc = tid & 7;
        double x[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = a[4 * br + i][4 * bc + j];
        bool bad = false;
#pragma unroll
        for (int p = 0; p < GJ_NB; ++p) {
            const int pb = p >> 2, pi = p & 3;
            const double piv = __shfl(x[pi][pi], pb * 9);
            double apc[4], arp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) apc[j] = __shfl(x[pi][j], pb * 8 + bc);     // pivot row, my columns
#pragma unroll
            for (int i = 0; i < 4; ++i) arp[i] = __shfl(x[i][pi], br * 8 + pb);     // pivot column, my rows
            bad |= !(piv > 0.0);                                                     // not positive definite (or NaN)
            const double inv = 1.0 / piv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool prow = br == pb && i == pi;
                const double f = arp[i] * inv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool pcol = bc == pb && j == pi;
                    x[i][j] = prow ? (pcol ? inv : apc[j] * inv) : (pcol ? -f : x[i][j] - f * apc[j]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[4 * br + i][4 * bc + j] = x[i][j];
        if (tid == 0 && bad) *fail = 1;
    }
    __syncthreads();
    const int r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) Pinv[r * GJ_NB + c0 + q] = 0.5 * (a[r][c0 + q] + a[c0 + q][r]);
}

// the first pivot block (the later ones are inverted by gj_update_kernel one step ahead)
__global__ __launch_bounds__(256) void gj_pivot_kernel(const double* __restrict__ A, int n, int k0, double* __restrict__ Pinv, int32_t* __restrict__ fail) {
    __shared__ double a[GJ_NB][GJ_NB + 1];
    const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = c0 + q, lo = r < c ? r : c, hi = r < c ? c : r;
        a[r][c] = A[(size_t)(k0 + lo) * n + k0 + hi];
    }
    __syncthreads();
    gj_block_inverse(a, Pinv, fail);
}

// one wavefront per 16 consecutive indices x
__global__ __launch_bounds__(256) void gj_panels_kernel(double* __restrict__ A, int n, int k0, const double* __restrict__ Pinv, double* __restrict__ UT, double* __restrict__ VT) {
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int x0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (x0 >= n) return;
    if (x0 >= k0 && x0 < k0 + GJ_NB) {           // pivot rows: the block becomes P; zero coupling keeps them out of the trailing update
#pragma unroll
        for (int kk = 0; kk < GJ_NB / 4; ++kk) {
            const int k = kk * 4 + lk;
            UT[(size_t)k * n + x0 + lr] = 0.0; VT[(size_t)k * n + x0 + lr] = 0.0;
            A[(size_t)(x0 + lr) * n + k0 + k] = Pinv[(x0 - k0 + lr) * GJ_NB + k];
        }
        return;
    }
    const bool before = x0 < k0;
    double u[GJ_NB / 4];
#pragma unroll
    for (int kk = 0; kk < GJ_NB / 4; ++kk) {
        const int k = kk * 4 + lk;
        u[kk] = before ? A[(size_t)(x0 + lr) * n + k0 + k] : A[(size_t)(k0 + k) * n + x0 + lr];
    }
    gj_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < GJ_NB / 4; ++kk) {
        const int k = kk * 4 + lk;
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pinv[lr * GJ_NB + k], u[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pinv[(16 + lr) * GJ_NB + k], u[kk], acc1, 0, 0, 0);
    }
    const double sgn = before ? -1.0 : 1.0;
#pragma unroll
    for (int kk = 0; kk < GJ_NB / 4; ++kk) UT[(size_t)(kk * 4 + lk) * n + x0 + lr] = u[kk];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = h * 16 + lk + 4 * reg;                       // D row -> index inside the pivot block, D column lr -> x
            const double w = sgn * (h == 0 ? acc0[reg] : acc1[reg]);
            VT[(size_t)c * n + x0 + lr] = w;
            if (before) A[(size_t)(x0 + lr) * n + k0 + c] = w;
            else A[(size_t)(k0 + c) * n + x0 + lr] = w;
        }
}

// trailing update of the tiles on or above the diagonal; the owner of the next pivot block also inverts it (into Pinv)
__global__ __launch_bounds__(256) void gj_update_kernel(double* __restrict__ A, int n, int k0, const double* __restrict__ UT, const double* __restrict__ VT, double* __restrict__ Pinv, int32_t* __restrict__ fail) {
    __shared__ double a[GJ_NB][GJ_NB + 1];
    const int k1 = k0 + GJ_NB, kt = k1 < n ? k1 / 64 : 0;
    int bi = blockIdx.y, bj = blockIdx.x;
    if (bi == 0 && bj == 0) bi = bj = kt;                             // the tile of the next pivot block is dispatched first: its
    else if (bi == kt && bj == kt) bi = bj = 0;                       // in-LDS inversion overlaps with the other tiles
    if (bj < bi) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int i0 = bi * 64 + wr * 32, j0 = bj * 64 + wc * 32;
    const bool ahead = k1 < n && bi == bj && bi == kt;                // this workgroup owns the next pivot block
    if (!(bi == bj && wr == 1 && wc == 0)) {                          // (that quadrant lies below the diagonal)
        gj_d4 acc[2][2];
        double old[2][2][4];                                          // the tile's current values: all 16 loads in flight before the first store
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[s][t] = gj_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) old[s][t][reg] = A[(size_t)(i0 + s * 16 + lk + 4 * reg) * n + j0 + t * 16 + lr];
            }
#pragma unroll
        for (int kk = 0; kk < GJ_NB / 4; ++kk) {
            const size_t row = (size_t)(kk * 4 + lk) * n;
            const double a0 = UT[row + i0 + lr], a1 = UT[row + i0 + 16 + lr], b0 = VT[row + j0 + lr], b1 = VT[row + j0 + 16 + lr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        const bool mine = ahead && i0 == k1 && j0 == k1;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int i = i0 + s * 16 + lk + 4 * reg, j = j0 + t * 16 + lr;
                    const double v = old[s][t][reg] - acc[s][t][reg];
                    A[(size_t)i * n + j] = v;
                    if (mine) a[i - k1][j - k1] = v;
                }
    }
    if (!ahead) return;
    __syncthreads();
    {   // only the upper triangle of the block is maintained: mirror it
        const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = c0 + q; m[q] = r <= c ? a[r][c] : a[c][r]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) a[r][c0 + q] = m[q];
        __syncthreads();
    }
    gj_block_inverse(a, Pinv, fail);
}

// ONE launch per block step (default; -DPGO_GJ_TWO_LAUNCHES keeps the panels + update pair).  What the panels kernel did is done by the update's own wavefronts:
//   * the raw couplings u_x of ALL indices to this step's pivot block are in scratch already — UT, written by the PREVIOUS step's launch from the values it had just updated
//     (two scratch copies alternate, as do two copies of the pivot inverse: every workgroup reads this step's while the owner of the next pivot block writes the next one's);
//   * a wavefront forms s_j v_j = s_j P u_j for the 32 columns of its quadrant with the same eight MFMA per 16 columns the panels kernel issued — the D layout of
//     v_mfma_f64_16x16x4 (row lk + 4 reg, column lr) IS the B layout the trailing update wants for k = reg + 4 h, so v goes from accumulator to operand without leaving registers;
//   * quadrants on the pivot block's columns (rows before it) store -v_i instead of an update, quadrants on its rows (columns beyond it) store v_j, the pivot quadrant stores P;
//   * quadrants on the NEXT pivot block's columns / rows also store their final values, transposed where needed, as the next step's UT.
// Same MFMA sequences on the same operands as the two-launch form: the inverse is bit for bit the same.
__global__ __launch_bounds__(256) void gj_first_panel_kernel(const double* __restrict__ A, int n, int ldu, double* __restrict__ UT) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
#pragma unroll 4
    for (int k = 0; k < GJ_NB; ++k) UT[(size_t)k * ldu + x] = x < GJ_NB ? 0.0 : A[(size_t)k * n + x];
}
__global__ __launch_bounds__(256) void gj_step_kernel(double* __restrict__ A, int n, int ldu, int k0, const double* __restrict__ UT, double* __restrict__ UTn,
                                                      const double* __restrict__ Pin, double* __restrict__ Pout, int32_t* __restrict__ fail) {
    __shared__ double a[GJ_NB][GJ_NB + 1];
    const int k1 = k0 + GJ_NB, kt = k1 < n ? k1 / 64 : 0;
    int bi = blockIdx.y, bj = blockIdx.x;
    if (bi == 0 && bj == 0) bi = bj = kt;                             // the tile of the next pivot block is dispatched first: its
    else if (bi == kt && bj == kt) bi = bj = 0;                       // in-LDS inversion overlaps with the other tiles
    if (bj < bi) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int i0 = bi * 64 + wr * 32, j0 = bj * 64 + wc * 32;
    const bool ahead = k1 < n && bi == bj && bi == kt;                // this workgroup owns the next pivot block
    if (!(bi == bj && wr == 1 && wc == 0)) {                          // (that quadrant lies below the diagonal)
        const bool rows_pivot = i0 == k0, cols_pivot = j0 == k0;      // quadrants are 32-aligned, as is the pivot block
        const bool next_cols = k1 < n && j0 == k1 && i0 < k1;         // final values = next step's couplings of the rows before the next pivot block
        const bool next_rows = k1 < n && i0 == k1 && j0 > k1;         // ... of the columns beyond it
        if (rows_pivot && cols_pivot) {                               // the pivot block becomes P (full block)
#pragma unroll
            for (int kk = 0; kk < GJ_NB / 4; ++kk) {
                const int k = kk * 4 + lk;
                A[(size_t)(k0 + lr) * n + k0 + k] = Pin[lr * GJ_NB + k];
                A[(size_t)(k0 + 16 + lr) * n + k0 + k] = Pin[(16 + lr) * GJ_NB + k];
            }
        } else if (cols_pivot) {                                      // rows before the pivot block: A[x][k] = -v_x, v_x = P u_x
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                gj_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                double u[GJ_NB / 4];
#pragma unroll
                for (int kk = 0; kk < GJ_NB / 4; ++kk) u[kk] = UT[(size_t)(kk * 4 + lk) * ldu + i0 + s * 16 + lr];
#pragma unroll
                for (int kk = 0; kk < GJ_NB / 4; ++kk) {
                    const int k = kk * 4 + lk;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pin[lr * GJ_NB + k], u[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pin[(16 + lr) * GJ_NB + k], u[kk], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int c = h * 16 + lk + 4 * reg;
                        A[(size_t)(i0 + s * 16 + lr) * n + k0 + c] = -1.0 * (h == 0 ? acc0[reg] : acc1[reg]);
                    }
            }
        } else {
            // s_j v_j of this quadrant's columns: vj[t][h][reg] = row (h 16 + lk + 4 reg), column (j0 + 16 t + lr)
            gj_d4 vj[2][2];
            const double sgn = j0 < k0 ? -1.0 : 1.0;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                gj_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                double u[GJ_NB / 4];
#pragma unroll
                for (int kk = 0; kk < GJ_NB / 4; ++kk) u[kk] = UT[(size_t)(kk * 4 + lk) * ldu + j0 + t * 16 + lr];
#pragma unroll
                for (int kk = 0; kk < GJ_NB / 4; ++kk) {
                    const int k = kk * 4 + lk;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pin[lr * GJ_NB + k], u[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pin[(16 + lr) * GJ_NB + k], u[kk], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) { vj[t][0][reg] = sgn * acc0[reg]; vj[t][1][reg] = sgn * acc1[reg]; }
            }
            if (rows_pivot) {                                         // columns beyond the pivot block: A[k][x] = v_x
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int c = h * 16 + lk + 4 * reg;
                            const double w = vj[t][h][reg];
                            A[(size_t)(k0 + c) * n + j0 + t * 16 + lr] = w;
                            if (next_cols) UTn[(size_t)(t * 16 + lr) * ldu + k0 + c] = w;      // (j0 == k1: these rows' couplings to the next pivot block)
                        }
            } else {
                gj_d4 acc[2][2];
                double old[2][2][4];                                  // the tile's current values: all 16 loads in flight before the first store
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[s][t] = gj_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) old[s][t][reg] = A[(size_t)(i0 + s * 16 + lk + 4 * reg) * n + j0 + t * 16 + lr];
                    }
#pragma unroll
                for (int kk = 0; kk < GJ_NB / 4; ++kk) {
                    const size_t row = (size_t)(kk * 4 + lk) * ldu;
                    const double a0 = UT[row + i0 + lr], a1 = UT[row + i0 + 16 + lr];
                    const double b0 = vj[0][kk >> 2][kk & 3], b1 = vj[1][kk >> 2][kk & 3];
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
                }
                const bool mine = ahead && i0 == k1 && j0 == k1;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int i = i0 + s * 16 + lk + 4 * reg, j = j0 + t * 16 + lr;
                            const double v = old[s][t][reg] - acc[s][t][reg];
                            A[(size_t)i * n + j] = v;
                            if (mine) a[i - k1][j - k1] = v;
                            if (next_cols) UTn[(size_t)(j - k1) * ldu + i] = v;
                            if (next_rows) UTn[(size_t)(i - k1) * ldu + j] = v;
                        }
            }
        }
    }
    if (!ahead) return;
    __syncthreads();
    {   // only the upper triangle of the block is maintained: mirror it
        const int r = threadIdx.x >> 3, c0 = (threadIdx.x & 7) * 4;
        double m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = c0 + q; m[q] = r <= c ? a[r][c] : a[c][r]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) a[r][c0 + q] = m[q];
        __syncthreads();
    }
    gj_block_inverse(a, Pout, fail);
}

__global__ void coarse_pad_identity_kernel(CoarseDev K) {    // rows/columns beyond 6 n_agg: a decoupled identity block
    const int i = 6 * K.n_agg + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K.nc) K.Ac[(size_t)i * K.nc + i] = 1.0;
}

__global__ void coarse_shift_kernel(CoarseDev K, double eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K.nc) K.Ac[(size_t)i * K.nc + i] *= 1.0 + eps;
}
void launch_coarse_shift(const CoarseDev& K, double eps, hipStream_t st) { hipLaunchKernelGGL(coarse_shift_kernel, dim3((unsigned)((K.nc + 255) / 256)), dim3(256), 0, st, K, eps); }

// debug aid: the dense inverse (and its fp32 image) with the sign flipped — the coarse part of the preconditioner becomes negative definite, r.z goes negative
// within an iteration or two, and the PCG must report a breakdown (tests/test_gpu_breakdown_retry.py drives the block-Jacobi retry of lm_step through it)
__global__ void coarse_negate_kernel(CoarseDev K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)K.nc * K.nc) { K.Ac[i] = -K.Ac[i]; if (K.Acf) K.Acf[i] = -K.Acf[i]; }
}
void launch_coarse_negate(const CoarseDev& K, hipStream_t st) {
    const size_t n = (size_t)K.nc * K.nc;
    if (n) hipLaunchKernelGGL(coarse_negate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, K);
}

// Ac (assembled, padded) -> Ac^-1, exactly symmetric; *fail != 0 when a pivot was not positive
void launch_coarse_invert(const CoarseDev& K, double* scratch /* 64 nc + 4096 doubles */, int32_t* fail, hipStream_t st) {
    const int n = K.nc;
    double* UT = scratch;
    double* VT = scratch + (size_t)n * GJ_NB;
    double* Pinv = scratch + (size_t)(n + 16) * GJ_NB * 2;
    hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, st, K.Ac, n, 0, Pinv, fail);
    // One launch per block step where the launches are latency-bound; from n = 2 560 on every tile row's recomputation of P u_j costs more than the panels launch it
    // saves (measured, ms, one launch vs two: n 384 0.181 / 0.206, 1 024 0.449 / 0.530, 1 216 0.55 / 0.66, 1 792 0.93 / 1.02, 2 304 1.31 / 1.42, 2 560 1.74 / 1.72, 3 072 2.84 / 2.27,
    // 4 608 7.3 / 5.6; n = 2 048 — the multigrid's dense level on C3 — is the exception below 2 560: 1.28 / 1.15, with or without the padded scratch rows).  Same bits either way.
#ifdef PGO_GJ_TWO_LAUNCHES
    const bool fused = false;
#else
    const bool fused = n <= 2304 && n != 2048;
#endif
    if (!fused) {
        for (int k0 = 0; k0 < n; k0 += GJ_NB) {
            hipLaunchKernelGGL(gj_panels_kernel, dim3((unsigned)((n / 16 + 3) / 4)), dim3(256), 0, st, K.Ac, n, k0, Pinv, UT, VT);
            hipLaunchKernelGGL(gj_update_kernel, dim3((unsigned)(n / 64), (unsigned)(n / 64)), dim3(256), 0, st, K.Ac, n, k0, UT, VT, Pinv, fail);
        }
    } else {
        const int ldu = n + 16;                    // (rows of the scratch copies 128 B off a multiple of n: the transposed stores of a power-of-two n would all hit one channel)
        double* U0 = scratch; double* U1 = scratch + (size_t)ldu * GJ_NB;
        double* Pinv2 = Pinv + GJ_NB * GJ_NB;      // (the scratch ends with 2 x 1024 doubles behind 64 (n + 16))
        hipLaunchKernelGGL(gj_first_panel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)K.Ac, n, ldu, U0);
        int step = 0;
        for (int k0 = 0; k0 < n; k0 += GJ_NB, ++step)
            hipLaunchKernelGGL(gj_step_kernel, dim3((unsigned)(n / 64), (unsigned)(n / 64)), dim3(256), 0, st, K.Ac, n, ldu, k0, (const double*)((step & 1) ? U1 : U0), (step & 1) ? U0 : U1,
                               (const double*)((step & 1) ? Pinv2 : Pinv), (step & 1) ? Pinv : Pinv2, fail);
    }
    launch_coarse_symmetrize(K, st);     // lower <- upper
}

#include "pgo_mg_kernels.hpp"

}  // namespace pgo
