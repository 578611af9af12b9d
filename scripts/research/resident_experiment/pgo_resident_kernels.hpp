// pgo_resident_kernels.hpp — the PCG of SESSION-SIZED graphs (the reference's own workload: hundreds to a few thousand keyframes) as ONE
// resident kernel per chunk of iterations.  Included at the end of pgo_kernels.hip (namespace pgo).
//
// Why: at these sizes a PCG iteration is two dependent kernel boundaries (~12 us) around ~1 us of work.  A grid-wide barrier over all 8 XCDs is
// no cheaper (measured, scripts/microbench/grid_barrier.hip: 12 us for 256 workgroups, the agent-scope release/acquire writes back and
// invalidates L2s), but INSIDE one XCD the 4 MiB L2 is the single point of coherence: 32 workgroups exchange data through it with agent-scope
// relaxed loads/stores (they bypass the per-CU vector L1) in 1.5 us per round, no cache maintenance at all.  So:
//   * the launch has 8 x 32 workgroups of 1024 lanes; those that landed on XCD 0 (HW_REG_XCC_ID) take a slot, the others leave at once;
//   * a participant owns whole matvec tiles (MfDev: <= 256 edge sides, <= 42 keyframes), four per pass (one per 256-lane quarter), and
//     runs BOTH halves of the iteration on them: q = A p for the tile's keyframes, then x, r, z = M^-1 r for the same keyframes;
//   * the two dot products of an iteration are the two barriers: every participant publishes its partial into a slot of an all-ones-initialised
//     row, and polls the row (one 64-lane load) until no slot is empty — arrival and reduction are the same round trip; every participant
//     sums the slots in the same order, so all take the same branch and the result is bitwise reproducible for a given participant count;
//   * every vector another participant may read (z, p) or that changes from iteration to iteration goes through L2 (ldc / stc below);
//     records, indices and factors are immutable during a PCG and use ordinary cached loads.
// The memory conventions at entry and exit (r/r2 and p/p2 ping-pong by iteration parity, r.z partials per parity, flags, scalars) are those of
// mf_spmv_kernel / cg_update_kernel, so chunks of both kinds can follow each other (start-up, resume with a tighter tolerance, final test).

constexpr int RES_QUARTERS = 4;
constexpr int RES_BLOCK = MF_BLOCK * RES_QUARTERS;
constexpr int RES_GRID = 256;                    // 8 XCDs x 32: one workgroup per CU of the chosen XCD
constexpr int RES_KF = MF_BLOCK / 6;             // keyframes per tile (42)
#define PGO_XCC_ID_REG ((3 << 11) | (0 << 6) | 20)   // s_getreg_b32 HW_REG_XCC_ID, bits [3:0]

__device__ __forceinline__ double ldc(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One value per participant in, the sum over participants (slot order, identical everywhere) out.  `row` = this exchange's slots, `other` = the
// slots of the same exchange one iteration earlier: every participant has read them before it arrived here, so the owner may empty its own.
__device__ __forceinline__ double res_exchange(unsigned long long* row, unsigned long long* other, int slot, int n, double mine, double* sh) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this wavefront's vector stores have reached L2
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) {
            unsigned long long b = (unsigned long long)__double_as_longlong(mine);
            if (b == RES_EMPTY) b = 0x7ff8000000000000ull;
            __hip_atomic_store(row + slot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned long long v = 0;
        for (;;) {
            if (lane < n) v = __hip_atomic_load(row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(v == RES_EMPTY) == 0ull) break;
        }
        if (lane == 0) __hip_atomic_store(other + slot, RES_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double d = lane < n ? __longlong_as_double((long long)v) : 0.0;
        d = wave_sum(d);
        if (lane == 0) *sh = d;
    }
    __syncthreads();
    return *sh;
}

__global__ __launch_bounds__(RES_BLOCK) void pcg_resident_kernel(GraphDev G, MfDev F, ScaleDev Sc, CgDev C, ResDev R, int k0, int len, int nparts_in) {
    __shared__ double contrib[RES_QUARTERS][MF_BLOCK * 7];
    __shared__ double pwin[RES_QUARTERS][MF_BLOCK];
    __shared__ __attribute__((aligned(16))) float lfs[RES_QUARTERS][RES_KF * LF_STRIDE];
    __shared__ double red[2 * (RES_BLOCK / 64)];
    __shared__ double sh_x;
    __shared__ int sh_slot, sh_n;
    // ---- who takes part ----
    const bool on_xcd = (int)(__builtin_amdgcn_s_getreg(PGO_XCC_ID_REG) & 15) == 0;
    if (threadIdx.x == 0) {
        int slot = -1;
        if (on_xcd) slot = (int)__hip_atomic_fetch_add(R.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(R.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int n = 0;
        if (slot >= 0 && slot < RES_MAX_PART) {
            while (__hip_atomic_load(R.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
            n = (int)__hip_atomic_load(R.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n > RES_MAX_PART) n = RES_MAX_PART;
        } else slot = -1;
        sh_slot = slot; sh_n = n;
    }
    __syncthreads();
    const int slot = sh_slot, n = sh_n;
    if (slot < 0) return;
    const int qd = threadIdx.x >> 8, l = threadIdx.x & (MF_BLOCK - 1);
    const int per_pass = n * RES_QUARTERS;
    const int npass = (F.tiles + per_pass - 1) / per_pass;
    const bool lead = slot == 0 && threadIdx.x == 0;
    // ---- entry state: exactly what mf_spmv_kernel of iteration k0 would see ----
    if (C.flags[0] != 0) return;
    bool breakdown = C.flags[1] != 0;
    const double tol_bb = C.scal[3] * C.scal[0];
    double rz_new, rz_old;
    block_total2(C.part_rz + (k0 & 1) * RZ_STRIDE, nparts_in, C.part_rz + ((k0 & 1) ^ 1) * RZ_STRIDE, nparts_in, red, rz_new, rz_old);
    int k = k0, done_its = 0;
#ifdef PGO_RES_TIMING
    unsigned long long ts[8][10];
#define RES_T(j) if (threadIdx.x == 0 && k >= k0 + 20 && k < k0 + 28) ts[k - k0 - 20][j] = wall_clock64();
#else
#define RES_T(j)
#endif
    bool finished = false;
    for (; k < k0 + len; ++k) {
        const int parity = k & 1;
        double beta = 0.0;
        if (k != 0) {
            if (breakdown || !(rz_new > tol_bb)) { finished = true; break; }
            beta = rz_new / rz_old;
        }
        ++done_its;
        const double* pprev = parity ? C.p : C.p2;
        double* pcur = parity ? C.p2 : C.p;
        const double* rin = parity ? C.r2 : C.r;
        double* rout = parity ? C.r : C.r2;
        RES_T(0)
        // ---- phase A: p = z + beta p_prev, q = A p on my tiles ----
        double pq = 0.0;
        for (int pass = 0; pass < npass; ++pass) {
            const int tile = (pass * n + slot) * RES_QUARTERS + qd;
            const bool have = tile < F.tiles;
            int64_t i0 = 0, i1 = 0; int32_t n0 = 0, n1 = 0; int sw0 = 0;
            if (have) { i0 = F.tile_inc0[tile]; i1 = F.tile_inc0[tile + 1]; n0 = F.tile_node0[tile]; n1 = F.tile_node0[tile + 1]; sw0 = F.tile_sw0[tile]; }
            const int nn = n1 - n0;
            const int64_t i = i0 + l;
            if (l < nn * 6) {
                const size_t vi = (size_t)n0 * 6 + l;
                pwin[qd][l] = ldc(C.z + vi) + beta * ldc(pprev + vi);
            }
            __syncthreads();
            RES_T(5)
            if (have && i < i1) {
                const uint32_t ent = F.einc[i];
                const bool is_sw = l >= sw0;
                const int side = (int)(ent & 1u);
                const int32_t other = F.einc_other[i];
                const int ownl = F.einc_ownl[i];
                double rec[COMPACT_DOUBLES];
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) { const double2 v = F.rec[(size_t)pl * F.ninc_pad + i]; rec[2 * pl] = v.x; rec[2 * pl + 1] = v.y; }
                double kscale = 0.0;
                if (is_sw) {
#pragma unroll
                    for (int pl = 8; pl < MF_PLANES; ++pl) { const double2 v = F.rec[(size_t)pl * F.ninc_pad + i]; rec[2 * pl] = v.x; rec[2 * pl + 1] = v.y; }
                    kscale = sqrt(Sc.a_inv[(ent & 0x7fffffffu) >> 1]);
                } else {
#pragma unroll
                    for (int kk = 16; kk < COMPACT_DOUBLES; ++kk) rec[kk] = 0.0;
                }
                double po[6], pt[6];
                {
                    const double* a = pwin[qd] + ownl * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) po[c] = a[c];
                }
                if (other >= n0 && other < n1) {
                    const double* b = pwin[qd] + (other - n0) * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) pt[c] = b[c];
                } else {
                    const double* zb = C.z + (size_t)other * 6;
                    const double* pb = pprev + (size_t)other * 6;
                    double zz[6], pp[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) { zz[c] = ldc(zb + c); pp[c] = ldc(pb + c); }
#pragma unroll
                    for (int c = 0; c < 6; ++c) pt[c] = zz[c] + beta * pp[c];
                }
                double y[6];
                compact_apply(rec, side, po, pt, kscale, y);
#pragma unroll
                for (int r = 0; r < 6; ++r) contrib[qd][l * 7 + r] = y[r];
            }
            __syncthreads();
            RES_T(6)
            if (l < nn * 6) {
                const int nl = l / 6, r = l - nl * 6;
                const int64_t node = (int64_t)n0 + nl;
                const size_t vi = (size_t)node * 6 + r;
                const double pr = pwin[qd][l];
                double acc = F.lam[vi] * pr;
                if (G.node_free[node]) {
                    const ushort4 rg = F.node_rng[node];
                    for (int j = rg.x; j < rg.y; ++j) acc += contrib[qd][j * 7 + r];
                    for (int j = rg.z; j < rg.w; ++j) acc += contrib[qd][j * 7 + r];
                    const int32_t pk = F.node_prior[node];
                    if (pk >= 0) {
                        const double* Jp = G.Jp + (size_t)pk * PRIOR_DOUBLES + 6;
                        double pn[6];
                        for (int c = 0; c < 6; ++c) pn[c] = pwin[qd][nl * 6 + c];
                        double s = 0.0;
                        for (int ii = 0; ii < 6; ++ii) { double t = 0.0; for (int c = 0; c < 6; ++c) t += Jp[ii * 6 + c] * pn[c]; s += Jp[ii * 6 + r] * t; }
                        acc += s;
                    }
                }
                stc(pcur + vi, pr); stc(C.q + vi, acc);
                pq += acc * pr;
            }
            __syncthreads();
        }
        RES_T(1)
        const double pq_part = block_sum(pq, red);
        const double pq_tot = res_exchange(R.slots + (0 * 2 + parity) * RES_MAX_PART, R.slots + (0 * 2 + (parity ^ 1)) * RES_MAX_PART, slot, n, pq_part, &sh_x);
        RES_T(2)
        if (!(pq_tot > 0.0)) {   // breakdown: x is left untouched; the next convergence test stops the PCG (as cg_update_kernel does)
            breakdown = true;
            rz_old = rz_new; rz_new = 0.0;
            // keep the exchange pattern: the r.z exchange of this iteration still takes place so that the slot rows stay in step
            const double dummy = res_exchange(R.slots + (1 * 2 + parity) * RES_MAX_PART, R.slots + (1 * 2 + (parity ^ 1)) * RES_MAX_PART, slot, n, 0.0, &sh_x);
            (void)dummy;
            continue;
        }
        const double alpha = rz_new / pq_tot;
        // ---- phase B: x += alpha p, r' = r - alpha q, z = M^-1 r' on the same keyframes ----
        double acc_rz = 0.0;
        for (int pass = 0; pass < npass; ++pass) {
            const int tile = (pass * n + slot) * RES_QUARTERS + qd;
            const bool have = tile < F.tiles;
            int32_t n0 = 0, n1 = 0;
            if (have) { n0 = F.tile_node0[tile]; n1 = F.tile_node0[tile + 1]; }
            const int nn = n1 - n0;
            const size_t vi = (size_t)n0 * 6 + l;
            double rr = 0.0;
            if (l < nn * 6) {
                rr = ldc(rin + vi) - alpha * ldc(C.q + vi);
                const double xx = ldc(C.x + vi) + alpha * ldc(pcur + vi);
                stc(rout + vi, rr); stc(C.x + vi, xx);
                reinterpret_cast<float4*>(lfs[qd])[l] = reinterpret_cast<const float4*>(C.Lf + (size_t)n0 * LF_STRIDE)[l];
            }
            contrib[qd][l] = rr;
            __syncthreads();
            if (l < nn * 6) {
                const int nl = l / 6, r = l - nl * 6;
                const double z = lf_apply_row(lfs[qd] + nl * LF_STRIDE, contrib[qd] + (l - r), r);
                stc(C.z + vi, z);
                acc_rz += rr * z;
            }
            __syncthreads();
        }
        RES_T(3)
        const double rz_part = block_sum(acc_rz, red);
        const double rz_next = res_exchange(R.slots + (1 * 2 + parity) * RES_MAX_PART, R.slots + (1 * 2 + (parity ^ 1)) * RES_MAX_PART, slot, n, rz_part, &sh_x);
        rz_old = rz_new; rz_new = rz_next;
        RES_T(4)
    }
#ifdef PGO_RES_TIMING
    if (threadIdx.x == 0 && (slot == 0 || slot == 1 || slot == n - 1) && k >= k0 + 28) for (int j = 0; j < 4; ++j)
        printf("slot %d/%d it %d: [A: window %llu, records+apply %llu, rows %llu] phaseA %llu  sum+exchange %llu  phaseB %llu  sum+exchange %llu (x10 ns)\n", slot, n, j, ts[j][5] - ts[j][0], ts[j][6] - ts[j][5], ts[j][1] - ts[j][6], ts[j][1] - ts[j][0], ts[j][2] - ts[j][1], ts[j][3] - ts[j][2], ts[j][4] - ts[j][3]);
#endif
    // ---- exit state for whichever kernel continues (or for the host) ----
    if (lead) {
        if (finished) { C.flags[0] = 1; if (!breakdown) C.scal[1] = rz_new; }
        else if (k > 0 && !breakdown) C.scal[1] = rz_new;
        if (breakdown) C.flags[1] = 1;
        C.flags[2] += done_its;
    }
    if (slot == 0) {
        double* a = C.part_rz + (k & 1) * RZ_STRIDE;
        double* b = C.part_rz + ((k & 1) ^ 1) * RZ_STRIDE;
        for (int i = threadIdx.x; i < nparts_in; i += blockDim.x) { a[i] = i == 0 ? rz_new : 0.0; b[i] = i == 0 ? rz_old : 0.0; }
    }
}

__global__ void resident_reset_kernel(ResDev R) {
    const int t = threadIdx.x;
    if (t < 2) R.ctl[t] = 0u;
    for (int i = t; i < 4 * RES_MAX_PART; i += blockDim.x) R.slots[i] = RES_EMPTY;
}
void launch_pcg_resident(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const ResDev& R, int k0, int len, hipStream_t st) {
    hipLaunchKernelGGL(resident_reset_kernel, dim3(1), dim3(256), 0, st, R);
    hipLaunchKernelGGL(pcg_resident_kernel, dim3(RES_GRID), dim3(RES_BLOCK), 0, st, G, F, Sc, C, R, k0, len, cg_grid(G));
}
