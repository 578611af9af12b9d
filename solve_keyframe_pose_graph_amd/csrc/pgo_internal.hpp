// pgo_internal.hpp — device data layout and kernel launch interface shared by pgo_kernels.hip (device code)
// and pgo_solver.hip (host LM controller + C-ABI).  gfx950 only.
//
// HBM layout (all fp64 unless noted; "tile" = 64 consecutive edges of one class = one wavefront of K1):
//   pose8   [N][8]            qx qy qz qw tx ty tz pad — one 64-B record per keyframe (coalesced 16-B/lane loads,
//                             LDS-staged per wavefront in K1).  The C-ABI keeps the reference's quat[4N]/t[3N] arrays.
//   sw      [S]               switch variables
//   edge inputs per class (relpose / switchable), SoA planes padded to a multiple of 64:
//     c1,c2 [Epad] int32      endpoint keyframes           meas [8][Epad]  q_obs(4) t_obs(3) weight
//     swidx [Epad] int32      (switchable only)            win  [tiles] int4 {lo1,n1,lo2,n2} pose windows for LDS staging
//   K1 output per tile, AoSoA [tile][k/2][lane][2] so every store is 16 B/lane, 1 KiB/wave-instruction:
//     relpose   78 doubles/edge: r[6]  J1[36] J2[36]                (row-major 6x6, cols [dtheta, dt])
//     switch    86 doubles/edge: r[7]  Js[7]  J1[36] J2[36]         (J1,J2 rows 0..5; row 6 is identically 0)
//   K2 outputs: Hd [N][36] (sum J^T J, incl. regularisers), g [N][6], Hoff [slot][36] = J1^T J2,
//               c [Es][12] = [J1^T Js ; J2^T Js], hss [Es], gs [Es]
//   LM system: BSR of the Schur-reduced damped normal matrix, one block row per keyframe:
//     bsr_rowptr [N+1], bsr_col [nnzb] int32 (static per graph), bsr_val [nnzb][36] (rebuilt per LM iteration)
//     Lf [N][24] block-Jacobi preconditioner (packed fp32 Cholesky factors), b [N][6] right-hand side, CG vectors [N][6]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pgo_device_math.hpp"

namespace pgo {

constexpr int TILE = 64;
constexpr int REL_DOUBLES = 78;    // r6 + J1 + J2
constexpr int SW_DOUBLES = 86;     // r7 + Js7 + J1 + J2
constexpr int K1_WAVES = 4;        // wavefronts (tiles) per K1 workgroup
constexpr int WIN_MAX = 72;        // poses per LDS window: 46 KB of LDS per workgroup -> 3 workgroups/CU (measured 34.3 us; 96 poses / 2 per CU: 36.0 us)
constexpr int WIN_STRIDE = 80;     // bytes per staged pose record (64 B + 16 B pad: conflict-free ds_read_b128)
constexpr int MAX_PARTIALS = 1024; // grid cap for kernels that emit per-block partial sums
constexpr int CG_MAX_GRID = 1024;   // cap on the PCG vector kernels' workgroups (= their r.z partial slots); 2048 (one trip per workgroup on C3) measured 46.5 vs 41.9 us per iteration
constexpr int RZ_STRIDE = CG_MAX_GRID + MAX_PARTIALS;   // one parity of the r.z partials: the update kernel's slots, then up to MAX_PARTIALS slots of the multigrid's fine prolongation / the two-level solve
constexpr int PQ_SLOTS = CG_MAX_GRID;   // p.q partial slots: the matvec's workgroups (MF_MAX_GRID matrix-free, the vector kernels' grid for block-CSR)
constexpr int PRIOR_DOUBLES = 42;  // r6 + J1
constexpr int MF_BLOCK = 256;      // lanes (edge sides) per workgroup tile of the matrix-free operator (measured per PCG iteration on C3:
                                   // 128 -> 49.2 us, 256 -> 42.8 us, 512 -> 44.4 us, 1024 -> 51.7 us)
                                   // with one lane per in-tile edge (~4.2 lanes per keyframe on C3 instead of 6): 192 lanes -> 32.4 us per matvec, 256 -> 27.3 us
constexpr int MF_MAX_NODES = 42;   // keyframes per tile (42 * 6 rows <= 256 lanes in the row phase; tiles of 60 keyframes with a second row pass measured slower: 43.5 vs 41.9 us per iteration on C3)
constexpr int MF_SLOTS = 384;      // edge SIDES per tile (LDS contribution slots): a lane that serves both sides of an in-tile edge fills two
constexpr int MF_MAX_GRID = 1024;  // cap on matvec workgroups = p.q partial sums = what 256 CUs hold at 4 workgroups each (measured per matvec on C3 / C4 with the round-2 lanes:
                                   // 512: 34.1 / 57.2 us, 768: 29.2 / 49.4, 896: 28.9 / 49.5, 1024: 27.3 / 45.3, 1536: 30.4 / 52.1, 2560: 34.6 / 51.8)
constexpr int MF_PLANES = 8;       // stored double2 planes of the compact record: q2 b a' dt (w|s, -); r6 (rec[16..21]) is recomputed by the matvec

struct PriorDev {        // NodePoseRegularization target (rigid), see prior_residual()
    double Rf[9];
    double tf[3];
    double qf[4];
    double w;
    int32_t node;
    int32_t pad_;
};

struct EdgeClassDev {
    const int32_t* c1;
    const int32_t* c2;
    const double* meas;     // 8 planes x Epad
    const int32_t* swidx;   // switchable only
    const int4* win;        // per tile
    double* J;              // tiles x DOUBLES x 64
    int64_t E, Epad;
    int32_t tiles;
};

struct GraphDev {
    int64_t N, S;
    EdgeClassDev rel, sw;
    const PriorDev* prior; int32_t n_prior; double* Jp;   // Jp [n_prior][42]
    // node -> incident (slot<<1 | side) CSR.  slot: relpose e -> e ; switch j -> rel.Epad + j ; prior k -> rel.Epad + sw.Epad + k
    const int64_t* inc_rowptr; const int64_t* inc;  // inc entries are int64: (slot << 1) | side
    const uint8_t* node_free;
    // multi-GPU (rank-local subgraph): 1.0 where this rank is the keyframe's owner (lowest rank touching it), else 0.0; nullptr on one GPU.
    // Owner-weighted sums count every shared keyframe once across ranks; the owner alone adds the all-reduced Hd, g and the damping.
    const double* own;
    // BSR structure
    const int64_t* bsr_rowptr; const int32_t* bsr_col; int64_t nnzb;
};

// Matrix-free operator (PGO_LINEAR_PCG_MATRIX_FREE): one compact record per edge-side in keyframe-major ("incident") order,
// stored as 11 double2 planes [plane][ninc_pad] so a workgroup tile reads 512 consecutive records with 1-KiB wave loads.
struct MfDev {
    // per LANE, tile-major.  A tile is a run of whole keyframes; its lanes are, in this order:
    //   pairs   relative-pose edges with BOTH keyframes in the tile: one lane reads the record once and produces both sides' contributions
    //   sides   relative-pose edge sides whose other keyframe lies outside the tile (its vector rows are gathered from global memory)
    //   sides   switchable edge sides (loop closures: the other keyframe is almost never in the tile) — so only the tail wavefronts touch r6
    // Contributions go to per-SIDE slots in LDS, ordered as the keyframes' incident lists (relative-pose sides by keyframe, then switchable
    // sides by keyframe): the per-keyframe sums run over the same terms in the same order whichever lane produced them.
    const uint32_t* einc;        // [ninc]  bit31 = switchable, bits 30..1 = edge index inside its class, bit0 = side of the OWN keyframe (pairs: 0)
    const int32_t* einc_other;   // [ninc]  the other endpoint
    const uint32_t* einc_slot;   // [ninc]  bits 0-8 slot of the own side, 9-17 slot of the other side (511: none), 18-23 own keyframe, tile-local
    const int64_t* tile_inc0;    // [tiles+1] first edge side of each workgroup tile (whole keyframes per tile, <= MF_BLOCK sides)
    const int32_t* tile_sw0;     // [tiles]   bits 0-15 tile-local index of the first switchable lane, bits 16-31 number of pair lanes
    const int32_t* tile_node0;   // [tiles+1]
    const ushort4* node_rng;     // [N] tile-local SLOT ranges {rel_begin, rel_end, sw_begin, sw_end} of the keyframe's sides
    const int32_t* node_prior;   // [N] regulariser index or -1
    double2* rec;                // [MF_PLANES][ninc_pad]: planes 0-6 q2 b a' dt, plane 7 (w|s, -)
    double* lam;                 // [N][6] LM damping in the unscaled space (identity rows for fixed keyframes)
    int64_t ninc, ninc_pad;
    int32_t tiles;
};

struct LinDev {            // per-linearisation products
    double* Hd; double* g; double* Hoff; double* c; double* hss; double* gs;
};

struct ScaleDev {
    double* scale_p;  // [N][6]
    double* scale_s;  // [Es]  (per switchable edge)
    double* diag_p;   // [N][6] clamped squared column norms of the scaled Jacobian
    double* diag_s;   // [Es]
    double* a_inv;    // [Es]  1 / (hss + lambda_s)
};

// Two-level preconditioner (large trust regions): block-Jacobi PLUS one coarse space — the rigid-body modes of `n_agg` aggregates of
// consecutive keyframes, z = D^-1 r + P Ac^-1 P^T r with Ac = P^T A P formed and inverted densely once per LM iteration.  Keyframe i
// of aggregate a moves with the aggregate: dtheta_i = dtheta_a, dt_i = dt_a - 2 [d_i]x dtheta_a, d_i = t_i - centroid_a.
struct CoarseDev {
    int32_t n_agg, nc;            // aggregates, coarse unknowns (6 per aggregate)
    int32_t m;                    // keyframes per aggregate: agg(i) = i / m
    int32_t n_blk;                // coarse 6x6 blocks (a <= b) that receive contributions
    double* cen;                  // [n_agg][3] centroid of the FREE keyframes of the aggregate
    double* d;                    // [N][3]     t_i - centroid
    double* Ac;                   // [nc][nc]   P^T A P, then its inverse (symmetric, full storage)
    double* rc;                   // [nc]       P^T r
    double* yc;                   // [nc]       Ac^-1 P^T r
    const int64_t* blk_ptr;       // [n_blk+1]  contribution list of each coarse block
    const int32_t* blk_ab;        // [n_blk][2] (a, b), a <= b
    const int64_t* contrib;       // (index << 3) | kind : 0 keyframe diagonal block, 1/2 relative-pose edge forward/transposed, 3/4 switchable edge
    const int32_t* agg_free;      // [n_agg] free keyframes in the aggregate (0: identity block)
    float* Acf;                   // [nc][nc]   the inverse rounded to fp32 (what the per-iteration solves stream: half the bytes; a preconditioner needs no more); null: not kept
};

// Aggregation multigrid (large graphs): z = D^-1 r + P V(P^T r), V = one V(1,1) cycle (block-Jacobi smoothing) over coarse levels 1..n_sparse
// of ever larger rigid aggregates (pgo_mg_host.hpp), the level above them solved densely.  All coarse levels are block-CSR with 6x6 blocks in
// the column-pair-major layout of the block-CSR PCG (element (row, col) at (row/2)*12 + col*2 + (row&1)), the diagonal block first in its row.
constexpr int MG_MAX_LEVELS = 12;
constexpr int MG_TILE_ROWS = 32;         // rows per workgroup tile of the level kernels (192 lanes = 32 rows x 6)
struct MgLevelDev {
    int32_t n, n_next, tiles, n_ps;                              // n_ps: blocks of Ps (smoothed transition to the level above; else 0)
    int64_t nnzb;
    const int64_t* rowptr; const int32_t* col; double* val;      // block-CSR
    const int64_t* g_ptr; const int64_t* g_ent;                  // Galerkin contribution lists of the blocks
    double* Dinv;                                                // [n][36] row-major: omega x inverse of the diagonal block
    double* pos; double* d;                                      // [n][3] position (centroid of the aggregate); offset to the parent's
    const int32_t* parent; const int32_t* agg_ptr; const int4* tile_info; const int2* tile_rows;   // tile_rows [tile][MG_TILE_ROWS]: block range of each row of the tile (no dependent tile_info -> rowptr load);   // nodes of level l+1: members contiguous; per workgroup tile {first aggregate, end aggregate, first row, end row}
    double* r; double* x; double* xt; double* xf;                // [n][6] restricted residual, pre-smoothed x, x + P x_next, final x
    float* valf;                                                 // the blocks rounded to fp32, exactly symmetric: what the cycle streams
    // smoothed transition to the level above (pgo_mg_host.hpp): Ps = (I - c Dinv A) P and W = A Ps as explicit 6x6 blocks (row-major), needed only to FORM the
    // level above (Ps^T W); inside the cycle Ps is applied implicitly — one more row product before the restriction and one after the prolongation
    int32_t smoothed, n_w;                                       // n_w: blocks of W = A Ps
    int32_t seg_shift, pad3_;                                    // log2 of the lane groups sharing a block row in the level kernels (tiles then hold <= 32 >> seg_shift rows)
    const int32_t* ps_rowptr; const int32_t* ps_col; const int32_t* w_rowptr; const int32_t* w_col; const int64_t* psT_ptr; const int64_t* psT_ent;
    double* ps_val; double* w_val;
    double* t; double* u; double* y; const double* zero;         // [n][6] residual after pre-smoothing, c Dinv t, the smoothed correction; a vector of zeros
    double* dlump;                                               // keyframe level, filtered smoothed transition (round 6): [n][36] the diagonal blocks with the dropped blocks lumped in; else null
    // set-up kernels run one wavefront per block: the block's row and the slot of its transposed block come from tables instead of a binary search over rowptr
    // (14 dependent loads at the head of every wavefront of level 1)
    const int32_t* row_of; const int32_t* tr_of;                 // [nnzb] row of block k; slot of block (col, row) (k itself on the diagonal or when absent)
    const int32_t* ps_row; const int32_t* w_row;                 // [n_ps], [n_w] rows of the blocks of Ps and W
    // explicit transfer operator of the smoothed transition (pgo_mg_host.hpp; null / 0 when off: the cycle then applies Ps implicitly, four row products on this level):
    // R^T = Ps - Dinv W as fp32 blocks on W's pattern (w_rowptr / w_col; per tile of THIS level the block range of each row: rt_rows), R by coarse row (rT_col = fine row,
    // r_valf = the transposed fp32 blocks — the same rounded numbers; tiles of (MG_TILE_ROWS >> rT_seg_shift) consecutive coarse rows, block ranges in rT_rows)
    float* rt_valf; float* r_valf;
    const int2* rt_rows; const int2* rT_rows; const int32_t* rT_col; const int32_t* rT_of_w; const int32_t* ps_of_w;
    int32_t rT_tiles, rT_seg_shift;
    // several ranks, distributed cycle (round 6): the CYCLE's kernels run on this rank's share of the level — its tiles [tile0, tile0 + tiles_own) and, for the restriction half of
    // mg_sdown_kernel, the coarse rows [rT_row0, rT_row1) (rT_rows / rT_tiles describe THAT range) — from vectors whose other entries the exchanges have brought in.  One GPU, and
    // levels every rank runs completely: tile0 = 0, tiles_own = tiles, rT_row0 = 0, rT_row1 = n_next.
    int32_t tile0, tiles_own, rT_row0, rT_row1;
    // several ranks, distributed SET-UP (round 6): the set-up kernels of a distributed level work on this rank's rows [su_row0, su_row1) — its blocks [su_blk0, su_blk1), its blocks
    // of Ps [su_ps0, su_ps1) and of W [su_w0, su_w1) — and the product Ps^T W on the blocks of the level above its rows contribute to (su_prod, ascending; null: every block, the
    // whole sum); what they read of other ranks' rows the block exchanges of pgo_solver.hip bring in.  One GPU, and levels every rank sets up completely: the whole level.
    int32_t su_row0, su_row1, su_ps0, su_ps1, su_w0, su_w1, n_su_prod, pad4_;
    int64_t su_blk0, su_blk1;
    const int32_t* su_prod;
};
struct MgDev {
    int32_t n_levels;                    // levels 1..n_levels; the last one is dense (CoarseDev: Ac, rc = its residual, yc = its solution)
    int32_t n1;                          // nodes of level 1
    const int32_t* agg0;                 // [N] level-1 node of each keyframe (-1: not part of the system)
    const int32_t* mem0_ptr; const int32_t* mem0;   // level-1 node -> keyframes
    double* d0;                          // [N][3] t_i - pos_1[agg0[i]]
    // restriction inside the PCG's vector update: per run of MG_BLOCK0 consecutive keyframes (one workgroup trip of cg_update) the level-1 aggregates it holds,
    // MG_BLOCK0 slots of {aggregate or -1, 8 member keyframes as run-local bytes (0xff: none)} — aggregates never cross a run boundary (pgo_mg_host.hpp)
    const int4* blk_tab;                 // [runs][MG_BLOCK0] {aggregate, members 0-3, members 4-7, unused}; null: restriction by its own kernel
    // several ranks (edge sharding): the keyframe arrays above are the rank's LOCAL keyframes, the level-1 ids GLOBAL (the hierarchy is the same on every rank)
    const double* inv_cnt;               // [n1] 1 / (members of the level-1 node over all ranks); null on one GPU
    int32_t a0, a1;                      // the level-1 aggregates this rank restricts to (its own: all their keyframes are local); one GPU: [0, n1)
    const int32_t* g0_slots; int32_t n_g0, pad_;      // several ranks, distributed set-up: the level-1 blocks this rank's edges and owned keyframes contribute to (ascending); null: every block
};
constexpr int MG_BLOCK0 = 64;

struct CgDev {
    double* val; float* Lf; double* Dtot; double* b;   // Lf [N][24]: packed fp32 Cholesky factor of the block-Jacobi blocks
    double* x; double* r; double* r2; double* z; double* p; double* p2; double* q;   // r/r2 and p/p2 ping-pong by iteration parity
    double* part_pq;      // [PQ_SLOTS]
    double* part_rz;      // [2][RZ_STRIDE]
    int32_t extra_rz;     // r.z partial slots that follow the update kernel's (written by the multigrid's level-1 kernel: the fine prolongation is fused into it)
    int32_t pad_;
    double* scal;         // [0]=||b||^2_{M^-1} [1]=last r.z [2]=unused [3]=squared relative tolerance
    int32_t* flags;       // [0]=done [1]=breakdown [2]=iterations
};

// ---- launchers (pgo_kernels.hip).  All asynchronous on `st`. ----
void launch_k1(const GraphDev& G, const double* pose8, const double* sw, bool want_jacobian, double* partials /*[MAX_PARTIALS]*/, int* n_partials, hipStream_t st);
void launch_prior(const GraphDev& G, const double* pose8, bool want_jacobian, double* partial_cost /*1 double*/, hipStream_t st);
void launch_k2(const GraphDev& G, const LinDev& L, bool want_offdiag, hipStream_t st, const MfDev* F = nullptr /* matrix-free tiles: the per-keyframe sums run on them */);
void launch_scale_init(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, int jacobi_scaling, hipStream_t st);
void launch_lm_diag(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, double min_diag, double max_diag, hipStream_t st);
void launch_build_rows(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, double radius, int add_lambda, double* lam_out /*null: write BSR blocks*/, hipStream_t st);
void launch_mf_compact(const GraphDev& G, const MfDev& F, const double* pose8, const double* sw, hipStream_t st);
void launch_mf_spmv(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, int k, double tol2, hipStream_t st);
void launch_mf_apply(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const double* x, double* y, hipStream_t st);
void launch_mf_apply_dot(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const double* x, double* y, hipStream_t st);   // + partial sums of x.y in C.part_pq[0 .. mf_grid_size)
void launch_invert_rows(const GraphDev& G, const CgDev& C, int32_t* fail_flag, hipStream_t st);
void launch_cg_init(const GraphDev& G, const CgDev& C, int warm /*x holds a previous solution, q = A x*/, double tol2, hipStream_t st);
// multi-GPU: the vector half of cg_init (owner-weighted partials of r.u in part_rz and of b.M^-1 b in part_pq); returns their count
int launch_cg_init_vectors(const GraphDev& G, const CgDev& C, int warm, hipStream_t st);
void launch_cg_init_scalars(const CgDev& C, int nparts, int nparts_bb, double tol2, hipStream_t st);   // scal[0] = b.M^-1 b, scal[1] = r.z from the partial sums; flags reset
void launch_cg_set_tolerance(const CgDev& C, double tol2, hipStream_t st);
void launch_cg_poll(const CgDev& C, int32_t* host_flags, double* host_scal, hipStream_t st);      // host_*: pinned, device-accessible
void launch_cg_spmv(const GraphDev& G, const CgDev& C, int k, double tol2, hipStream_t st);   // iteration k: direction + matvec (+ convergence test)
void launch_cg_update(const GraphDev& G, const CgDev& C, int k, int n_pq_partials, hipStream_t st);
// single-reduction (Chronopoulos-Gear) form on one GPU: w = A u with the partials of u.w (stops with the PCG), then ONE update kernel that re-reduces both dot products
void launch_mf_apply_dot_live(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, hipStream_t st);
void launch_cg_update_sr(const GraphDev& G, const CgDev& C, int k, int first, int n_pq_partials, hipStream_t st);
void launch_mf_apply_dot_live_coarse(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, int pending, hipStream_t st);      // two-level method, fused form
void launch_cg_update_restrict_sr(const GraphDev& G, const CgDev& C, const CoarseDev& K, int k, int first, int pending, int n_pq_partials, hipStream_t st);
int cg_grid_size(const GraphDev& G);
int mf_grid_size(const MfDev& F);
void launch_apply_operator(const GraphDev& G, const CgDev& C, const double* x, double* y, hipStream_t st);
void launch_model_change(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const double* delta_p, double* delta_s, double* partials, int* n_partials, hipStream_t st);
void launch_plus(const GraphDev& G, const double* pose8, const double* sw, const double* delta_p, const double* delta_s,
                 double* pose8_out, double* sw_out, double* part_step2, double* part_sw_step2, int* n_partials, hipStream_t st);
void launch_state_norms(const GraphDev& G, const LinDev& L, const double* pose8, const double* sw,
                        double* part_xnorm2, double* part_sw_xnorm2, double* part_gmax, int* n_partials, hipStream_t st);
void launch_reduce(const double* partials, int n, int op /*0 sum,1 max*/, double* out, hipStream_t st);
void launch_manifold_plus(int64_t n, const double* quat, const double* t, const double* delta, double* quat_out, double* t_out, hipStream_t st);   // parity hook
void launch_pack_pose(const double* quat, const double* t, double* pose8, int64_t N, hipStream_t st);
void launch_unpack_pose(const double* pose8, double* quat, double* t, int64_t N, hipStream_t st);
void launch_unpack_k1(const GraphDev& G, int kind, int64_t first, int64_t count, double* r, double* J1, double* J2, double* Js, hipStream_t st);
// neighbour exchanges (round 6).  Gather: buf[j][K] <- (a1[idx[j]][k1], a2[idx[j]][k2]) for the rows listed; scatter: the reverse; sum: for every shared keyframe the parts of
// its row in ascending rank order (src -1: the rank's own row in a1 / a2, else row `src` of the receive buffer) — the same bits on every rank.  `stop`: skip when the flag is set.
void launch_gather_rows(double* buf, const double* a1, int k1, const double* a2, int k2, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st);
void launch_scatter_rows(const double* buf, double* a1, int k1, double* a2, int k2, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st);
void launch_scatter_rows_dinv(const double* buf, double* r, double* x, const double* Dinv, int64_t n, const int32_t* idx, const int32_t* stop, hipStream_t st);      // r rows in, x = Dinv r formed on the spot
void launch_sum_rows(const double* buf, double* a1, int k1, double* a2, int k2, int64_t n_sh, const int32_t* sh_loc, const int32_t* sum_ptr, const int32_t* sum_src, const int32_t* stop, hipStream_t st);
// in-process communicator (pgo_comm_init_local): out[i] = sum / max over the ranks' staged buffers in rank order; peers' segments copied into the receive buffer
struct LocalPeers { const double* src[16]; int64_t off[16]; int64_t cnt[16]; int n; };      // off: destination offset (copy) — unused by the reduction
void launch_local_reduce(double* out, const LocalPeers& P, int64_t n, int op, hipStream_t st);
void launch_local_copy(double* recv, const LocalPeers& P, hipStream_t st);
// two-level preconditioner (CoarseDev)
void launch_coarse_geometry(const GraphDev& G, const CoarseDev& K, const double* pose8, hipStream_t st);       // centroids + d
void launch_coarse_assemble(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, hipStream_t st);   // Ac = P^T A P (deterministic)
void launch_coarse_symmetrize(const CoarseDev& K, hipStream_t st);                                             // mirror one triangle
void launch_coarse_shift(const CoarseDev& K, double eps, hipStream_t st);   // Ac_ii *= 1 + eps
void launch_coarse_invert(const CoarseDev& K, double* scratch /* 64 nc + 4096 doubles */, int32_t* fail, hipStream_t st);   // Ac -> Ac^-1 (blocked Gauss-Jordan)
void launch_coarse_negate(const CoarseDev& K, hipStream_t st);   // debug aid (PGO_DEBUG_BREAK_COARSE): Ac^-1 <- -Ac^-1, a preconditioner that is NOT positive definite
// z += P Ac^-1 P^T r for the vectors of the PCG (r of the given parity), r.z partials updated in place (same workgroup -> slot mapping as cg_update)
void launch_coarse_apply(const GraphDev& G, const CgDev& C, const CoarseDev& K, const double* r, double* z, double* part_rz, bool inside_iteration, hipStream_t st);
// the same preconditioner inside the PCG iteration in THREE kernels (matrix-free operator, aggregates of <= 64 keyframes; pgo_kernels.hip):
int coarse_group_keyframes(const CoarseDev& K);                 // keyframes per workgroup trip of the update kernel (whole aggregates); 0: not available
int coarse_update_grid(const GraphDev& G, const CoarseDev& K);  // its workgroups = its r.z partial slots
int coarse_solve_grid(const CoarseDev& K);                      // workgroups of the dense solve = the coarse r.z partial slots that follow them
void launch_mf_spmv_coarse(const GraphDev& G, const MfDev& F, const ScaleDev& Sc, const CgDev& C, const CoarseDev& K, int k, double tol2, int nparts, int pending, hipStream_t st);
void launch_cg_update_restrict(const GraphDev& G, const CgDev& C, const CoarseDev& K, int k, int n_pq_partials, hipStream_t st);
void launch_coarse_solve_dot(const CoarseDev& K, const int32_t* stop, double* part, hipStream_t st);
// multi-GPU PCG in Chronopoulos-Gear form (one collective per iteration): see pgo_kernels.hip
void launch_cgcg_dots(const GraphDev& G, const CgDev& C, hipStream_t st);                                     // part_pq[block] = partial of u.w
void launch_cg_reduce2_live(const CgDev& C, const double* pa, int na, const double* pb, int nb, double* out, hipStream_t st);
void launch_cgcg_update(const GraphDev& G, const CgDev& C, int k, int first, hipStream_t st, const double* xq = nullptr, const int32_t* sh_of = nullptr, const double* two = nullptr);
void launch_cgcg_scalars_init(const CgDev& C, const double* bb_src, double tol2, hipStream_t st);
// write-back: owned keyframes of the rank-local (quat[n][4], t[n][3]) into zero-initialised global arrays
void launch_scatter_owned_pose(const double* quat, const double* t, int64_t n, const int32_t* l2g, const double* own, double* gquat, double* gt, hipStream_t st);
// K0: graph construction from raw VIO poses
void launch_vio_odometry(int64_t n, const int32_t* c1, const int32_t* c2, const double* vio, int yaw_weight, double* meas8, hipStream_t st);
void launch_vio_initial_guess(int64_t u_begin, int64_t count, const double* left, const int32_t* left_of_node, const double* vio, double* quat, double* t, hipStream_t st);

// aggregation multigrid (MgDev / MgLevelDev); levels[0] = level 1
void launch_mg_geometry(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st);
// several ranks: level 1's positions are the centroids over the members on ALL ranks: owner-weighted partial sums into levels[0].pos (all-reduced by the caller), then the rest
void launch_mg_geometry0_sum(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st);
void launch_mg_geometry_finish(const GraphDev& G, const MgDev& M, const MgLevelDev* levels, const double* pose8, hipStream_t st);
// numeric Galerkin products of the current LM system, level by level, block-Jacobi inverses of the sparse levels, the dense coarsest operator
// into K.Ac (K.nc padded), which is then inverted by launch_coarse_invert; *fail != 0: some diagonal block was not positive definite
void launch_mg_assemble(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, double omega, int32_t* fail, hipStream_t st, double prolong_scale = 0.0, bool hoff_valid = false);
// its two halves: level 1 from the keyframe system (several ranks: this rank's contributions; the caller all-reduces levels[0].val), then everything above
void launch_mg_galerkin0(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgDev& M, const MgLevelDev* levels, hipStream_t st, bool hoff_valid = false);
void launch_k2_offdiag(const GraphDev& G, const LinDev& L, hipStream_t st);
void launch_mg_assemble_fine(const GraphDev& G, const LinDev& L, const ScaleDev& Sc, const CgDev& C, const MgLevelDev& F, const MgLevelDev& T, const MgLevelDev& L1, double omega, int32_t* fail, hipStream_t st,
                             double prolong_scale, bool hoff_valid, const double* pose8 = nullptr /* filtered form (F.dlump): the keyframes' positions of the current linearisation */);      // level 1 = Ps_0^T A Ps_0 (smoothed keyframe transition), then launch_mg_assemble_rest
void launch_mg_assemble_rest(const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, double omega, int32_t* fail, hipStream_t st, double prolong_scale = 0.0 /* c = w_p / w of the smoothed transitions */,
                             int first_level = 0 /* levels[first_level].val is complete already: its inverses and everything above */);
// ... and the pieces it is made of, each on the level's set-up share (MgLevelDev::su_*): several ranks run them with block exchanges in between (pgo_solver.hip: build_mg_ranks)
void launch_mg_level_inverses(const MgLevelDev& A, double omega, int32_t* fail, hipStream_t st);       // Dinv = omega D^-1 and the fp32 copy of the blocks
void launch_mg_level_power(const MgLevelDev& A, double omega, hipStream_t st);                        // lambda_max(D^-1 A) estimate of a whole level -> A.xf[0]
// ... of a distributed level, step by step (the caller exchanges the iterate's halo before every step and all-reduces the sums): start vector on the own rows of A.x (A.xt zeroed),
// w = D^-1 A v on the own tiles, {|a|^2, |b|^2 over the own rows, failure flag} -> out3, lambda = sqrt(in3[1] / in3[0]) and the flag back
void launch_mg_power_init(const MgLevelDev& A, hipStream_t st);
void launch_mg_power_step(const MgLevelDev& A, const double* v, double* w, double omega, hipStream_t st);
void launch_mg_power_sums(const MgLevelDev& A, const double* a, const double* b, const int32_t* fail, double* out3, hipStream_t st);
void launch_mg_power_finish(const double* in3, int32_t* fail, double* lam, hipStream_t st);
void launch_mg_level_rescale(const MgLevelDev& A, const double* lam, double omega, hipStream_t st);    // Dinv scaled down where omega lambda > 1.75
void launch_mg_transition_ps(const MgLevelDev& A, double prolong_scale, hipStream_t st);              // Ps = (I - c Dinv A) P
void launch_mg_transition_w(const MgLevelDev& A, hipStream_t st);                                     // W = A Ps, R^T = Ps - Dinv W (both orientations)
void launch_mg_transition_product(const MgLevelDev& A, const MgLevelDev& B, hipStream_t st);          // B = Ps^T W (several ranks: this rank's rows' part of it)
void launch_mg_level_galerkin(const MgLevelDev& A, const MgLevelDev& B, hipStream_t st);              // B = P^T A P (plain transition)
void launch_mg_dense_top(const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, hipStream_t st);
// Several ranks: the cycle is cut into segments by the exchanges its kernels need (pgo_solver.hip issues them); launch_mg_apply calls the hook BEFORE the kernel that reads the
// exchanged vectors.  point: 0 = down-sweep of `level` (1-based; n_levels = the dense solve) is about to read x (and r) of that level, 1 = the up-sweep of `level` is about to
// read xt of that level (plain transition) or xf of level + 1 (explicit transfer operator), 2 = the prolongation to the keyframes is about to read xf of level 1.
struct MgExchangeHook { void* ctx; int (*fn)(void* ctx, int point, int level); };
// z += scale P V(P^T r) (every coarse correction inside V scaled alike), r.z partials updated in place (cg_update's workgroup -> slot mapping)
void launch_mg_apply(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, const double* r, double* z, double* part_rz, double scale, bool inside_iteration, hipStream_t st,
                     bool restricted = false /* r_1 (and x_1) already formed by launch_cg_update_mg */, double prolong_scale = 0.0 /* c of the smoothed transitions */,
                     const MgLevelDev* fine = nullptr /* smoothed keyframe transition: the keyframe level's transfer view (r_1 = Ps_0^T r and z += s Ps_0 x_1 by kernels of their own) */,
                     const MgExchangeHook* hook = nullptr /* several ranks: called where the cycle needs rows of other ranks */, int* hook_rc = nullptr);
// cg_update + r_1 = P_0^T r', x_1 = w D_1^-1 r_1 of the multigrid (M.blk_tab)
void launch_cg_update_mg(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, int k, int n_pq_partials, hipStream_t st);
void launch_cg_update_mg_sr(const GraphDev& G, const CgDev& C, const MgDev& M, const MgLevelDev* levels, const CoarseDev& K, int k, int first, int n_pq_partials, hipStream_t st);      // single-reduction form (cg_update_kernel<true>) with the same restriction

double k1_algorithmic_bytes(const GraphDev& G, bool want_jacobian);

}  // namespace pgo
