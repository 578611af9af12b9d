// pgo_solver.hip — host side of libpgo: persistent problem, device graph construction, the Ceres-compatible
// Levenberg-Marquardt trust-region controller, PCG driver, optional RCCL edge sharding, and the C-ABI (include/pgo.h).
//
// What it replaces in the reference: the `ceres::Problem` bookkeeping calls of
// PoseGraphSLAM::reinit_ceres_problem_onnewloopedge_optimize6DOF (src/PoseGraphSLAM.cpp:1340-1367,1550-1556,
// 1629-1633,1803-1849) and `ceres::Solve` (:1903) with the options at :1268-1272.  The minimiser follows
// Ceres' trust_region_minimizer.cc / levenberg_marquardt_strategy.cc control flow; the
// linear solve is a device PCG instead of SPARSE_NORMAL_CHOLESKY.  There is NO CPU fallback: without a HIP
// device pgo_create fails.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "pgo.h"
#include "pgo_internal.hpp"
#include "pgo_mg_host.hpp"
#include "pgo_comm_local.hpp"

using namespace pgo;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Debug hooks.  None of them is honoured unless the process ALSO sets PGO_ENABLE_DEBUG_HOOKS=1 (read once per process): a PGO_DEBUG_* variable that leaks into a production
// environment on its own does nothing.  The test suite sets the master switch in tests/conftest.py.
bool debug_hooks_enabled() { static const bool on = []() { const char* e = std::getenv("PGO_ENABLE_DEBUG_HOOKS"); return e && e[0] == '1' && e[1] == 0; }(); return on; }
// PGO_DEBUG_POISON=1 (read once per process): every new device allocation is filled with 0xFF bytes — a NaN in every double / float, -1 in every index —
// so that a kernel reading memory nobody wrote fails the same way on every box instead of depending on what the allocation held before (tests/test_gpu_determinism.py runs
// its solves under it and compares the results bit for bit with an unpoisoned run).
bool debug_poison() { static const bool on = []() { const char* e = std::getenv("PGO_DEBUG_POISON"); return debug_hooks_enabled() && e && e[0] == '1' && e[1] == 0; }(); return on; }

// PGO_DEBUG_BREAK_COARSE=1 (read at every operator build so that a test can switch it inside one process): the dense coarse inverse of the two-level method /
// of the multigrid's coarsest level is applied with the wrong sign — a preconditioner that is not positive definite, i.e. a forced PCG breakdown.
bool debug_break_coarse() { if (!debug_hooks_enabled()) return false; const char* e = std::getenv("PGO_DEBUG_BREAK_COARSE"); return e && e[0] == '1' && e[1] == 0; }
// PGO_DEBUG_GRAPH_AFTER=<n> (read once): the PCG captures its chunk as a hipGraph after n eager iterations instead of 192; values that are not an even number in [2, 10^6] are ignored
int debug_graph_after() {
    static const int v = []() {
        const char* e = std::getenv("PGO_DEBUG_GRAPH_AFTER");
        if (!debug_hooks_enabled() || !e) return 192;
        char* end = nullptr; const long n = std::strtol(e, &end, 10);
        return (end && *end == 0 && n >= 2 && n <= 1000000 && (n & 1) == 0) ? (int)n : 192;
    }();
    return v;
}

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        if (e == hipSuccess && debug_poison()) { e = hipMemset(p, 0xFF, want * sizeof(T)); if (e == hipSuccess) e = hipDeviceSynchronize(); }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// a pair of timing events destroyed on EVERY exit of the function that holds it (the HIPCHK early returns included)
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&e0); if (e == hipSuccess) e = hipEventCreate(&e1); return e; }
    ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

// function-local device scratch (freed at scope exit; DBuf members of pgo_problem are released by pgo_destroy)
template <class T>
struct ScopedBuf : DBuf<T> { ~ScopedBuf() { this->release(); } };

struct HostClass {
    std::vector<int32_t> c1, c2, sw;
    std::vector<double> meas;   // 8 per edge: q_obs(4) t_obs(3) w
    int64_t size() const { return (int64_t)c1.size(); }
};

// scalar slots
enum { S_COST = 0, S_PRIOR_COST = 1, S_MODEL = 2, S_SW_STEP2 = 3, S_SW_XNORM2 = 4, S_GMAX = 5, S_STEP2 = 6, S_XNORM2 = 7, S_N = 8 };

// ---- RCCL through dlopen (only touched when pgo_comm_* is used) ----
struct Rccl {
    struct Uid { char b[128]; };   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
    void* h = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Uid, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;      // ncclSend(buf, count, type, peer, comm, stream)
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

// Host half of a multigrid build (pgo_mg_host.hpp's hierarchy + the pooled index arrays and offsets of its device image): no HIP call, no collective in here when the handle is a
// single rank's — that half can therefore run on a worker thread while the stream works on other LM steps (regroup, below); mg_install() uploads it.
struct MgPrepared {
    bool ok = false;
    pgo_mg::Hierarchy H;
    std::vector<int32_t> agg0_l, mem0_ptr_l, mem0_l;      // several ranks: the keyframe-indexed arrays in the handle's local numbering
    std::vector<double> inv_cnt;
    std::vector<int32_t> pi32; std::vector<int64_t> pi64; size_t nf64 = 0;
    struct Off { size_t col, parent, agg_ptr, tile, tile_rows, rowptr, g_ptr, g_ent, val, Dinv, pos, d, r, x, xt, xf, valf, ps_rowptr, ps_col, w_rowptr, w_col, psT_ptr, psT_ent, ps_val, w_val, t, u, y, zero, row_of, tr_of, ps_row, w_row,
                        rt_rows, rT_rows, rT_col, rT_of_w, ps_of_w, rt_valf, r_valf; int rT_tiles, rT_seg_shift; };
    std::vector<Off> off;
    size_t o_agg0 = 0, o_mem0_ptr = 0, o_mem0 = 0, o_blk_tab = 0, o_d0 = 0, o_inv = 0;
    bool have_tab = false;
    // smoothed keyframe transition: the keyframe level F (pgo_mg_host.hpp) — its own block pattern and contribution ids, Ps / W structure, Ps by level-1 row for the restriction
    bool fine = false;
    struct FineOff { size_t rowptr, col, ent, ps_rowptr, ps_col, w_rowptr, w_col, psT_ptr, psT_ent, ps_row, w_row, rT_of_ps, rT_col, rT_rows, val, Dinv, ps_val, w_val, rt_valf, r_valf, lump; int rT_tiles, rT_seg_shift; bool filtered; } fo{};
    // several ranks: who sends which rows of the level vectors to whom (plans[l]: level l+1; the last: the dense level's residual), and this rank's share of every level
    std::vector<pgo_mg::ExchangePlan> plans;
    std::vector<size_t> o_plan_send, o_plan_recv;      // offsets of the plans' index lists in the int32 pool
    struct Share { int32_t tile0 = 0, tiles_own = 0, rT_row0 = 0, rT_row1 = 0; bool distributed = false; };
    std::vector<Share> share;
    int32_t a0 = 0, a1 = 0;                            // the rank's own level-1 aggregates
    // several ranks, distributed set-up: the block exchanges of every distributed level (pgo_mg_host.hpp: SetupPlans) and where their index lists sit in the int32 pool
    pgo_mg::SetupPlans setup;
    struct SetupOff { size_t val_send = 0, val_dst = 0, val_sum_ptr = 0, val_sum_src = 0, ps_send = 0, ps_recv = 0, rv_send = 0, rv_recv = 0, prod = 0; };
    std::vector<SetupOff> o_setup;
    std::vector<int32_t> g0_slots; size_t o_g0 = 0;      // distributed set-up: the level-1 blocks this rank contributes to
    std::vector<double> sw_built;          // [Es] s^2 of every switchable edge this hierarchy was matched with
    double moved = 0.0, of_edges = 0.0, host_ms = 0.0;
};

}  // namespace

struct pgo_problem {
    pgo_options opt;
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;

    HostClass rel, swe;
    std::vector<PriorDev> priors;
    std::vector<int32_t> constant_nodes;
    bool graph_dirty = true, priors_dirty = true;

    // device graph.  N = keyframes this handle works on: all of the caller's (one GPU), or — multi-GPU — only those touched by the
    // rank's own residual blocks, renumbered densely in ascending global order ("rank-local subgraph"); N_global = the caller's count.
    int64_t N = 0, S = 0, N_global = 0;
    bool local_ids = false;
    std::vector<int32_t> l2g, g2l;            // local -> global, global -> local (-1: not touched by this rank)
    std::vector<uint8_t> h_touched_any;       // [N_global] some rank holds a residual block on the keyframe
    std::vector<double> h_own;                // [N] 1.0 where this rank is the keyframe's owner (lowest rank touching it)
    std::vector<double> h_init_q, h_init_t;   // multi-GPU: the caller's state at solve_begin (keyframes no rank touches are returned as given)
    std::vector<uint64_t> h_touch_mask;       // [N_global] bit r: rank r holds a residual block on the keyframe (one all-reduce at graph build)
    std::vector<int32_t> h_owner;             // [N_global] the rank that owns the keyframe: the one holding most of its residual blocks (a second all-reduce), -1: nobody touches it
    pgo_mg::FinePlan fine_plan;               // the keyframes' neighbour exchange: who shares which keyframes with this rank, and the order their parts are summed in
    DBuf<int32_t> d_l2g, d_fp_send, d_fp_shloc, d_fp_sumptr, d_fp_sumsrc;
    DBuf<double> d_own, d_xsend[2], d_xrecv, d_xscal;   // owner weights; send buffers (by collective parity), receive buffer, the iteration's two scalars
    int64_t n_sh_mine = 0, n_sh_global = 0;
    // multigrid level exchanges (installed with the hierarchy)
    struct LevelPlanDev { const int32_t* send_idx = nullptr; const int32_t* recv_idx = nullptr; const pgo_mg::ExchangePlan* plan = nullptr; };
    std::vector<LevelPlanDev> lvl_plan;
    std::vector<pgo_mg::ExchangePlan> mg_plans;   // several ranks: the installed hierarchy's level plans (their segment bounds are read at every exchange)
    std::vector<uint8_t> mg_dist;             // per sparse level: its kernels run on the owner's rows only
    // distributed set-up (round 6): levels [0, mg_first_whole) form the numbers of their own rows only, level mg_first_whole is gathered, the rest is set up by every rank;
    // 0: the set-up is replicated (one GPU; level 1 not distributed; pgo_options.mg_dist_setup = 0)
    pgo_mg::SetupPlans mg_setup; int mg_first_whole = 0;
    int32_t mg_fw_row0 = 0, mg_fw_row1 = 0; int64_t mg_fw_blk0 = 0, mg_fw_blk1 = 0;      // this rank's rows / blocks of level mg_first_whole (it forms them, then all ranks gather the level)
    struct SetupPlanDev { const int32_t* val_send = nullptr; const int32_t* val_dst = nullptr; const int32_t* val_sum_ptr = nullptr; const int32_t* val_sum_src = nullptr;
                          const int32_t* ps_send = nullptr; const int32_t* ps_recv = nullptr; const int32_t* rv_send = nullptr; const int32_t* rv_recv = nullptr; };
    std::vector<SetupPlanDev> su_plan;
    struct OwnRange { int64_t row0 = 0, row1 = 0, blk0 = 0, blk1 = 0, ps0 = 0, ps1 = 0, w0 = 0, w1 = 0, rT0 = 0, rT1 = 0; };      // this rank's rows of every level and the block ranges they span (one GPU, and levels every rank runs completely: everything)
    std::vector<OwnRange> mg_own;
    int mg_levels_distributed = 0; int64_t mg_rows_total = 0, mg_rows_own = 0, mg_blocks_total = 0, mg_blocks_own = 0;
    // exchange accounting (pgo_get_sharding_stats)
    int64_t st_exchanges = 0, st_allreduces = 0, st_pcg_iterations = 0; double st_bytes_neighbour = 0.0, st_bytes_allreduce = 0.0;
    DBuf<int32_t> d_rc1, d_rc2, d_sc1, d_sc2, d_sidx, d_bsr_col;
    DBuf<double> d_rmeas, d_smeas;
    DBuf<int4> d_rwin, d_swin;
    DBuf<PriorDev> d_prior;
    DBuf<int64_t> d_inc_rowptr, d_inc, d_bsr_rowptr;
    DBuf<uint8_t> d_node_free;
    DBuf<double> d_Jr, d_Js, d_Jp;
    DBuf<double> d_Hd_g;             // Hd [N][36] followed by g [N][6]  (contiguous: one all-reduce)
    DBuf<double> d_Hoff, d_c, d_hss, d_gs;
    DBuf<double> d_scale_p, d_scale_s, d_diag_p, d_diag_s, d_a_inv;
    DBuf<double> d_val, d_Dtot_b;   // Dtot [N][36] followed by b [N][6]
    DBuf<float> d_Lf;
    DBuf<double> d_cgvec;            // x r r2 z p p2 q  (7 x [N][6])
    DBuf<double> d_part;             // partial-sum scratch: several arrays of n_part
    DBuf<double> d_cgpart;           // part_pq [MAX] + part_rz [2][MAX] + scal[4]
    DBuf<int32_t> d_flags;           // cg flags [4] + invert fail [1]
    DBuf<double> d_scal;             // S_N doubles
    DBuf<double> d_pose[2], d_swv[2], d_delta_s, d_io;   // state ping-pong, staging for quat/t
    DBuf<double> d_tmp;
    DBuf<double> d_vio;              // raw VIO poses [n_vio][16] (graph construction, K0)
    DBuf<int32_t> d_vio_idx; DBuf<double> d_vio_meas;   // K0's edge endpoints and measurements of one call
    // two-level preconditioner (CoarseDev)
    DBuf<double> d_ccen, d_cd, d_cAc, d_crc, d_cscr;
    DBuf<float> d_cAcf;              // the dense inverse rounded to fp32
    DBuf<int64_t> d_cblk_ptr, d_ccontrib;
    DBuf<int32_t> d_cblk_ab, d_cagg_free, d_cinfo;
    CoarseDev K{};
    bool coarse_built = false, coarse_active = false;
    int coarse_mode = 0;             // per solve: 0 not yet compared with plain block-Jacobi, 1 keep, 2 dropped (it did not pay on this graph)
    int coarse_retests = 0; bool coarse_skip_all = false; double coarse_drop_radius = 0.0;   // dropped at a small radius: one more comparison once the radius reaches coarse_min_radius
    int coarse_backoff = 0, coarse_skip = 0;   // a handle that keeps dropping it (incremental triggers on the same kind of graph) retests ever more rarely
    int coarse_keep_streak = 0;      // consecutive solves that kept it: the comparison is then repeated only every 4th solve
    uint64_t coarse_geometry_epoch = 0, lin_epoch = 0;   // lin_epoch counts linearisations (the centroids follow the poses)
    // aggregation multigrid (MgDev): hierarchy arrays live in three pooled buffers
    DBuf<double> d_mg_f64; DBuf<int32_t> d_mg_i32; DBuf<int64_t> d_mg_i64;
    MgDev M{}; MgLevelDev mg_levels[MG_MAX_LEVELS];
    int mg_fine_auto = -1;                 // mg_smoothed_fine < 0: the decision of this graph build (-1 not taken yet, 0 / 1), written by the hierarchy's worker before it is joined
    bool mg_fine = false; MgLevelDev mg_fineF{}, mg_fineT{};      // smoothed keyframe transition (opt.mg_smoothed_fine): the keyframe level's set-up view and transfer view
    bool mg_built = false, mg_active = false;
    pgo_mg::BuildCache mg_cache;           // what the hierarchy builder keeps for a regroup of the same graph
    std::vector<double> mg_sw_built;       // [Es] s^2 of every switchable edge the current hierarchy was built with
    int mg_regroups = 0;                   // regroups of this solve
    // a regroup in flight: the host half of the rebuild runs on a worker thread from the LM iteration that found the switches moved (after an accepted step) and is
    // installed where multigrid operators are next built — both points depend on the solve's own history only, never on timing
    std::thread mg_job; std::unique_ptr<MgPrepared> mg_job_out, mg_job_old; bool mg_job_running = false;
    int rc_job = 0;
    // the hierarchy of a freshly built graph (one GPU): its host half is started on a worker thread at the top of build_graph and INSTALLED where it is first needed —
    // build_mg (the first LM system that wants multigrid operators), the end of the solve, or the next solve_begin; until then mg_built is true ("this graph has a
    // hierarchy") and the level descriptors are empty.  When it is needed is a property of the solve, how long the wait is not: results do not depend on timing.
    std::thread mg_init_thread; std::unique_ptr<MgPrepared> mg_init_out; bool mg_init_pending = false; int rc_mg_init = 0;
    uint64_t mg_geometry_epoch = 0;
    uint64_t hoff_epoch = 0;               // linearisation whose J1^T J2 blocks L.Hoff holds (matrix-free solver: formed on demand for the multigrid's level-1 product)
    int64_t n_vio = 0;
    // matrix-free operator
    DBuf<uint32_t> d_einc;
    DBuf<uint32_t> d_einc_slot;
    DBuf<ushort4> d_node_rng;
    DBuf<int64_t> d_tile_inc0;
    DBuf<int32_t> d_einc_other, d_tile_node0, d_tile_sw0, d_node_prior;
    DBuf<double2> d_rec;
    DBuf<double> d_lam;
    MfDev F{};
    bool built_mf = false;
    int64_t mf_pair_lanes = 0, mf_rel_side_lanes = 0, mf_sw_lanes = 0;   // lanes of the matrix-free operator by kind (pgo_time_kernel's bytes)
    int cur = 0;
    int64_t n_part = MAX_PARTIALS;
    std::vector<uint8_t> h_node_free, h_sw_used;
    int64_t nnzb = 0;

    GraphDev G{};
    LinDev L{};
    ScaleDev Sc{};
    CgDev C{};

    // LM state
    bool in_solve = false, scale_ready = false, terminated = false, have_prev_step = false;
    double radius = 0, decrease_factor = 2, x_cost = 0, x_norm = 0, gmax = 0;
    bool reuse_diagonal = false;
    int iteration = 0, invalid = 0;
    double t_begin = 0, t_device0 = 0;
    pgo_summary sum;

    // comm
    Rccl nccl; void* comm = nullptr; int rank = 0, world = 1;
    pgo_allreduce_fn custom_allreduce = nullptr; void* custom_ctx = nullptr;
    pgo_exchange_fn custom_exchange = nullptr;
    pgo_local::Group* local_group = nullptr; uint64_t lc_count = 0;      // in-process communicator: collectives issued so far (parity = count & 1)
    std::vector<int64_t> x_off_send, x_off_recv;                          // scratch: segment bounds in doubles of the exchange in flight

    // pipelined convergence polling: pinned host copies of {flags[4], scal[4]} for two chunks in flight
    struct Poll { int32_t flags[4]; double scal[4]; };
    Poll* poll = nullptr; hipEvent_t poll_ev[2] = {nullptr, nullptr};      // poll[2]: snapshot at the start of a PCG phase (base point of the convergence-rate estimate)

    // hipGraph of one PCG chunk (launch-bound inner loop); valid for (graph build epoch, tolerance, chunk length, solver)
    // one captured chunk per preconditioner (0 block-Jacobi, 1 two-level, 2 multigrid): the hybrid policy changes between them inside a solve
    struct CapturedChunk { hipGraphExec_t exec = nullptr; int len = 0; uint64_t epoch = 0; double scale = 0.0; bool sr = false; };   // scale: mg_correction_scale is a by-value kernel argument of the captured cycle
    CapturedChunk cg_chunk[3];
    hipGraphExec_t cg_graph = nullptr;   // the one in use (not owned)
    uint64_t build_epoch = 1; bool cg_graph_failed = false;
    double cg_predicted = 0.0;      // block-Jacobi-equivalent iterations predicted for the current LM system (build_system); 0: none
    double cg_prev_equiv = 0.0, cg_prev_radius = 0.0;   // block-Jacobi-equivalent PCG iterations and radius of the last fully solved LM system of this solve
    int mg_switch_at = 400;              // in-flight switch point of the current LM system (build_system)
    int cg_extra = 0;                    // PCG iterations of the current LM step spent before a change of preconditioner
    bool mg_failed = false;              // the multigrid operators of the current system could not be built
    double last_rho = 1.0;               // relative decrease of the last accepted step of this solve
    bool mg_start_deferred = false;      // the current system is predicted hard, but its multigrid operators are built only once the step has survived the first early-rejection pause
};

namespace {

#define HIPCHK(p, expr)                                                                            \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (p)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                         \
            return e__ == hipErrorOutOfMemory ? PGO_ERR_OUT_OF_MEMORY : PGO_ERR_HIP;               \
        }                                                                                          \
    } while (0)

int set_device(pgo_problem* p) { HIPCHK(p, hipSetDevice(p->device)); return PGO_OK; }

// Matrix4d (column-major 16) -> Meas fields
void meas_from_matrix(const double* T, double w, double* out8) {
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = T[c * 4 + r];
    double q[4];
    eigen_matrix_to_quat(R, q);   // CeresResidues.h:24 / :150
    out8[0] = q[0]; out8[1] = q[1]; out8[2] = q[2]; out8[3] = q[3];
    out8[4] = T[12]; out8[5] = T[13]; out8[6] = T[14]; out8[7] = w;
}

int upload_class(pgo_problem* p, const HostClass& H, bool is_sw, DBuf<int32_t>& dc1, DBuf<int32_t>& dc2, DBuf<int32_t>& dsw, DBuf<double>& dmeas,
                 DBuf<int4>& dwin, EdgeClassDev& out) {
    const int32_t* g2l = p->local_ids ? p->g2l.data() : nullptr;
    const int64_t E = H.size();
    const int64_t Epad = (E + TILE - 1) / TILE * TILE;
    const int tiles = (int)(Epad / TILE);
    std::vector<int32_t> c1(Epad), c2(Epad), sw(is_sw ? Epad : 0);
    std::vector<double> meas((size_t)8 * Epad);
    std::vector<int4> win(tiles);
    for (int64_t e = 0; e < Epad; ++e) {
        const int64_t s = e < E ? e : E - 1;   // padding lanes replicate the last edge (computed, never stored or counted)
        c1[e] = g2l ? g2l[H.c1[s]] : H.c1[s]; c2[e] = g2l ? g2l[H.c2[s]] : H.c2[s];
        if (is_sw) sw[e] = H.sw[s];
        for (int k = 0; k < 8; ++k) meas[(size_t)k * Epad + e] = H.meas[(size_t)s * 8 + k];
    }
    for (int t = 0; t < tiles; ++t) {
        int lo1 = INT32_MAX, hi1 = -1, lo2 = INT32_MAX, hi2 = -1;
        for (int l = 0; l < TILE; ++l) {
            const int64_t e = (int64_t)t * TILE + l;
            lo1 = std::min(lo1, c1[e]); hi1 = std::max(hi1, c1[e]); lo2 = std::min(lo2, c2[e]); hi2 = std::max(hi2, c2[e]);
        }
        const int n1 = hi1 - lo1 + 1, n2 = hi2 - lo2 + 1;
        win[t] = make_int4(lo1, n1 <= WIN_MAX ? n1 : 0, lo2, n2 <= WIN_MAX ? n2 : 0);
    }
    if (Epad > 0) {
        HIPCHK(p, dc1.ensure(Epad)); HIPCHK(p, dc2.ensure(Epad)); HIPCHK(p, dmeas.ensure((size_t)8 * Epad)); HIPCHK(p, dwin.ensure(tiles));
        HIPCHK(p, hipMemcpyAsync(dc1.p, c1.data(), Epad * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(dc2.p, c2.data(), Epad * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(dmeas.p, meas.data(), (size_t)8 * Epad * sizeof(double), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(dwin.p, win.data(), tiles * sizeof(int4), hipMemcpyHostToDevice, p->st));
        if (is_sw) { HIPCHK(p, dsw.ensure(Epad)); HIPCHK(p, hipMemcpyAsync(dsw.p, sw.data(), Epad * sizeof(int32_t), hipMemcpyHostToDevice, p->st)); }
        HIPCHK(p, hipStreamSynchronize(p->st));   // host vectors die at scope exit
    }
    out.c1 = dc1.p; out.c2 = dc2.p; out.meas = dmeas.p; out.swidx = is_sw ? dsw.p : nullptr; out.win = dwin.p;
    out.E = E; out.Epad = Epad; out.tiles = tiles; out.J = nullptr;
    return PGO_OK;
}

int allreduce(pgo_problem* p, double* buf, size_t n, int op);
int host_allreduce(pgo_problem* p, std::vector<double>& v, int op);

// The aggregation multigrid's hierarchy for the graph of this handle and the given switch values (host array over the caller's switches, or null): host-side structure
// (pgo_mg_host.hpp), pooled device arrays, level descriptors.  Called by build_graph, and again inside a solve when the switch values have moved far from the ones the
// hierarchy was built with (regroup): the levels above level 1 are matched along the couplings that are alive NOW.  p->mg_cache keeps what does not depend on the switches.
// host half: hierarchy + pooled index arrays.  Reads the handle's edge lists, options and mg_cache only (single rank: no HIP, no collective -> may run on a worker thread)
int mg_prepare_impl(pgo_problem* p, const double* sw_now, MgPrepared& Q);
// (runs on worker threads as well: nothing may escape — the C-ABI never throws, and an exception leaving a std::thread is std::terminate)
int mg_prepare(pgo_problem* p, const double* sw_now, MgPrepared& Q) {
    try { return mg_prepare_impl(p, sw_now, Q); }
    catch (const std::bad_alloc&) { Q.ok = false; return PGO_ERR_OUT_OF_MEMORY; }
    catch (...) { Q.ok = false; return PGO_ERR_OUT_OF_MEMORY; }
}
int mg_prepare_impl(pgo_problem* p, const double* sw_now, MgPrepared& Q) {
    const int64_t N = p->N, Ng = p->N_global, S = p->S;
    const int64_t Er = p->rel.size(), Es = p->swe.size();
    int rc;
    const double t0 = now_s();
    pgo_mg::Hierarchy& H = Q.H;
    pgo_mg::timing() = p->opt.verbosity > 1;
    const int dense_max = std::max(1, std::min(p->opt.mg_dense_max_nodes, 512));
    // smoothed prolongators (denser coarse operators, two more row products per cycle on each such level) pay while the coarse levels are latency-bound: measured
    // C4 (200k keyframes) 3.56 -> 2.37 s, C5 (1M keyframes, level 1 = 125k nodes: bandwidth-bound) 8.5 -> 11.1 s.  -1 = by size; with them aggregates of 4 above level 1, else of 8
    const double loop_discount = std::max(0.0, p->opt.mg_loop_discount);
    const int n_smoothed = p->opt.mg_smoothed_levels < 0 ? (Ng <= 500000 ? 1 : 0) : std::min(p->opt.mg_smoothed_levels, MG_MAX_LEVELS);
    const int passes0 = std::max(1, std::min(p->opt.mg_first_passes, 3)), passes = p->opt.mg_passes <= 0 ? (n_smoothed > 0 ? 2 : 3) : std::min(p->opt.mg_passes, 3);
    std::vector<double> sw_w;
    if (sw_now && S > 0) { sw_w.resize((size_t)Es); for (int64_t e = 0; e < Es; ++e) { const double sv = sw_now[p->swe.sw[e]]; sw_w[e] = sv * sv; } }
    Q.sw_built.assign((size_t)Es, 1.0);
    if (!sw_w.empty()) Q.sw_built = sw_w;
    bool ok;
    std::vector<int32_t>& agg0_l = Q.agg0_l; std::vector<int32_t>& mem0_ptr_l = Q.mem0_ptr_l; std::vector<int32_t>& mem0_l = Q.mem0_l;
    std::vector<double>& inv_cnt = Q.inv_cnt;
    std::vector<int64_t> fine_rowptr, fine_ent; std::vector<int32_t> fine_col;
    // smoothed keyframe transition (one GPU): 1 = on, 0 = off, < 0 = BY THE DENSITY OF THE LEVELS IT MAKES (round 6).  It halves the multigrid iterations everywhere and pays
    // while its denser levels are still latency-sized: over the eight graph types measured in round 5 the sparse levels of the smoothed hierarchy hold 43 000 - 375 000 blocks where
    // it wins (+9 ... +52 %) and 714 000 - 3.9 M where it loses (-19 ... -43 %).  So the hierarchy is built WITH it (graphs beyond 80 000 keyframes are not tried: C3's 100 000 give
    // 734 000 blocks), its blocks are counted, and above SMOOTHED_FINE_MAX_BLOCKS it is built again without (level-0 matching and level-1 structure come from the cache; all of
    // this runs on the worker thread beside build_graph).  Decided once per graph build — a regroup keeps the decision.
    // The limit: round 5's eight graph types are separated by anything between 375 000 and 714 000; round 6's soak of 36 random graphs of 5 000 - 80 000 keyframes
    // (scripts/gpu_mid_soak.py, profiles/r06_mid_soak.txt) found the zone in between mixed — 300 000 blocks -28 % (10 000 keyframes, f = 1..5 + yaw, plain loops), 351 000 +4.5 %,
    // 371 000 -24 %, 375 000 +18 % — and nothing below 280 000 that loses: a missed gain costs less than a regression, so the limit sits under the mixed zone.
    constexpr int64_t SMOOTHED_FINE_MAX_BLOCKS = 280000, SMOOTHED_FINE_TRY_MAX_KEYFRAMES = 80000;
    bool want_fine = !p->local_ids && (p->opt.mg_smoothed_fine > 0 || (p->opt.mg_smoothed_fine < 0 && (p->mg_fine_auto == 1 || (p->mg_fine_auto < 0 && Ng <= SMOOTHED_FINE_TRY_MAX_KEYFRAMES))));
    const bool fine_on_trial = want_fine && p->opt.mg_smoothed_fine < 0 && p->mg_fine_auto < 0;
    auto fine_pattern = [&]() {
        // the keyframe level's block pattern: row i = block (i, i), then one block per incident edge (relative-pose edges first, each class in edge order), and what each block IS for
        // fine_block_value (kind 0: the keyframe's reduced diagonal block; 1 / 2: a relative-pose edge seen from its first / second keyframe; 3 / 4: a switchable edge)
        fine_rowptr.assign((size_t)N + 1, 0);
        for (int64_t e = 0; e < Er; ++e) { fine_rowptr[(size_t)p->rel.c1[e] + 1]++; fine_rowptr[(size_t)p->rel.c2[e] + 1]++; }
        for (int64_t e = 0; e < Es; ++e) { fine_rowptr[(size_t)p->swe.c1[e] + 1]++; fine_rowptr[(size_t)p->swe.c2[e] + 1]++; }
        for (int64_t n = 0; n < N; ++n) fine_rowptr[(size_t)n + 1] += fine_rowptr[n] + 1;
        fine_col.resize((size_t)fine_rowptr[N]); fine_ent.resize((size_t)fine_rowptr[N]);
        std::vector<int64_t> fillb((size_t)N);
        for (int64_t n = 0; n < N; ++n) { fine_col[(size_t)fine_rowptr[n]] = (int32_t)n; fine_ent[(size_t)fine_rowptr[n]] = (n << 3) | 0; fillb[n] = fine_rowptr[n] + 1; }
        auto add = [&](int64_t e, int32_t a, int32_t b, int kind) {
            fine_col[(size_t)fillb[a]] = b; fine_ent[(size_t)fillb[a]++] = (e << 3) | kind;
            fine_col[(size_t)fillb[b]] = a; fine_ent[(size_t)fillb[b]++] = (e << 3) | (kind + 1);
        };
        for (int64_t e = 0; e < Er; ++e) add(e, p->rel.c1[e], p->rel.c2[e], 1);
        for (int64_t e = 0; e < Es; ++e) add(e, p->swe.c1[e], p->swe.c2[e], 3);
    };
    if (want_fine) fine_pattern();
    // filtered smoothed keyframe transition: the blocks that enter the prolongator — the keyframe's own block and its relative-pose (odometry) edges; switchable loop closures do not
    std::vector<uint8_t> fine_keep;
    const bool filtered = want_fine && p->opt.mg_fine_filter != 0 && Es > 0;
    if (filtered) { fine_keep.resize(fine_ent.size()); for (size_t k = 0; k < fine_ent.size(); ++k) fine_keep[k] = (fine_ent[k] & 7) <= 2 ? 1 : 0; }
    if (!p->local_ids) {
        ok = pgo_mg::build_hierarchy(N, p->h_node_free, p->rel.c1, p->rel.c2, p->rel.meas.data() + 7, 8, p->swe.c1, p->swe.c2, sw_w.empty() ? nullptr : sw_w.data(), passes0, passes, dense_max, MG_TILE_ROWS,
                                     MG_MAX_LEVELS, H, false, MG_BLOCK0, nullptr, n_smoothed, loop_discount, &p->mg_cache, want_fine ? &fine_rowptr : nullptr, want_fine ? &fine_col : nullptr,
                                     nullptr, filtered ? &fine_keep : nullptr);
        if (fine_on_trial) {
            int64_t blocks = 0;
            if (ok) for (size_t l = 0; l + 1 < H.L.size(); ++l) blocks += (int64_t)H.L[l].col.size();
            const bool keep = ok && blocks <= SMOOTHED_FINE_MAX_BLOCKS;
            if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] multigrid: smoothed keyframe transition on trial: its sparse levels hold %lld blocks (limit %lld) -> %s\n", (long long)blocks, (long long)SMOOTHED_FINE_MAX_BLOCKS, keep ? "kept" : "not used");
            p->mg_fine_auto = keep ? 1 : 0;
            if (!keep) {
                want_fine = false;
                ok = pgo_mg::build_hierarchy(N, p->h_node_free, p->rel.c1, p->rel.c2, p->rel.meas.data() + 7, 8, p->swe.c1, p->swe.c2, sw_w.empty() ? nullptr : sw_w.data(), passes0, passes, dense_max, MG_TILE_ROWS,
                                             MG_MAX_LEVELS, H, false, MG_BLOCK0, nullptr, n_smoothed, loop_discount, &p->mg_cache, nullptr, nullptr);
            }
        }
    } else {
        // Several ranks: every rank gathers the endpoints and weights of ALL edges (one all-reduce of a zero-padded buffer: 24 B per edge, once per graph build)
        // and builds the same hierarchy from the global graph; its own edges and owned keyframes are what it contributes to level 1 (pgo_mg_host.hpp).
        std::vector<double> cnt((size_t)2 * p->world, 0.0);
        cnt[(size_t)2 * p->rank] = (double)Er; cnt[(size_t)2 * p->rank + 1] = (double)Es;
        if ((rc = host_allreduce(p, cnt, 0)) != PGO_OK) return rc;
        int64_t ErT = 0, EsT = 0, my_r = 0, my_s = 0;
        for (int r = 0; r < p->world; ++r) { if (r == p->rank) { my_r = ErT; my_s = EsT; } ErT += (int64_t)(cnt[(size_t)2 * r] + 0.5); EsT += (int64_t)(cnt[(size_t)2 * r + 1] + 0.5); }
        std::vector<double> buf((size_t)3 * (ErT + EsT), 0.0);
        double* b_rc1 = buf.data(); double* b_rc2 = b_rc1 + ErT; double* b_rw = b_rc2 + ErT; double* b_sc1 = b_rw + ErT; double* b_sc2 = b_sc1 + EsT; double* b_sw = b_sc2 + EsT;
        for (int64_t e = 0; e < Er; ++e) { b_rc1[my_r + e] = p->rel.c1[e]; b_rc2[my_r + e] = p->rel.c2[e]; b_rw[my_r + e] = p->rel.meas[(size_t)8 * e + 7]; }
        for (int64_t e = 0; e < Es; ++e) { b_sc1[my_s + e] = p->swe.c1[e]; b_sc2[my_s + e] = p->swe.c2[e]; b_sw[my_s + e] = sw_w.empty() ? 1.0 : sw_w[e]; }
        if ((rc = host_allreduce(p, buf, 0)) != PGO_OK) return rc;
        std::vector<int32_t> grc1((size_t)ErT), grc2((size_t)ErT), gsc1((size_t)EsT), gsc2((size_t)EsT);
        std::vector<double> grw(b_rw, b_rw + ErT), gsw(b_sw, b_sw + EsT);
        for (int64_t e = 0; e < ErT; ++e) { grc1[e] = (int32_t)(b_rc1[e] + 0.5); grc2[e] = (int32_t)(b_rc2[e] + 0.5); }
        for (int64_t e = 0; e < EsT; ++e) { gsc1[e] = (int32_t)(b_sc1[e] + 0.5); gsc2[e] = (int32_t)(b_sc2[e] + 0.5); }
        std::vector<uint8_t> gfree((size_t)Ng);
        for (int64_t g = 0; g < Ng; ++g) gfree[g] = p->h_touched_any[g];
        for (int32_t c : p->constant_nodes) if (c >= 0 && c < Ng) gfree[c] = 0;
        const pgo_mg::LocalContrib local{&p->l2g, &p->h_own, &p->rel.c1, &p->rel.c2, &p->swe.c1, &p->swe.c2};
        // distributed cycle: aggregates never mix owners, every level is numbered owner-major (pgo_mg_host.hpp: Owners)
        pgo_mg::Owners OW; OW.touch_mask = &p->h_touch_mask; OW.owner = &p->h_owner; OW.world = p->world; OW.dist_min_rows = p->opt.mg_dist_min_rows > 0 ? p->opt.mg_dist_min_rows : 8192;
        ok = pgo_mg::build_hierarchy(Ng, gfree, grc1, grc2, grw.data(), 1, gsc1, gsc2, (sw_now && S > 0) ? gsw.data() : nullptr, passes0, passes, dense_max, MG_TILE_ROWS, MG_MAX_LEVELS, H, false, 0, &local, n_smoothed, loop_discount, &p->mg_cache,
                                     nullptr, nullptr, p->world > 1 ? &OW : nullptr);
        if (ok && p->world > 1) {
            // the cycle's plans, and beside them on a thread of its own — the two read the finished hierarchy and write their own results — the set-up's (the set-up distributed
            // like the cycle: who contributes to / needs which blocks; the gathered edge lists are rank by rank)
            const bool want_setup = p->opt.mg_dist_setup != 0;
            std::vector<int64_t> rel_off((size_t)p->world + 1, 0), sw_off((size_t)p->world + 1, 0);
            for (int r = 0; r < p->world; ++r) { rel_off[(size_t)r + 1] = rel_off[(size_t)r] + (int64_t)(cnt[(size_t)2 * r] + 0.5); sw_off[(size_t)r + 1] = sw_off[(size_t)r] + (int64_t)(cnt[(size_t)2 * r + 1] + 0.5); }
            const bool tm = pgo_mg::timing();
            std::atomic<bool> worker_failed{false};      // (declared before the thread and its joiner: destroyed after them)
            std::thread worker;
            struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_worker{worker};
            bool started = false;
            if (want_setup && pgo_mg::host_threads() > 1) {
                try {
                    worker = std::thread([&]() { try { pgo_mg::timing() = tm; pgo_mg::build_setup_plans(H, p->rank, p->world, grc1, grc2, rel_off, gsc1, gsc2, sw_off, Q.setup); } catch (...) { worker_failed.store(true); } });
                    started = true;
                } catch (...) {}
            }
            pgo_mg::build_level_plans(H, OW, p->rank, Q.plans);
            if (started) { worker.join(); if (worker_failed.load()) throw std::bad_alloc(); }
            else if (want_setup) pgo_mg::build_setup_plans(H, p->rank, p->world, grc1, grc2, rel_off, gsc1, gsc2, sw_off, Q.setup);
        }
        if (ok) {
            const int32_t n1g = (int32_t)H.mem0_ptr.size() - 1;
            inv_cnt.resize((size_t)n1g);
            for (int32_t a = 0; a < n1g; ++a) inv_cnt[a] = 1.0 / (double)std::max(1, H.mem0_ptr[a + 1] - H.mem0_ptr[a]);
            agg0_l.resize((size_t)N); mem0_ptr_l.assign((size_t)n1g + 1, 0);
            for (int64_t l = 0; l < N; ++l) { agg0_l[l] = p->h_node_free[l] ? H.agg0[p->l2g[l]] : -1; if (agg0_l[l] >= 0) mem0_ptr_l[(size_t)agg0_l[l] + 1]++; }
            for (int32_t a = 0; a < n1g; ++a) mem0_ptr_l[(size_t)a + 1] += mem0_ptr_l[a];
            mem0_l.resize((size_t)mem0_ptr_l[n1g]);
            std::vector<int32_t> fillm(mem0_ptr_l.begin(), mem0_ptr_l.end() - 1);
            for (int64_t l = 0; l < N; ++l) if (agg0_l[l] >= 0) mem0_l[(size_t)fillm[agg0_l[l]]++] = (int32_t)l;
        }
    }
    Q.ok = ok;
    if (!ok) return PGO_OK;
    const std::vector<int32_t>& A0 = p->local_ids ? agg0_l : H.agg0;
    const std::vector<int32_t>& M0P = p->local_ids ? mem0_ptr_l : H.mem0_ptr;
    const std::vector<int32_t>& M0 = p->local_ids ? mem0_l : H.mem0;
    const int nl = (int)H.L.size();
    // this rank's share of every sparse level (one GPU, and levels every rank runs completely: all of it)
    const bool dist = H.world > 1;
    Q.share.assign((size_t)nl, MgPrepared::Share{});
    for (int l = 0; l + 1 < nl; ++l) {
        const pgo_mg::HostLevel& A = H.L[(size_t)l];
        MgPrepared::Share& sh = Q.share[(size_t)l];
        const int32_t tiles = A.tile_agg0.empty() ? 0 : (int32_t)A.tile_agg0.size() - 1;
        sh.distributed = dist && A.distributed;
        if (sh.distributed) { sh.tile0 = A.tile_ptr[(size_t)p->rank]; sh.tiles_own = A.tile_ptr[(size_t)p->rank + 1] - sh.tile0; sh.rT_row0 = H.L[(size_t)l + 1].own_ptr[(size_t)p->rank]; sh.rT_row1 = H.L[(size_t)l + 1].own_ptr[(size_t)p->rank + 1]; }
        else { sh.tile0 = 0; sh.tiles_own = tiles; sh.rT_row0 = 0; sh.rT_row1 = H.L[(size_t)l + 1].n; }
    }
    Q.a0 = dist ? H.L[0].own_ptr[(size_t)p->rank] : 0; Q.a1 = dist ? H.L[0].own_ptr[(size_t)p->rank + 1] : H.L[0].n;
    // pooled arrays: (offset, count) per array; doubles rounded up to even counts (16-B loads)
    std::vector<int32_t>& pi32 = Q.pi32; std::vector<int64_t>& pi64 = Q.pi64;
    {   // one allocation per pool (the arrays are appended one by one: without the reservation the 10-MB pools are reallocated and copied a dozen times)
        size_t n32 = A0.size() + M0P.size() + M0.size() + (size_t)((N + MG_BLOCK0 - 1) / MG_BLOCK0) * MG_BLOCK0 * 4 + 64, n64 = 0;
        for (const pgo_mg::HostLevel& A : H.L) {
            const size_t tiles = A.tile_agg0.empty() ? 0 : A.tile_agg0.size() - 1;
            n32 += 3 * A.col.size() + A.parent.size() + A.agg_ptr.size() + tiles * (4 + 2 * (size_t)MG_TILE_ROWS) + A.ps_rowptr.size() + 2 * A.ps_col.size() + A.w_rowptr.size() + 5 * A.w_col.size() + (A.smoothed ? tiles * 2 * (size_t)MG_TILE_ROWS + ((size_t)A.rT_rowptr.size() / 4 + 2) * 2 * (size_t)MG_TILE_ROWS : 0) + 16;
            n64 += A.rowptr.size() + A.g_ptr.size() + A.g_ent.size() + A.psT_ptr.size() + A.psT_ent.size();
        }
        if (H.fine_smoothed) {
            const pgo_mg::HostLevel& F = H.F;
            n32 += F.col.size() + F.ps_rowptr.size() + 4 * F.ps_col.size() + F.w_rowptr.size() + 2 * F.w_col.size() + ((size_t)H.L[0].n / 4 + 2) * 2 * (size_t)MG_TILE_ROWS + 16;
            n64 += F.rowptr.size() + F.col.size() + F.psT_ptr.size() + F.psT_ent.size();
        }
        pi32.reserve(n32); pi64.reserve(n64);
    }
    auto put32 = [&](const std::vector<int32_t>& v) { const size_t o = pi32.size(); pi32.insert(pi32.end(), v.begin(), v.end()); return o; };
    auto put64 = [&](const std::vector<int64_t>& v) { const size_t o = pi64.size(); pi64.insert(pi64.end(), v.begin(), v.end()); return o; };
    size_t& nf64 = Q.nf64;
    auto take = [&](size_t cnt) { const size_t o = nf64; nf64 += (cnt + 1) & ~(size_t)1; return o; };
    Q.off.assign((size_t)nl, MgPrepared::Off{});
    Q.o_agg0 = put32(A0); Q.o_mem0_ptr = put32(M0P); Q.o_mem0 = put32(M0);
    // slot table of the restriction inside the vector update: per run of MG_BLOCK0 keyframes its aggregates {id, 8 members as run-local bytes}
    bool have_tab = !p->local_ids && !H.fine_smoothed;      // (smoothed keyframe transition: restriction and prolongation need neighbouring runs — kernels of their own)
    if (have_tab) {
        const int64_t runs = (N + MG_BLOCK0 - 1) / MG_BLOCK0;
        std::vector<int32_t> tab((size_t)runs * MG_BLOCK0 * 4);
        for (size_t k = 0; k < tab.size(); k += 4) { tab[k] = -1; tab[k + 1] = -1; tab[k + 2] = -1; tab[k + 3] = 0; }
        std::vector<int> fill((size_t)runs, 0);
        const int32_t n1h = (int32_t)H.mem0_ptr.size() - 1;
        for (int32_t a = 0; a < n1h && have_tab; ++a) {
            const int32_t m0 = H.mem0_ptr[a], m1 = H.mem0_ptr[a + 1];
            if (m1 <= m0) continue;
            const int64_t run = H.mem0[m0] / MG_BLOCK0;
            if (m1 - m0 > 8 || fill[run] >= MG_BLOCK0) { have_tab = false; break; }
            uint32_t w[2] = {0xffffffffu, 0xffffffffu};
            for (int32_t m = m0; m < m1; ++m) {
                if (H.mem0[m] / MG_BLOCK0 != run) { have_tab = false; break; }
                const int j = m - m0;
                w[j >> 2] = (w[j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)(H.mem0[m] - run * MG_BLOCK0) << (8 * (j & 3)));
            }
            int32_t* e = &tab[((size_t)run * MG_BLOCK0 + fill[run]++) * 4];
            e[0] = a; e[1] = (int32_t)w[0]; e[2] = (int32_t)w[1];
        }
        if (have_tab) { while (pi32.size() % 4) pi32.push_back(0); Q.o_blk_tab = put32(tab); }
    }
    Q.have_tab = have_tab;
    Q.o_d0 = take((size_t)N * 3);
    const size_t n1_all = (size_t)H.L[0].n;
    Q.o_inv = p->local_ids ? take(n1_all) : 0;
    for (int l = 0; l < nl; ++l) {
        const pgo_mg::HostLevel& A = H.L[l];
        MgPrepared::Off& o = Q.off[l];
        o.col = put32(A.col); o.parent = put32(A.parent); o.agg_ptr = put32(A.agg_ptr);
        {   // block -> row, block -> slot of the transposed block (rows hold the diagonal block first, the others by ascending column)
            std::vector<int32_t> row_of(A.col.size()), tr_of(A.col.size());
            pgo_mg::parallel_ranges(A.n, pgo_mg::host_threads(), [&](int, int32_t lo, int32_t hi) {      // (rows are independent; level 2 of C3 holds 193 000 blocks)
                for (int32_t i = lo; i < hi; ++i)
                    for (int64_t k = A.rowptr[i]; k < A.rowptr[(size_t)i + 1]; ++k) {
                        row_of[(size_t)k] = i;
                        const int32_t j = A.col[(size_t)k];
                        int64_t t = k;
                        if (j != i) {
                            const int32_t* b = A.col.data() + A.rowptr[j] + 1; const int32_t* e = A.col.data() + A.rowptr[(size_t)j + 1];
                            const int32_t* f = std::lower_bound(b, e, i);
                            if (f != e && *f == i) t = f - A.col.data();
                        }
                        tr_of[(size_t)k] = (int32_t)t;
                    }
            });
            o.row_of = put32(row_of); o.tr_of = put32(tr_of);
        }
        {   // per tile {a0, a1, i0, i1}, 16-B aligned
            std::vector<int32_t> info;
            for (size_t tt = 0; tt + 1 < A.tile_agg0.size(); ++tt) { const int32_t a0 = A.tile_agg0[tt], a1 = A.tile_agg0[tt + 1]; info.insert(info.end(), {a0, a1, A.agg_ptr[a0], A.agg_ptr[a1]}); }
            while (pi32.size() % 4) pi32.push_back(0);
            o.tile = put32(info);
            std::vector<int32_t> rows;      // [tile][MG_TILE_ROWS] {first block, end block} of each row of the tile
            for (size_t tt = 0; tt + 1 < A.tile_agg0.size(); ++tt) {
                const int32_t i0 = A.agg_ptr[A.tile_agg0[tt]], i1 = A.agg_ptr[A.tile_agg0[tt + 1]];
                for (int li = 0; li < MG_TILE_ROWS; ++li) { const int32_t r = i0 + li; rows.push_back(r < i1 ? (int32_t)A.rowptr[r] : 0); rows.push_back(r < i1 ? (int32_t)A.rowptr[r + 1] : 0); }
            }
            o.tile_rows = put32(rows);
        }
        o.rowptr = put64(A.rowptr); o.g_ptr = put64(A.g_ptr); o.g_ent = put64(A.g_ent);
        o.val = take(A.col.size() * 36); o.Dinv = take((size_t)A.n * 36); o.pos = take((size_t)A.n * 3); o.d = take((size_t)A.n * 3);
        o.r = take((size_t)A.n * 6); o.x = take((size_t)A.n * 6); o.xt = take((size_t)A.n * 6); o.xf = take((size_t)A.n * 6);
        o.valf = take((A.col.size() * 36 + 1) / 2);      // fp32 copy of the blocks, carved out of the fp64 pool
        if (A.smoothed) {
            o.ps_rowptr = put32(A.ps_rowptr); o.ps_col = put32(A.ps_col); o.w_rowptr = put32(A.w_rowptr); o.w_col = put32(A.w_col);
            o.psT_ptr = put64(A.psT_ptr); o.psT_ent = put64(A.psT_ent);
            {
                std::vector<int32_t> ps_row(A.ps_col.size()), w_row(A.w_col.size());
                for (int32_t i = 0; i < A.n; ++i) {
                    for (int32_t k = A.ps_rowptr[i]; k < A.ps_rowptr[(size_t)i + 1]; ++k) ps_row[(size_t)k] = i;
                    for (int32_t k = A.w_rowptr[i]; k < A.w_rowptr[(size_t)i + 1]; ++k) w_row[(size_t)k] = i;
                }
                o.ps_row = put32(ps_row); o.w_row = put32(w_row);
            }
            o.ps_val = take(A.ps_col.size() * 36); o.w_val = take(A.w_col.size() * 36);
            o.t = take((size_t)A.n * 6); o.u = take((size_t)A.n * 6); o.y = take((size_t)A.n * 6); o.zero = take((size_t)A.n * 6);
            {   // explicit transfer operator: index arrays, per-tile block ranges of R^T (this level's tiles) and of R (tiles of consecutive coarse rows), fp32 block arrays
                o.rT_col = put32(A.rT_col); o.rT_of_w = put32(A.rT_of_w); o.ps_of_w = put32(A.ps_of_w);
                std::vector<int32_t> rows;
                for (size_t tt = 0; tt + 1 < A.tile_agg0.size(); ++tt) {
                    const int32_t i0 = A.agg_ptr[A.tile_agg0[tt]], i1 = A.agg_ptr[A.tile_agg0[tt + 1]];
                    for (int li = 0; li < MG_TILE_ROWS; ++li) { const int32_t r = i0 + li; rows.push_back(r < i1 ? A.w_rowptr[r] : 0); rows.push_back(r < i1 ? A.w_rowptr[(size_t)r + 1] : 0); }
                }
                while (pi32.size() % 2) pi32.push_back(0);
                o.rt_rows = put32(rows);
                o.rT_seg_shift = A.rT_seg >= 8 ? 3 : A.rT_seg >= 4 ? 2 : A.rT_seg >= 2 ? 1 : 0;
                const int rpt = MG_TILE_ROWS >> o.rT_seg_shift;
                const int32_t nb0 = Q.share[(size_t)l].rT_row0, nb = Q.share[(size_t)l].rT_row1;      // (several ranks: the restriction's tiles cover the rank's own coarse rows)
                o.rT_tiles = (nb - nb0 + rpt - 1) / rpt;
                rows.clear();
                for (int tt = 0; tt < o.rT_tiles; ++tt)
                    for (int li = 0; li < MG_TILE_ROWS; ++li) { const int32_t r = nb0 + tt * rpt + li; const bool in = li < rpt && r < nb; rows.push_back(in ? A.rT_rowptr[r] : 0); rows.push_back(in ? A.rT_rowptr[(size_t)r + 1] : 0); }
                o.rT_rows = put32(rows);
                o.rt_valf = take((A.w_col.size() * 36 + 1) / 2); o.r_valf = take((A.w_col.size() * 36 + 1) / 2);
            }
        }
    }
    Q.o_plan_send.assign(Q.plans.size(), 0); Q.o_plan_recv.assign(Q.plans.size(), 0);
    for (size_t l = 0; l < Q.plans.size(); ++l) { Q.o_plan_send[l] = put32(Q.plans[l].send_idx); Q.o_plan_recv[l] = put32(Q.plans[l].recv_idx); }
    Q.g0_slots.clear(); Q.o_g0 = 0;
    if (Q.setup.first_whole > 0) {
        const pgo_mg::HostLevel& L1 = H.L[0];
        for (size_t k = 0; k + 1 < L1.g_ptr.size(); ++k) if (L1.g_ptr[k + 1] > L1.g_ptr[k]) Q.g0_slots.push_back((int32_t)k);
        Q.o_g0 = put32(Q.g0_slots);
    }
    Q.o_setup.assign(Q.setup.val.size(), MgPrepared::SetupOff{});
    for (size_t l = 0; l < Q.setup.val.size(); ++l) {
        MgPrepared::SetupOff& so = Q.o_setup[l];
        const pgo_mg::BlockPlan& B = Q.setup.val[l];
        so.val_send = put32(B.x.send_idx); so.val_dst = put32(B.dst); so.val_sum_ptr = put32(B.sum_ptr); so.val_sum_src = put32(B.sum_src);
        if (l < Q.setup.ps.size()) {
            so.ps_send = put32(Q.setup.ps[l].send_idx); so.ps_recv = put32(Q.setup.ps[l].recv_idx); so.rv_send = put32(Q.setup.rv[l].send_idx); so.rv_recv = put32(Q.setup.rv[l].recv_idx);
            so.prod = put32(Q.setup.prod[l]);
        }
    }
    Q.fine = H.fine_smoothed;
    if (Q.fine) {
        const pgo_mg::HostLevel& F = H.F;
        MgPrepared::FineOff& o = Q.fo;
        const int32_t n1 = H.L[0].n;
        o.rowptr = put64(F.rowptr); o.ent = put64(fine_ent); o.col = put32(F.col);
        o.ps_rowptr = put32(F.ps_rowptr); o.ps_col = put32(F.ps_col); o.w_rowptr = put32(F.w_rowptr); o.w_col = put32(F.w_col);
        o.psT_ptr = put64(F.psT_ptr); o.psT_ent = put64(F.psT_ent);
        std::vector<int32_t> ps_row(F.ps_col.size()), w_row(F.w_col.size());
        for (int32_t i = 0; i < F.n; ++i) {
            for (int32_t k = F.ps_rowptr[i]; k < F.ps_rowptr[(size_t)i + 1]; ++k) ps_row[(size_t)k] = i;
            for (int32_t k = F.w_rowptr[i]; k < F.w_rowptr[(size_t)i + 1]; ++k) w_row[(size_t)k] = i;
        }
        o.ps_row = put32(ps_row); o.w_row = put32(w_row);
        // Ps by level-1 row (the restriction r_1 = Ps_0^T r runs as the restriction half of mg_sdown_kernel): position e of psT_ent = slot of the transposed block
        std::vector<int32_t> rT_of_ps(F.ps_col.size()), rT_col(F.ps_col.size());
        for (size_t e = 0; e < F.psT_ent.size(); ++e) { rT_of_ps[(size_t)(F.psT_ent[e] & 0xffffffffll)] = (int32_t)e; rT_col[e] = (int32_t)(F.psT_ent[e] >> 32); }
        o.rT_of_ps = put32(rT_of_ps); o.rT_col = put32(rT_col);
        const double mean_row = (double)F.ps_col.size() / (double)std::max(1, n1);
        int seg = 1;
        while (seg < 8 && mean_row > 5.0 * seg) seg *= 2;
        o.rT_seg_shift = seg >= 8 ? 3 : seg >= 4 ? 2 : seg >= 2 ? 1 : 0;
        const int rpt = MG_TILE_ROWS >> o.rT_seg_shift;
        o.rT_tiles = (n1 + rpt - 1) / rpt;
        std::vector<int32_t> rows;
        rows.reserve((size_t)o.rT_tiles * MG_TILE_ROWS * 2);
        for (int tt = 0; tt < o.rT_tiles; ++tt)
            for (int li = 0; li < MG_TILE_ROWS; ++li) { const int32_t r = tt * rpt + li; const bool in = li < rpt && r < n1; rows.push_back(in ? (int32_t)F.psT_ptr[r] : 0); rows.push_back(in ? (int32_t)F.psT_ptr[(size_t)r + 1] : 0); }
        while (pi32.size() % 2) pi32.push_back(0);
        o.rT_rows = put32(rows);
        o.val = take(F.col.size() * 36); o.Dinv = take((size_t)F.n * 36); o.ps_val = take(F.ps_col.size() * 36); o.w_val = take(F.w_col.size() * 36);
        o.rt_valf = take((F.ps_col.size() * 36 + 1) / 2); o.r_valf = take((F.ps_col.size() * 36 + 1) / 2);
        o.filtered = filtered && want_fine;
        o.lump = o.filtered ? take((size_t)F.n * 36) : 0;
    }
    Q.host_ms = (now_s() - t0) * 1e3;
    if (p->opt.verbosity > 1) std::fprintf(stderr, "[pgo] hierarchy (host): pooled arrays + descriptors      (total %.2f ms, %u hardware threads reported)\n", Q.host_ms, std::thread::hardware_concurrency());
    return PGO_OK;
}

// several ranks: send / receive buffers for the largest exchange of the handle — 42 doubles per row of the keyframes' plan (diagonal block + gradient), 12 per row of a
// level plan (x and r of a level travel together).  Two send buffers: the in-process communicator double-buffers by collective parity.
int ensure_exchange_buffers(pgo_problem* p) {
    if (!p->local_ids) return PGO_OK;
    size_t ns = (size_t)p->fine_plan.x.n_send() * 42, nr = (size_t)p->fine_plan.x.n_recv() * 42;
    for (const pgo_problem::LevelPlanDev& L : p->lvl_plan) if (L.plan) { ns = std::max(ns, (size_t)L.plan->n_send() * 12); nr = std::max(nr, (size_t)L.plan->n_recv() * 12); }
    if (p->mg_first_whole > 0) {      // distributed set-up: 36 doubles per block of the levels, of Ps and per row of Dinv (the level plans), 18 per fp32 block of R
        for (int l = 0; l < p->mg_first_whole && (size_t)l < p->lvl_plan.size(); ++l) if (p->lvl_plan[(size_t)l].plan) { ns = std::max(ns, (size_t)p->lvl_plan[(size_t)l].plan->n_send() * 36); nr = std::max(nr, (size_t)p->lvl_plan[(size_t)l].plan->n_recv() * 36); }
        for (const pgo_mg::BlockPlan& B : p->mg_setup.val) { ns = std::max(ns, (size_t)B.x.n_send() * 36); nr = std::max(nr, (size_t)B.x.n_recv() * 36); }
        for (const pgo_mg::ExchangePlan& X : p->mg_setup.ps) { ns = std::max(ns, (size_t)X.n_send() * 36); nr = std::max(nr, (size_t)X.n_recv() * 36); }
        for (const pgo_mg::ExchangePlan& X : p->mg_setup.rv) { ns = std::max(ns, (size_t)X.n_send() * 18); nr = std::max(nr, (size_t)X.n_recv() * 18); }
    }
    HIPCHK(p, p->d_xsend[0].ensure(ns + 64)); HIPCHK(p, p->d_xsend[1].ensure(ns + 64)); HIPCHK(p, p->d_xrecv.ensure(nr + 64)); HIPCHK(p, p->d_xscal.ensure(16));
    return PGO_OK;
}

// device half: pools (re)allocated, index arrays uploaded, level descriptors filled.  The stream must not be running multigrid kernels of the previous hierarchy.
int mg_install(pgo_problem* p, MgPrepared& Q) {
    const int64_t N = p->N;
    p->coarse_built = false; p->coarse_active = false; p->K = CoarseDev{};
    p->mg_built = false; p->mg_active = false; p->M = MgDev{}; p->mg_geometry_epoch = 0; p->lvl_plan.clear(); p->su_plan.clear(); p->mg_first_whole = 0; p->mg_own.clear();
    if (!Q.ok) { if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] multigrid: the graph does not coarsen (isolated keyframes?) -> off\n"); return PGO_OK; }
    const pgo_mg::Hierarchy& H = Q.H;
    const int nl = (int)H.L.size();
    const std::vector<int32_t>& pi32 = Q.pi32; const std::vector<int64_t>& pi64 = Q.pi64; const size_t nf64 = Q.nf64;
    const int n_top = H.L[nl - 1].n;
    const int nc = (6 * n_top + 63) / 64 * 64;
    HIPCHK(p, p->d_mg_i32.ensure(std::max<size_t>(pi32.size(), 1))); HIPCHK(p, p->d_mg_i64.ensure(std::max<size_t>(pi64.size(), 1))); HIPCHK(p, p->d_mg_f64.ensure(std::max<size_t>(nf64, 2)));
    HIPCHK(p, p->d_cAc.ensure((size_t)nc * nc)); HIPCHK(p, p->d_cAcf.ensure((size_t)nc * nc)); HIPCHK(p, p->d_crc.ensure((size_t)nc * 2)); HIPCHK(p, p->d_cscr.ensure((size_t)nc * 64 + 4096)); HIPCHK(p, p->d_cinfo.ensure(4));
    HIPCHK(p, hipMemcpyAsync(p->d_mg_i32.p, pi32.data(), pi32.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(p->d_mg_i64.p, pi64.data(), pi64.size() * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemsetAsync(p->d_mg_f64.p, 0, nf64 * sizeof(double), p->st));
    HIPCHK(p, hipMemsetAsync(p->d_crc.p, 0, (size_t)nc * 2 * sizeof(double), p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    const int32_t* b32 = p->d_mg_i32.p; const int64_t* b64 = p->d_mg_i64.p; double* bf = p->d_mg_f64.p;
    p->M = MgDev{nl, H.L[0].n, b32 + Q.o_agg0, b32 + Q.o_mem0_ptr, b32 + Q.o_mem0, bf + Q.o_d0, Q.have_tab ? reinterpret_cast<const int4*>(b32 + Q.o_blk_tab) : nullptr, nullptr, Q.a0, Q.a1,
                 Q.setup.first_whole > 0 ? b32 + Q.o_g0 : nullptr, Q.setup.first_whole > 0 ? (int32_t)Q.g0_slots.size() : 0, 0};
    if (p->local_ids) {
        HIPCHK(p, hipMemcpyAsync(bf + Q.o_inv, Q.inv_cnt.data(), Q.inv_cnt.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        p->M.inv_cnt = bf + Q.o_inv;
    }
    p->mg_levels_distributed = 0; p->mg_rows_total = p->mg_rows_own = p->mg_blocks_total = p->mg_blocks_own = 0;
    p->mg_dist.assign((size_t)nl, 0);
    for (int l = 0; l + 1 < nl; ++l) p->mg_dist[(size_t)l] = Q.share[(size_t)l].distributed ? 1 : 0;
    for (int l = 0; l < nl; ++l) {
        const pgo_mg::HostLevel& A = H.L[l];
        const MgPrepared::Off& o = Q.off[l];
        MgLevelDev& D = p->mg_levels[l];
        D = MgLevelDev{};
        D.n = A.n; D.n_next = l + 1 < nl ? H.L[l + 1].n : 0; D.tiles = A.tile_agg0.empty() ? 0 : (int32_t)A.tile_agg0.size() - 1; D.nnzb = (int64_t)A.col.size();
        D.rowptr = b64 + o.rowptr; D.col = b32 + o.col; D.val = bf + o.val; D.g_ptr = b64 + o.g_ptr; D.g_ent = b64 + o.g_ent;
        D.Dinv = bf + o.Dinv; D.pos = bf + o.pos; D.d = bf + o.d; D.parent = b32 + o.parent; D.agg_ptr = b32 + o.agg_ptr; D.tile_info = reinterpret_cast<const int4*>(b32 + o.tile); D.tile_rows = reinterpret_cast<const int2*>(b32 + o.tile_rows);
        D.r = bf + o.r; D.x = bf + o.x; D.xt = bf + o.xt; D.xf = bf + o.xf; D.valf = reinterpret_cast<float*>(bf + o.valf);
        D.row_of = b32 + o.row_of; D.tr_of = b32 + o.tr_of;
        D.seg_shift = A.seg >= 8 ? 3 : A.seg >= 4 ? 2 : A.seg >= 2 ? 1 : 0;
        D.pad3_ = l;      // (the level's index: read by the timeline variant build only)
        {   // what the rank's cycle kernels read of the level (pgo_mg_level_norms: the same ranges whichever way the set-up ran)
            pgo_problem::OwnRange R;
            const bool mine = l + 1 < nl && Q.share[(size_t)l].distributed;
            const int32_t r0 = mine ? A.own_ptr[(size_t)p->rank] : 0, r1 = mine ? A.own_ptr[(size_t)p->rank + 1] : A.n;
            R.row0 = r0; R.row1 = r1; R.blk0 = A.rowptr[(size_t)r0]; R.blk1 = A.rowptr[(size_t)r1];
            if (A.smoothed) {
                R.ps0 = A.ps_rowptr[(size_t)r0]; R.ps1 = A.ps_rowptr[(size_t)r1]; R.w0 = A.w_rowptr[(size_t)r0]; R.w1 = A.w_rowptr[(size_t)r1];
                R.rT0 = A.rT_rowptr[(size_t)Q.share[(size_t)l].rT_row0]; R.rT1 = A.rT_rowptr[(size_t)Q.share[(size_t)l].rT_row1];
            }
            p->mg_own.push_back(R);
        }
        {   // the set-up's share of the level: a distributed level under the distributed set-up forms its own rows, every other one all of them
            const bool part = l < Q.setup.first_whole;
            const int32_t r0 = part ? A.own_ptr[(size_t)p->rank] : 0, r1 = part ? A.own_ptr[(size_t)p->rank + 1] : A.n;
            D.su_row0 = r0; D.su_row1 = r1; D.su_blk0 = A.rowptr[(size_t)r0]; D.su_blk1 = A.rowptr[(size_t)r1];
            D.su_ps0 = A.smoothed ? A.ps_rowptr[(size_t)r0] : 0; D.su_ps1 = A.smoothed ? A.ps_rowptr[(size_t)r1] : 0;
            D.su_w0 = A.smoothed ? A.w_rowptr[(size_t)r0] : 0; D.su_w1 = A.smoothed ? A.w_rowptr[(size_t)r1] : 0;
            D.su_prod = nullptr; D.n_su_prod = 0;
            if (part && A.smoothed) { D.su_prod = b32 + Q.o_setup[(size_t)l].prod; D.n_su_prod = (int32_t)Q.setup.prod[(size_t)l].size(); }
        }
        {   // the cycle's share of the level (several ranks: the owner's rows; else all of it)
            const MgPrepared::Share& sh = Q.share[(size_t)l];
            D.tile0 = sh.tile0; D.tiles_own = l + 1 < nl ? sh.tiles_own : 0; D.rT_row0 = sh.rT_row0; D.rT_row1 = l + 1 < nl ? sh.rT_row1 : 0;
            if (l + 1 < nl) {
                if (sh.distributed) ++p->mg_levels_distributed;
                const int64_t blocks = (int64_t)A.col.size() + (A.smoothed ? 2 * (int64_t)A.w_col.size() : 0);
                int64_t own_rows = A.n, own_blocks = blocks;
                if (sh.distributed) {
                    const int32_t r0 = A.own_ptr[(size_t)p->rank], r1 = A.own_ptr[(size_t)p->rank + 1];
                    own_rows = r1 - r0;
                    own_blocks = A.rowptr[(size_t)r1] - A.rowptr[(size_t)r0];
                    if (A.smoothed) own_blocks += (int64_t)(A.w_rowptr[(size_t)r1] - A.w_rowptr[(size_t)r0]) + (int64_t)(A.rT_rowptr[(size_t)sh.rT_row1] - A.rT_rowptr[(size_t)sh.rT_row0]);
                }
                p->mg_rows_total += A.n; p->mg_rows_own += own_rows; p->mg_blocks_total += blocks; p->mg_blocks_own += own_blocks;
            }
        }
        if (A.smoothed) {
            D.smoothed = 1; D.n_ps = (int32_t)A.ps_col.size(); D.n_w = (int32_t)A.w_col.size();
            D.ps_rowptr = b32 + o.ps_rowptr; D.ps_col = b32 + o.ps_col; D.w_rowptr = b32 + o.w_rowptr; D.w_col = b32 + o.w_col; D.psT_ptr = b64 + o.psT_ptr; D.psT_ent = b64 + o.psT_ent;
            D.ps_row = b32 + o.ps_row; D.w_row = b32 + o.w_row;
            D.ps_val = bf + o.ps_val; D.w_val = bf + o.w_val; D.t = bf + o.t; D.u = bf + o.u; D.y = bf + o.y; D.zero = bf + o.zero;
            if (p->opt.mg_explicit_transfer != 0 || p->local_ids) {      // (several ranks: always the explicit form — the implicit one would need two more exchanges per level)
                D.rt_valf = reinterpret_cast<float*>(bf + o.rt_valf); D.r_valf = reinterpret_cast<float*>(bf + o.r_valf);
                D.rt_rows = reinterpret_cast<const int2*>(b32 + o.rt_rows); D.rT_rows = reinterpret_cast<const int2*>(b32 + o.rT_rows);
                D.rT_col = b32 + o.rT_col; D.rT_of_w = b32 + o.rT_of_w; D.ps_of_w = b32 + o.ps_of_w;
                D.rT_tiles = o.rT_tiles; D.rT_seg_shift = o.rT_seg_shift;
            }
        }
    }
    p->mg_fine = Q.fine; p->mg_fineF = MgLevelDev{}; p->mg_fineT = MgLevelDev{};
    if (Q.fine) {
        const pgo_mg::HostLevel& Fh = H.F;
        const MgPrepared::FineOff& o = Q.fo;
        MgLevelDev& F = p->mg_fineF;
        F.n = Fh.n; F.n_next = H.L[0].n; F.tiles = 0; F.nnzb = (int64_t)Fh.col.size();
        F.tile0 = 0; F.tiles_own = 0; F.rT_row0 = 0; F.rT_row1 = H.L[0].n;      // (one GPU: the restriction covers every level-1 row)
        F.su_row0 = 0; F.su_row1 = Fh.n; F.su_blk0 = 0; F.su_blk1 = (int64_t)Fh.col.size(); F.su_ps0 = 0; F.su_ps1 = (int32_t)Fh.ps_col.size(); F.su_w0 = 0; F.su_w1 = (int32_t)Fh.w_col.size(); F.su_prod = nullptr; F.n_su_prod = 0;
        F.dlump = o.filtered ? bf + o.lump : nullptr;
        F.rowptr = b64 + o.rowptr; F.col = b32 + o.col; F.val = bf + o.val; F.g_ent = b64 + o.ent; F.Dinv = bf + o.Dinv;
        F.d = p->M.d0; F.parent = p->M.agg0;
        F.smoothed = 1; F.n_ps = (int32_t)Fh.ps_col.size(); F.n_w = (int32_t)Fh.w_col.size();
        F.ps_rowptr = b32 + o.ps_rowptr; F.ps_col = b32 + o.ps_col; F.w_rowptr = b32 + o.w_rowptr; F.w_col = b32 + o.w_col; F.psT_ptr = b64 + o.psT_ptr; F.psT_ent = b64 + o.psT_ent;
        F.ps_row = b32 + o.ps_row; F.w_row = b32 + o.w_row; F.ps_val = bf + o.ps_val; F.w_val = bf + o.w_val;
        // the transfer view: the explicit operator's fields describe Ps itself (its own pattern by keyframe row; by level-1 row for the restriction)
        MgLevelDev& T = p->mg_fineT;
        T = F;
        T.rt_valf = reinterpret_cast<float*>(bf + o.rt_valf); T.r_valf = reinterpret_cast<float*>(bf + o.r_valf);
        T.rT_of_w = b32 + o.rT_of_ps; T.rT_col = b32 + o.rT_col; T.rT_rows = reinterpret_cast<const int2*>(b32 + o.rT_rows);
        T.rT_tiles = o.rT_tiles; T.rT_seg_shift = o.rT_seg_shift;
    }
    // the dense coarsest level shares the buffers of the two-level preconditioner, which the multigrid replaces on this graph
    p->K.n_agg = n_top; p->K.nc = nc; p->K.Ac = p->d_cAc.p; p->K.Acf = p->d_cAcf.p; p->K.rc = p->d_crc.p; p->K.yc = p->d_crc.p + nc;
    p->mg_built = true;
    p->mg_sw_built.swap(Q.sw_built);
    if (p->opt.verbosity > 0) {
        std::fprintf(stderr, "[pgo] multigrid: %lld keyframes", (long long)N);
        for (int l = 0; l < nl; ++l) {
            int64_t longest = 0;
            for (int32_t i = 0; i < H.L[l].n; ++i) longest = std::max<int64_t>(longest, H.L[l].rowptr[(size_t)i + 1] - H.L[l].rowptr[(size_t)i]);
            std::fprintf(stderr, " -> %d (%lld blocks, longest row %lld%s)", H.L[l].n, (long long)H.L[l].col.size(), (long long)longest, H.L[l].smoothed ? ", smoothed prolongator above" : "");
        }
        std::fprintf(stderr, ", coarsest dense %d (host %.1f ms)\n", nc, Q.host_ms);
    }
    // several ranks: the level exchanges' index lists live in the int32 pool; the segment bounds stay on the host (p->mg_plans)
    p->mg_plans.swap(Q.plans);
    p->lvl_plan.assign(p->mg_plans.size(), pgo_problem::LevelPlanDev{});
    for (size_t l = 0; l < p->mg_plans.size(); ++l) p->lvl_plan[l] = pgo_problem::LevelPlanDev{b32 + Q.o_plan_send[l], b32 + Q.o_plan_recv[l], &p->mg_plans[l]};
    p->mg_first_whole = Q.setup.first_whole;
    if (p->mg_first_whole > 0) {
        const pgo_mg::HostLevel& W = H.L[(size_t)p->mg_first_whole];
        p->mg_fw_row0 = W.own_ptr[(size_t)p->rank]; p->mg_fw_row1 = W.own_ptr[(size_t)p->rank + 1];
        p->mg_fw_blk0 = W.rowptr[(size_t)p->mg_fw_row0]; p->mg_fw_blk1 = W.rowptr[(size_t)p->mg_fw_row1];
    }
    p->mg_setup = std::move(Q.setup);
    p->su_plan.assign(Q.o_setup.size(), pgo_problem::SetupPlanDev{});
    for (size_t l = 0; l < Q.o_setup.size(); ++l) {
        const MgPrepared::SetupOff& so = Q.o_setup[l];
        p->su_plan[l] = pgo_problem::SetupPlanDev{b32 + so.val_send, b32 + so.val_dst, b32 + so.val_sum_ptr, b32 + so.val_sum_src, b32 + so.ps_send, b32 + so.ps_recv, b32 + so.rv_send, b32 + so.rv_recv};
    }
    int rcx;
    if ((rcx = ensure_exchange_buffers(p)) != PGO_OK) return rcx;
    return PGO_OK;
}

// a regroup in flight is waited for and dropped (its result belongs to a solve state that is gone, or the graph is about to change)
void mg_job_cancel(pgo_problem* p) {
    if (p->mg_job.joinable()) p->mg_job.join();
    p->mg_job_running = false; p->mg_job_out.reset();
}

// the pending hierarchy of a fresh graph build is not wanted any more (the graph is about to change, the handle to go): the worker is waited for, its result dropped, and the
// graph marked for a rebuild (mg_built was an announcement, not a fact)
void mg_init_drop(pgo_problem* p) {
    if (p->mg_init_thread.joinable()) p->mg_init_thread.join();
    if (p->mg_init_pending) { p->mg_init_pending = false; p->mg_init_out.reset(); p->mg_built = false; p->graph_dirty = true; }
}
int build_multigrid(pgo_problem* p, const double* sw_now, MgPrepared* ready);
int build_two_level_aggregates(pgo_problem* p);
// ... or it is needed now: waited for and installed
int mg_init_finish(pgo_problem* p) {
    if (!p->mg_init_pending) return PGO_OK;
    const double t0 = now_s();
    if (p->mg_init_thread.joinable()) p->mg_init_thread.join();
    p->mg_init_pending = false;
    std::unique_ptr<MgPrepared> Q = std::move(p->mg_init_out);
    if (p->rc_mg_init != PGO_OK || !Q) {      // the worker failed: the handle keeps working with what a graph without a hierarchy gets (build_graph's synchronous path does the same)
        p->mg_built = false;
        const int rc_worker = p->rc_mg_init != PGO_OK ? p->rc_mg_init : PGO_ERR_STATE;
        HIPCHK(p, hipStreamSynchronize(p->st));
        const int rc2 = build_two_level_aggregates(p);
        ++p->build_epoch;
        return rc2 != PGO_OK ? rc2 : rc_worker;
    }
    const double waited = (now_s() - t0) * 1e3;
    int rc;
    HIPCHK(p, hipStreamSynchronize(p->st));
    if ((rc = build_multigrid(p, nullptr, Q.get())) != PGO_OK) return rc;
    // a hierarchy that does not coarsen: the graph falls back to the two-level method — exactly what build_graph's synchronous path (several ranks) gives the same graph
    if (!p->mg_built && (rc = build_two_level_aggregates(p)) != PGO_OK) return rc;
    ++p->build_epoch;
    if (p->opt.verbosity > 1) std::fprintf(stderr, "[pgo] multigrid: hierarchy of the new graph installed at its first use: host half %.2f ms on a worker thread, waited %.2f ms, installed in %.2f ms%s\n",
                                           Q->host_ms, waited, (now_s() - t0) * 1e3 - waited, p->mg_built ? "" : " — it does not coarsen: the two-level method on this graph");
    p->mg_job_old = std::move(Q);      // (freed off the solve's critical path: regroup_install's note on munmap and the GPU's address space)
    return PGO_OK;
}

// The aggregation multigrid's hierarchy for the graph of this handle and the given switch values (host array over the caller's switches, or null): host-side structure
// (pgo_mg_host.hpp), pooled device arrays, level descriptors.  Called by build_graph, and again when the switch values have moved far from the ones the hierarchy was
// built with (regroup): the levels above level 1 are matched along the couplings that are alive NOW.  p->mg_cache keeps what does not depend on the switches.
// the caller's keyframe and switch counts decide (the same answer on every rank): graphs with switchable loop closures — all of the reference's — take the multigrid from
// mg_min_keyframes_switchable on, graphs without from mg_min_keyframes; mg_min_keyframes = 0 turns it off altogether
bool wants_multigrid(const pgo_problem* p) {
    int64_t mg_from = p->opt.mg_min_keyframes;
    if (mg_from > 0 && p->S > 0 && p->opt.mg_min_keyframes_switchable > 0) mg_from = std::min<int64_t>(mg_from, p->opt.mg_min_keyframes_switchable);
    return mg_from > 0 && p->N_global >= mg_from;
}
// `ready`: the host half prepared beforehand (build_graph runs it on a worker thread beside its own host work and uploads)
int build_multigrid(pgo_problem* p, const double* sw_now, MgPrepared* ready) {
    int rc;
    mg_job_cancel(p);
    p->coarse_built = false; p->coarse_active = false; p->K = CoarseDev{};
    p->mg_built = false; p->mg_active = false; p->M = MgDev{}; p->mg_geometry_epoch = 0; p->lvl_plan.clear(); p->su_plan.clear(); p->mg_first_whole = 0; p->mg_own.clear();
    if (wants_multigrid(p)) {
        MgPrepared Q;
        if (!ready && (rc = mg_prepare(p, sw_now, Q)) != PGO_OK) return rc;
        if ((rc = mg_install(p, ready ? *ready : Q)) != PGO_OK) return rc;
    }
    if (p->local_ids && !p->mg_built) { p->lvl_plan.clear(); if ((rc = ensure_exchange_buffers(p)) != PGO_OK) return rc; }
    return PGO_OK;
}

// The two-level method's aggregates (consecutive keyframes) and the contribution lists of its dense coarse operator, for the graph as built: what a graph WITHOUT a multigrid
// hierarchy preconditions with.  Called by build_graph, and by mg_init_finish when the hierarchy a worker thread prepared turns out not to coarsen (the synchronous path —
// several ranks — decides that inside build_graph; one GPU only learns it where the hierarchy is first needed: both end up with the same preconditioner).
int build_two_level_aggregates(pgo_problem* p) {
    const int64_t N = p->N, Er = p->rel.size(), Es = p->swe.size();
    const int32_t* g2l = p->local_ids ? p->g2l.data() : nullptr;
    auto L = [g2l](int32_t g) -> int32_t { return g2l ? g2l[g] : g; };
    int n_agg = p->opt.coarse_aggregates;
    // a graph with no more keyframes than `half` (256 by default) gets one aggregate per keyframe: the coarse operator IS the reduced system and the "preconditioner"
    // its dense inverse (a direct solve; the PCG around it only refines); larger graphs: at least 8 keyframes per aggregate, but not fewer than `half` aggregates — the
    // dense inverse (cubic in the aggregates) is what small graphs pay for (scripts/gpu_small_graphs.py) — and at most coarse_aggregates (768: measured on
    // chain-like session graphs of 6 000 - 23 000 keyframes, scripts/gpu_session_aggregates.py: 768 beats 512 by 3 - 45 %, 1024 and 1536 lose to the cubic inverse)
    const int half = std::min(n_agg / 2, 256);
    if (N <= half) n_agg = (int)N;
    else n_agg = (int)std::min<int64_t>(n_agg, std::max<int64_t>(N / 8, half));
    if (n_agg >= 2 && !p->local_ids && (N + n_agg - 1) / n_agg <= 1024) {    // aggregates of thousands of keyframes are never used (build_coarse)
        const int m = (int)((N + n_agg - 1) / n_agg);
        n_agg = (int)((N + m - 1) / m);
        std::vector<int32_t> agg_free((size_t)n_agg, 0);
        for (int64_t n = 0; n < N; ++n) if (p->h_node_free[n]) agg_free[n / m]++;
        // (block key, entry) pairs; key = a * n_agg + b with a <= b
        std::vector<std::pair<int64_t, int64_t>> ent;
        ent.reserve((size_t)N + 2 * (size_t)(Er + Es));
        for (int64_t n = 0; n < N; ++n) if (p->h_node_free[n]) ent.push_back({(int64_t)(n / m) * n_agg + n / m, (n << 3) | 0});
        auto edge = [&](int64_t e, int32_t c1, int32_t c2, int kind_fwd) {
            if (!p->h_node_free[c1] || !p->h_node_free[c2]) return;       // rows and columns of fixed keyframes are not part of the system
            const int64_t a = c1 / m, b = c2 / m;
            if (a < b) ent.push_back({a * n_agg + b, (e << 3) | kind_fwd});
            else if (a > b) ent.push_back({b * n_agg + a, (e << 3) | (kind_fwd + 1)});
            else { ent.push_back({a * n_agg + a, (e << 3) | kind_fwd}); ent.push_back({a * n_agg + a, (e << 3) | (kind_fwd + 1)}); }
        };
        for (int64_t e = 0; e < Er; ++e) edge(e, L(p->rel.c1[e]), L(p->rel.c2[e]), 1);
        for (int64_t e = 0; e < Es; ++e) edge(e, L(p->swe.c1[e]), L(p->swe.c2[e]), 3);
        for (int a = 0; a < n_agg; ++a) if (agg_free[a] == 0) ent.push_back({(int64_t)a * n_agg + a, -1});   // identity block: listed, no contribution
        std::stable_sort(ent.begin(), ent.end(), [](const std::pair<int64_t, int64_t>& x, const std::pair<int64_t, int64_t>& y) { return x.first < y.first; });
        std::vector<int64_t> blk_ptr, contrib;
        std::vector<int32_t> blk_ab;
        int64_t prev = -1;
        for (const auto& kv : ent) {
            if (kv.first != prev) { blk_ptr.push_back((int64_t)contrib.size()); blk_ab.push_back((int32_t)(kv.first / n_agg)); blk_ab.push_back((int32_t)(kv.first % n_agg)); prev = kv.first; }
            if (kv.second >= 0) contrib.push_back(kv.second);
        }
        blk_ptr.push_back((int64_t)contrib.size());
        const int n_blk = (int)blk_ab.size() / 2;
        const int nc = (6 * n_agg + 63) / 64 * 64;      // padded with a decoupled identity block (the dense kernels work on 64-wide tiles)
        HIPCHK(p, p->d_ccen.ensure((size_t)n_agg * 3)); HIPCHK(p, p->d_cd.ensure((size_t)N * 3)); HIPCHK(p, p->d_cAc.ensure((size_t)nc * nc)); HIPCHK(p, p->d_cAcf.ensure((size_t)nc * nc));
        HIPCHK(p, p->d_crc.ensure((size_t)nc * 2)); HIPCHK(p, p->d_cscr.ensure((size_t)nc * 64 + 4096)); HIPCHK(p, hipMemsetAsync(p->d_crc.p, 0, (size_t)nc * 2 * sizeof(double), p->st)); HIPCHK(p, p->d_cblk_ptr.ensure(blk_ptr.size())); HIPCHK(p, p->d_ccontrib.ensure(std::max<size_t>(contrib.size(), 1)));
        HIPCHK(p, p->d_cblk_ab.ensure(blk_ab.size())); HIPCHK(p, p->d_cagg_free.ensure(n_agg)); HIPCHK(p, p->d_cinfo.ensure(4));
        HIPCHK(p, hipMemcpyAsync(p->d_cblk_ptr.p, blk_ptr.data(), blk_ptr.size() * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
        if (!contrib.empty()) HIPCHK(p, hipMemcpyAsync(p->d_ccontrib.p, contrib.data(), contrib.size() * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_cblk_ab.p, blk_ab.data(), blk_ab.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_cagg_free.p, agg_free.data(), n_agg * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        p->K = CoarseDev{n_agg, nc, m, n_blk, p->d_ccen.p, p->d_cd.p, p->d_cAc.p, p->d_crc.p, p->d_crc.p + nc, p->d_cblk_ptr.p, p->d_cblk_ab.p, p->d_ccontrib.p, p->d_cagg_free.p, p->d_cAcf.p};
        p->coarse_built = true;
    }
    return PGO_OK;
}

int build_graph(pgo_problem* p, int64_t N, int64_t S, const double* sw_now) {
    mg_job_cancel(p);      // (a regroup's worker reads the host arrays rebuilt below)
    mg_init_drop(p);
    double t_phase = now_s();
    auto phase = [&](const char* what) { if (p->opt.verbosity > 1) { const double t = now_s(); std::fprintf(stderr, "[pgo] build_graph: %-34s %7.2f ms\n", what, (t - t_phase) * 1e3); t_phase = t; } };
    // ---- validate against the array sizes the caller solves with
    for (const HostClass* H : {&p->rel, &p->swe})
        for (int64_t e = 0; e < H->size(); ++e)
            if (H->c1[e] < 0 || H->c1[e] >= N || H->c2[e] < 0 || H->c2[e] >= N) { p->err = "edge endpoint out of range for n_nodes"; return PGO_ERR_INVALID_ARG; }
    p->h_sw_used.assign((size_t)S, 0);
    for (int64_t e = 0; e < p->swe.size(); ++e) {
        const int32_t si = p->swe.sw[e];
        if (si < 0 || si >= S) { p->err = "switch index out of range for n_switch"; return PGO_ERR_INVALID_ARG; }
        if (p->h_sw_used[si]) { p->err = "switch index used by more than one edge"; return PGO_ERR_INVALID_ARG; }
        p->h_sw_used[si] = 1;
    }
    for (const PriorDev& pr : p->priors) if (pr.node < 0 || pr.node >= N) { p->err = "regulariser node out of range"; return PGO_ERR_INVALID_ARG; }
    p->S = S; p->N_global = N;
    GraphDev& G = p->G;
    G = GraphDev{};
    // ---- multi-GPU: rank-local subgraph.  This rank works on the keyframes its own residual blocks touch, renumbered densely; keyframes
    // touched by >= 2 ranks are "shared" (their rows are summed over ranks by exchange_rows), the lowest touching rank is the owner.
    const int64_t Ng = N;
    p->local_ids = p->comm != nullptr || p->custom_allreduce != nullptr || p->local_group != nullptr;   // also with a 1-rank communicator: the same code path, every collective issued
    p->n_sh_mine = p->n_sh_global = 0;
    if (p->local_ids) {
        std::vector<uint8_t> touched((size_t)Ng, 0);
        std::vector<int32_t> deg((size_t)Ng, 0);      // residual blocks of THIS rank on each keyframe
        for (const HostClass* H : {&p->rel, &p->swe}) for (int64_t e = 0; e < H->size(); ++e) { touched[H->c1[e]] = 1; touched[H->c2[e]] = 1; ++deg[H->c1[e]]; ++deg[H->c2[e]]; }
        for (const PriorDev& pr : p->priors) { touched[pr.node] = 1; ++deg[pr.node]; }
        bool any = false;
        for (int64_t g = 0; g < Ng && !any; ++g) any = touched[g] != 0;
        if (!any) touched[0] = 1;   // a rank without residual blocks still takes part in every collective: give it one (zero-contribution) keyframe
        // Two all-reduces of Ng doubles, once per graph build.  Sum: every rank adds 2^rank for the keyframes it touches — the set of touching ranks (exact in a double up to
        // 52 ranks): how many they are, and who exchanges the keyframe's rows with whom.  Max of (blocks + 1) * 64 + 63 - rank: the OWNER — the rank holding most of the
        // keyframe's residual blocks, the lowest of them on a tie (pgo_mg_host.hpp: Owners).
        if (p->world > 52) { p->err = "more than 52 ranks"; return PGO_ERR_INVALID_ARG; }
        std::vector<double> buf((size_t)Ng), obuf((size_t)Ng);
        for (int64_t g = 0; g < Ng; ++g) { buf[g] = touched[g] ? std::ldexp(1.0, p->rank) : 0.0; obuf[g] = touched[g] ? (double)(((int64_t)deg[g] + 1) * 64 + 63 - p->rank) : 0.0; }
        int rc2;
        if ((rc2 = host_allreduce(p, buf, 0)) != PGO_OK) return rc2;
        if ((rc2 = host_allreduce(p, obuf, 2)) != PGO_OK) return rc2;
        p->h_touch_mask.assign((size_t)Ng, 0); p->h_owner.assign((size_t)Ng, -1);
        p->l2g.clear(); p->g2l.assign((size_t)Ng, -1); p->h_own.clear(); p->h_touched_any.assign((size_t)Ng, 0);
        int64_t pos = 0, n_mine = 0;
        for (int64_t g = 0; g < Ng; ++g) {
            const uint64_t m = (uint64_t)(buf[g] + 0.5);
            p->h_touch_mask[g] = m;
            const int cnt = __builtin_popcountll(m);
            if (m) { p->h_owner[g] = 63 - (int32_t)((int64_t)(obuf[g] + 0.5) % 64); if (!((m >> p->h_owner[g]) & 1)) { p->err = "graph build: a keyframe's owner does not touch it (the ranks' all-reduces disagree)"; return PGO_ERR_COMM; } }
            p->h_touched_any[g] = cnt > 0;
            if (touched[g]) {
                if (!((m >> p->rank) & 1)) { p->err = "touch masks: the all-reduce did not return this rank's own bit"; return PGO_ERR_COMM; }
                p->g2l[g] = (int32_t)p->l2g.size();
                if (cnt >= 2) ++n_mine;
                p->l2g.push_back((int32_t)g);
                p->h_own.push_back(p->h_owner[g] == p->rank ? 1.0 : 0.0);
            }
            if (cnt >= 2) ++pos;
        }
        p->n_sh_global = pos; p->n_sh_mine = n_mine;
        N = (int64_t)p->l2g.size();
        HIPCHK(p, p->d_l2g.ensure(N)); HIPCHK(p, p->d_own.ensure(N));
        HIPCHK(p, hipMemcpyAsync(p->d_l2g.p, p->l2g.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_own.p, p->h_own.data(), N * sizeof(double), hipMemcpyHostToDevice, p->st));
        {   // the keyframes' neighbour exchange: segments per peer, and for every shared keyframe the order its parts are summed in (pgo_mg_host.hpp: build_fine_plan)
            pgo_mg::build_fine_plan(p->h_touch_mask, p->l2g, p->rank, p->world, p->fine_plan);
            const pgo_mg::FinePlan& F = p->fine_plan;
            auto up = [&](DBuf<int32_t>& d, const std::vector<int32_t>& v) -> int {
                HIPCHK(p, d.ensure(std::max<size_t>(v.size(), 1)));
                if (!v.empty()) HIPCHK(p, hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
                return PGO_OK;
            };
            if ((rc2 = up(p->d_fp_send, F.x.send_idx)) != PGO_OK || (rc2 = up(p->d_fp_shloc, F.sh_loc)) != PGO_OK || (rc2 = up(p->d_fp_sumptr, F.sum_ptr)) != PGO_OK || (rc2 = up(p->d_fp_sumsrc, F.sum_src)) != PGO_OK) return rc2;
            p->lvl_plan.clear();
            if ((rc2 = ensure_exchange_buffers(p)) != PGO_OK) return rc2;
        }
        HIPCHK(p, hipStreamSynchronize(p->st));
        G.own = p->d_own.p;
    } else {
        p->l2g.clear(); p->g2l.clear(); p->h_own.clear(); p->h_touched_any.clear();
        G.own = nullptr;
    }
    p->N = N;
    phase("validation, rank-local numbering");
    const int32_t* g2l = p->local_ids ? p->g2l.data() : nullptr;
    auto L = [g2l](int32_t g) -> int32_t { return g2l ? g2l[g] : g; };
    G.N = N; G.S = S;
    int rc;
    // a keyframe is part of the program when a residual block touches it: on one GPU that is a non-empty incident list; in a rank-local
    // subgraph every keyframe is touched by construction (by this rank or, for the stand-in keyframe of an idle rank, possibly by none)
    p->h_node_free.assign((size_t)N, 0);
    {
        std::vector<uint8_t> touched_here((size_t)N, 0);      // (= a non-empty incident list, known before the lists are built: the hierarchy worker below starts at once)
        for (const HostClass* H : {&p->rel, &p->swe}) for (int64_t e = 0; e < H->size(); ++e) { touched_here[L(H->c1[e])] = 1; touched_here[L(H->c2[e])] = 1; }
        for (const PriorDev& pr : p->priors) touched_here[L(pr.node)] = 1;
        for (int64_t n = 0; n < N; ++n) p->h_node_free[n] = (touched_here[n] || (p->local_ids && p->h_touched_any[p->l2g[n]])) ? 1 : 0;
    }
    for (int32_t c : p->constant_nodes) if (c >= 0 && c < Ng && L(c) >= 0) p->h_node_free[L(c)] = 0;
    // One GPU: the HOST half of the multigrid hierarchy (pgo_mg_host.hpp: ~0.1 s for C3, single-threaded sorts and matchings) needs the edge lists and the free flags
    // only, so it runs on a worker thread beside the rest of this function — incident-list upload, matrix-free tile packing, buffer allocation — and is installed where
    // build_multigrid used to compute it.  Nothing here depends on timing: the result is the same hierarchy.  (Several ranks: its host half holds collectives.)
    struct Guard { pgo_problem* p; bool committed = false; ~Guard() { if (!committed) mg_init_drop(p); } } mg_guard{p};     // an early return below waits for the worker and drops its result
    p->mg_cache.valid = false; p->mg_fine_auto = -1;
    bool mg_async = false;
    if (!p->local_ids && wants_multigrid(p)) {
        p->mg_init_out.reset(new MgPrepared());
        p->rc_mg_init = PGO_OK; p->mg_init_pending = true; mg_async = true;
        MgPrepared* Qp = p->mg_init_out.get();
        std::vector<double> sw_copy;      // the caller's switch array is only guaranteed to live as long as this call
        if (sw_now && S > 0) sw_copy.assign(sw_now, sw_now + S);
        try { p->mg_init_thread = std::thread([p, sw_copy, Qp]() { p->rc_mg_init = mg_prepare(p, sw_copy.empty() ? nullptr : sw_copy.data(), *Qp); }); }
        catch (...) { p->rc_mg_init = mg_prepare(p, sw_now, *Qp); }
    }
    if ((rc = upload_class(p, p->rel, false, p->d_rc1, p->d_rc2, p->d_sidx /*unused*/, p->d_rmeas, p->d_rwin, G.rel)) != PGO_OK) return rc;
    if ((rc = upload_class(p, p->swe, true, p->d_sc1, p->d_sc2, p->d_sidx, p->d_smeas, p->d_swin, G.sw)) != PGO_OK) return rc;
    const int64_t Er = G.rel.E, Es = G.sw.E, Eg = (int64_t)p->priors.size();
    phase("edge classes packed + uploaded");
    std::vector<PriorDev> pri = p->priors;
    for (PriorDev& x : pri) x.node = L(x.node);
    // ---- node -> incident list (edges in slot order, then regularisers), BSR structure
    std::vector<int64_t> rowptr(N + 1, 0), bsr_rowptr(N + 1, 0);
    for (int64_t e = 0; e < Er; ++e) { rowptr[L(p->rel.c1[e]) + 1]++; rowptr[L(p->rel.c2[e]) + 1]++; }
    for (int64_t e = 0; e < Es; ++e) { rowptr[L(p->swe.c1[e]) + 1]++; rowptr[L(p->swe.c2[e]) + 1]++; }
    for (int64_t n = 0; n < N; ++n) bsr_rowptr[n + 1] = bsr_rowptr[n] + 1 + rowptr[n + 1];
    for (int64_t k = 0; k < Eg; ++k) rowptr[pri[k].node + 1]++;
    for (int64_t n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
    const int64_t ninc = rowptr[N];
    p->nnzb = bsr_rowptr[N];
    std::vector<int64_t> inc((size_t)ninc), fill(rowptr.begin(), rowptr.end() - 1);
    std::vector<int32_t> bsr_col((size_t)p->nnzb);
    std::vector<int64_t> bfill(N);
    for (int64_t n = 0; n < N; ++n) { bsr_col[bsr_rowptr[n]] = (int32_t)n; bfill[n] = bsr_rowptr[n] + 1; }
    auto add_edge = [&](int64_t slot, int32_t a, int32_t b) {
        inc[fill[a]++] = (slot << 1) | 0; bsr_col[bfill[a]++] = b;
        inc[fill[b]++] = (slot << 1) | 1; bsr_col[bfill[b]++] = a;
    };
    for (int64_t e = 0; e < Er; ++e) add_edge(e, L(p->rel.c1[e]), L(p->rel.c2[e]));
    for (int64_t e = 0; e < Es; ++e) add_edge(G.rel.Epad + e, L(p->swe.c1[e]), L(p->swe.c2[e]));
    for (int64_t k = 0; k < Eg; ++k) inc[fill[pri[k].node]++] = ((G.rel.Epad + G.sw.Epad + k) << 1);

    HIPCHK(p, p->d_inc_rowptr.ensure(N + 1)); HIPCHK(p, p->d_bsr_rowptr.ensure(N + 1)); HIPCHK(p, p->d_inc.ensure(std::max<int64_t>(ninc, 1)));
    HIPCHK(p, p->d_bsr_col.ensure(std::max<int64_t>(p->nnzb, 1))); HIPCHK(p, p->d_node_free.ensure(std::max<int64_t>(N, 1)));
    HIPCHK(p, p->d_prior.ensure(std::max<int64_t>(Eg, 1)));
    HIPCHK(p, hipMemcpyAsync(p->d_inc_rowptr.p, rowptr.data(), (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(p->d_bsr_rowptr.p, bsr_rowptr.data(), (N + 1) * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
    if (ninc) HIPCHK(p, hipMemcpyAsync(p->d_inc.p, inc.data(), ninc * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
    if (p->nnzb) HIPCHK(p, hipMemcpyAsync(p->d_bsr_col.p, bsr_col.data(), p->nnzb * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    if (N) HIPCHK(p, hipMemcpyAsync(p->d_node_free.p, p->h_node_free.data(), N, hipMemcpyHostToDevice, p->st));
    if (Eg) HIPCHK(p, hipMemcpyAsync(p->d_prior.p, pri.data(), Eg * sizeof(PriorDev), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));

    phase("incident lists + block-CSR structure");
    // ---- matrix-free operator: edge-sides in keyframe-major order, packed into workgroup tiles of whole keyframes
    bool mf = p->opt.linear_solver == PGO_LINEAR_PCG_MATRIX_FREE;
    if (mf) {
        // a keyframe with more edge sides than a workgroup tile holds (a hub revisited hundreds of times), or with several regularisers,
        // is served by the assembled block-CSR operator instead
        const int64_t slot_pr0 = G.rel.Epad + G.sw.Epad;
        for (int64_t n = 0; n < N && mf; ++n) {
            int64_t deg = 0, npri = 0;
            for (int64_t k = rowptr[n]; k < rowptr[n + 1]; ++k) { if ((inc[k] >> 1) >= slot_pr0) ++npri; else ++deg; }
            if (deg > MF_BLOCK || npri > 1) mf = false;
        }
    }
    p->built_mf = mf;
    p->F = MfDev{};
    if (mf) {
        // per keyframe: its relative-pose sides and its switchable sides (both in incident-list order), regulariser index
        std::vector<int32_t> node_prior(N, -1), deg_rel(N, 0), deg_sw(N, 0);
        const int64_t slot_pr = G.rel.Epad + G.sw.Epad;
        for (int64_t n = 0; n < N; ++n) {
            for (int64_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
                const int64_t slot = inc[k] >> 1;
                if (slot >= slot_pr) {
                    if (node_prior[n] >= 0) { p->err = "matrix-free operator: more than one regulariser on a keyframe"; return PGO_ERR_INVALID_ARG; }
                    node_prior[n] = (int32_t)(slot - slot_pr);
                } else if (slot >= G.rel.Epad) ++deg_sw[n]; else ++deg_rel[n];
            }
            if (deg_rel[n] + deg_sw[n] > MF_BLOCK) { p->err = "matrix-free operator: a keyframe with more incident edges than a matrix-free tile holds (use PGO_LINEAR_PCG_BLOCK_JACOBI)"; return PGO_ERR_INVALID_ARG; }
        }
        if ((int64_t)std::max(G.rel.E, G.sw.E) >= (1ll << 30)) { p->err = "matrix-free operator: more than 2^30 edges in one class"; return PGO_ERR_INVALID_ARG; }
        // pack whole keyframes into workgroup tiles: <= MF_SLOTS edge sides, <= MF_BLOCK lanes (a relative-pose edge with both keyframes in
        // the tile takes ONE lane for its two sides), <= MF_MAX_NODES keyframes
        auto rel_other_of = [&](int64_t k) -> int32_t {      // incident entry k of a relative-pose side: the other keyframe, or -1
            const int64_t slot = inc[k] >> 1; const int side = (int)(inc[k] & 1);
            if (slot >= G.rel.Epad) return -1;
            const int32_t a = L(p->rel.c1[slot]), b = L(p->rel.c2[slot]);
            return a == b ? -1 : (side == 0 ? b : a);
        };
        std::vector<int32_t> tile_node0; tile_node0.push_back(0);
        { int64_t sides = 0, pairs = 0; int cur_nodes = 0; int32_t start = 0;
          for (int64_t n = 0; n < N; ++n) {
              const int64_t d = deg_rel[n] + deg_sw[n];
              auto pairs_with = [&](int32_t lo) { int64_t c = 0; for (int64_t k = rowptr[n]; k < rowptr[n + 1]; ++k) { const int32_t o = rel_other_of(k); if (o >= lo && o < (int32_t)n) ++c; } return c; };
              int64_t np = pairs_with(start);
              if (sides + d > MF_SLOTS || sides + d - (pairs + np) > MF_BLOCK || cur_nodes >= MF_MAX_NODES) {
                  tile_node0.push_back((int32_t)n); sides = 0; pairs = 0; cur_nodes = 0; start = (int32_t)n; np = 0;
              }
              sides += d; pairs += np; ++cur_nodes;
          }
          tile_node0.push_back((int32_t)N); }
        const int tiles = (int)tile_node0.size() - 1;
        std::vector<int64_t> tile_inc0(tiles + 1, 0);
        std::vector<int32_t> tile_sw0(std::max(tiles, 1), 0);
        std::vector<uint32_t> einc, eslot; std::vector<int32_t> eoth; std::vector<ushort4> node_rng(std::max<int64_t>(N, 1));
        einc.reserve((size_t)(Er + 2 * Es) + 64); eoth.reserve(einc.capacity()); eslot.reserve(einc.capacity());
        std::vector<uint16_t> side_slot((size_t)(rowptr[N]), 0);       // slot of incident entry k inside its tile
        std::vector<uint16_t> rel_slot1((size_t)std::max<int64_t>(Er, 1), 0);   // per relative-pose edge: slot of its side 1 (own = c2)
        for (int t = 0; t < tiles; ++t) {
            const int32_t n0 = tile_node0[t], n1 = tile_node0[t + 1];
            tile_inc0[t] = (int64_t)einc.size();
            // slots: the keyframes' relative-pose sides, then their switchable sides, each in incident-list order
            int slot_n = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int32_t n = n0; n < n1; ++n) {
                    const unsigned short begin = (unsigned short)slot_n;
                    for (int64_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
                        const int64_t slot = inc[k] >> 1;
                        if (slot >= slot_pr || (int)(slot >= G.rel.Epad) != pass) continue;
                        side_slot[k] = (uint16_t)slot_n;
                        if (pass == 0 && (inc[k] & 1)) rel_slot1[slot] = (uint16_t)slot_n;
                        ++slot_n;
                    }
                    if (pass == 0) { node_rng[n].x = begin; node_rng[n].y = (unsigned short)slot_n; } else { node_rng[n].z = begin; node_rng[n].w = (unsigned short)slot_n; }
                }
            // lanes: pairs, then the other relative-pose sides, then the switchable sides
            int n_pairs = 0;
            for (int group = 0; group < 3; ++group) {
                if (group == 2) tile_sw0[t] = (int32_t)(((int64_t)einc.size() - tile_inc0[t]) | ((int64_t)n_pairs << 16));
                for (int32_t n = n0; n < n1; ++n)
                    for (int64_t k = rowptr[n]; k < rowptr[n + 1]; ++k) {
                        const int64_t slot = inc[k] >> 1; const int side = (int)(inc[k] & 1);
                        if (slot >= slot_pr) continue;
                        const bool is_sw = slot >= G.rel.Epad;
                        if (is_sw != (group == 2)) continue;
                        const int64_t e = is_sw ? slot - G.rel.Epad : slot;
                        const int32_t a = L(is_sw ? p->swe.c1[e] : p->rel.c1[e]), b = L(is_sw ? p->swe.c2[e] : p->rel.c2[e]);
                        const int32_t other = side == 0 ? b : a;
                        const bool paired = !is_sw && a != b && other >= n0 && other < n1;
                        if (group == 0) {
                            if (!paired || side != 0) continue;          // the pair's lane stands at side 0 (own = c1)
                            einc.push_back((uint32_t)(e << 1));
                            eoth.push_back(b);
                            eslot.push_back((uint32_t)side_slot[k] | ((uint32_t)rel_slot1[e] << 9) | ((uint32_t)(n - n0) << 18));
                            ++n_pairs;
                        } else {
                            if (group == 1 && paired) continue;
                            einc.push_back((is_sw ? 0x80000000u : 0u) | (uint32_t)(e << 1) | (uint32_t)side);
                            eoth.push_back(other);
                            eslot.push_back((uint32_t)side_slot[k] | (511u << 9) | ((uint32_t)(n - n0) << 18));
                        }
                    }
            }
        }
        tile_inc0[tiles] = (int64_t)einc.size();
        p->mf_pair_lanes = 0; p->mf_sw_lanes = 0;
        for (int t = 0; t < tiles; ++t) { p->mf_pair_lanes += (uint32_t)tile_sw0[t] >> 16; p->mf_sw_lanes += (tile_inc0[t + 1] - tile_inc0[t]) - (tile_sw0[t] & 0xffff); }
        p->mf_rel_side_lanes = (int64_t)einc.size() - p->mf_pair_lanes - p->mf_sw_lanes;
        const int64_t ninc_e = (int64_t)einc.size();
        const int64_t ninc_pad = (ninc_e + 63) / 64 * 64 + 64;
        HIPCHK(p, p->d_einc.ensure(std::max<int64_t>(ninc_e, 1))); HIPCHK(p, p->d_einc_slot.ensure(std::max<int64_t>(ninc_e, 1))); HIPCHK(p, p->d_einc_other.ensure(std::max<int64_t>(ninc_e, 1)));
        HIPCHK(p, p->d_node_rng.ensure(std::max<int64_t>(N, 1))); HIPCHK(p, p->d_tile_inc0.ensure(tiles + 1)); HIPCHK(p, p->d_tile_node0.ensure(tiles + 1));
        HIPCHK(p, p->d_tile_sw0.ensure(std::max(tiles, 1))); HIPCHK(p, p->d_node_prior.ensure(std::max<int64_t>(N, 1)));
        HIPCHK(p, p->d_rec.ensure((size_t)MF_PLANES * ninc_pad)); HIPCHK(p, p->d_lam.ensure(std::max<int64_t>(N * 6, 1)));
        if (ninc_e) {
            HIPCHK(p, hipMemcpyAsync(p->d_einc.p, einc.data(), ninc_e * sizeof(uint32_t), hipMemcpyHostToDevice, p->st));
            HIPCHK(p, hipMemcpyAsync(p->d_einc_slot.p, eslot.data(), ninc_e * sizeof(uint32_t), hipMemcpyHostToDevice, p->st));
            HIPCHK(p, hipMemcpyAsync(p->d_einc_other.p, eoth.data(), ninc_e * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        }
        HIPCHK(p, hipMemcpyAsync(p->d_node_rng.p, node_rng.data(), N * sizeof(ushort4), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_tile_inc0.p, tile_inc0.data(), (tiles + 1) * sizeof(int64_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_tile_node0.p, tile_node0.data(), (tiles + 1) * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        if (tiles) HIPCHK(p, hipMemcpyAsync(p->d_tile_sw0.p, tile_sw0.data(), tiles * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_node_prior.p, node_prior.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        p->F = MfDev{p->d_einc.p, p->d_einc_other.p, p->d_einc_slot.p, p->d_tile_inc0.p, p->d_tile_sw0.p, p->d_tile_node0.p, p->d_node_rng.p, p->d_node_prior.p,
                     p->d_rec.p, p->d_lam.p, ninc_e, ninc_pad, tiles};
    }
    phase("matrix-free tiles");
    // ---- work buffers
    const int64_t slots = G.rel.Epad + G.sw.Epad;
    HIPCHK(p, p->d_Jr.ensure(std::max<int64_t>((int64_t)G.rel.tiles * REL_DOUBLES * TILE, 1)));
    HIPCHK(p, p->d_Js.ensure(std::max<int64_t>((int64_t)G.sw.tiles * SW_DOUBLES * TILE, 1)));
    HIPCHK(p, p->d_Jp.ensure(std::max<int64_t>(Eg * PRIOR_DOUBLES, 1)));
    HIPCHK(p, p->d_Hd_g.ensure(std::max<int64_t>(N * 42, 1)));
    HIPCHK(p, p->d_Hoff.ensure(mf ? 1 : std::max<int64_t>(slots * 36, 1)));
    HIPCHK(p, p->d_c.ensure(std::max<int64_t>(Es * 12, 1))); HIPCHK(p, p->d_hss.ensure(std::max<int64_t>(Es, 1))); HIPCHK(p, p->d_gs.ensure(std::max<int64_t>(Es, 1)));
    HIPCHK(p, p->d_scale_p.ensure(std::max<int64_t>(N * 6, 1))); HIPCHK(p, p->d_diag_p.ensure(std::max<int64_t>(N * 6, 1)));
    HIPCHK(p, p->d_scale_s.ensure(std::max<int64_t>(Es, 1))); HIPCHK(p, p->d_diag_s.ensure(std::max<int64_t>(Es, 1))); HIPCHK(p, p->d_a_inv.ensure(std::max<int64_t>(Es, 1)));
    HIPCHK(p, p->d_val.ensure(mf ? 1 : std::max<int64_t>(p->nnzb * 36, 1))); HIPCHK(p, p->d_Lf.ensure(std::max<int64_t>(N * 24 + 64 * 24, 1))); HIPCHK(p, p->d_Dtot_b.ensure(std::max<int64_t>(N * 42, 1)));
    HIPCHK(p, p->d_cgvec.ensure(std::max<int64_t>(N * 42, 1)));
    p->n_part = std::max<int64_t>(MAX_PARTIALS, (G.rel.tiles + G.sw.tiles + 3) / 4 + 1);
    HIPCHK(p, p->d_part.ensure(p->n_part * 6));
    HIPCHK(p, p->d_cgpart.ensure(PQ_SLOTS + 2 * RZ_STRIDE + 16));   // partial sums + 16 PCG scalars (C.scal)
    HIPCHK(p, p->d_flags.ensure(8)); HIPCHK(p, p->d_scal.ensure(S_N));
    for (int k = 0; k < 2; ++k) { HIPCHK(p, p->d_pose[k].ensure(std::max<int64_t>(N * 8, 1))); HIPCHK(p, p->d_swv[k].ensure(std::max<int64_t>(S, 1))); }
    HIPCHK(p, p->d_delta_s.ensure(std::max<int64_t>(Es, 1))); HIPCHK(p, p->d_io.ensure(std::max<int64_t>(N * 7, 1)));

    G.rel.J = p->d_Jr.p; G.sw.J = p->d_Js.p;
    G.prior = p->d_prior.p; G.n_prior = (int32_t)Eg; G.Jp = p->d_Jp.p;
    G.inc_rowptr = p->d_inc_rowptr.p; G.inc = p->d_inc.p; G.node_free = p->d_node_free.p;
    G.bsr_rowptr = p->d_bsr_rowptr.p; G.bsr_col = p->d_bsr_col.p; G.nnzb = p->nnzb;
    p->L = LinDev{p->d_Hd_g.p, p->d_Hd_g.p + (size_t)N * 36, p->d_Hoff.p, p->d_c.p, p->d_hss.p, p->d_gs.p};
    p->Sc = ScaleDev{p->d_scale_p.p, p->d_scale_s.p, p->d_diag_p.p, p->d_diag_s.p, p->d_a_inv.p};
    CgDev& C = p->C;
    C.val = p->d_val.p; C.Lf = p->d_Lf.p; C.Dtot = p->d_Dtot_b.p; C.b = p->d_Dtot_b.p + (size_t)N * 36;
    double* v = p->d_cgvec.p; const size_t n6 = (size_t)N * 6;
    C.x = v; C.r = v + n6; C.r2 = v + 2 * n6; C.z = v + 3 * n6; C.p = v + 4 * n6; C.p2 = v + 5 * n6; C.q = v + 6 * n6;
    C.part_pq = p->d_cgpart.p; C.part_rz = p->d_cgpart.p + PQ_SLOTS; C.scal = p->d_cgpart.p + PQ_SLOTS + 2 * RZ_STRIDE; C.extra_rz = 0;
    C.flags = p->d_flags.p;
    // ---- aggregation multigrid for large graphs: hierarchy of graph-following rigid aggregates (pgo_mg_host.hpp), built by build_multigrid() below — which a solve
    // may call again with the current switch values (regroup)
    phase("work buffers");
    if (mg_async) {
        // not waited for here: installed where it is first needed (mg_init_finish).  What build_multigrid would have reset:
        mg_job_cancel(p);
        p->coarse_built = false; p->coarse_active = false; p->K = CoarseDev{};
        p->mg_active = false; p->M = MgDev{}; p->mg_geometry_epoch = 0; p->mg_sw_built.clear();
        p->mg_built = true;      // announced; mg_init_finish corrects it should the graph not coarsen
    } else if ((rc = build_multigrid(p, sw_now, nullptr)) != PGO_OK) return rc;
    phase("multigrid hierarchy");
    if (p->mg_built && p->built_mf) { HIPCHK(p, p->d_Hoff.ensure((size_t)(p->G.rel.Epad + p->G.sw.Epad) * 36)); p->L.Hoff = p->d_Hoff.p; }      // the multigrid's level-1 product reads J1^T J2 per edge
    p->hoff_epoch = 0;
    if (!p->mg_built && (rc = build_two_level_aggregates(p)) != PGO_OK) return rc;      // (a graph that got the multigrid never uses the two-level method: its dense operator would be built and uploaded for nothing)
    phase("two-level aggregates");
    mg_guard.committed = true;
    p->graph_dirty = false; p->priors_dirty = false;
    ++p->build_epoch;   // invalidates the captured PCG graph (kernel arguments hold device pointers / sizes)
    return PGO_OK;
}

// ---- collectives (no-ops without a communicator; a 1-rank communicator still issues every call) ----
// in-process communicator (pgo_comm_local.hpp): step 1 of its protocol — the parity buffers of this collective may be overwritten once the peers' reads of two collectives ago are done
int local_pre(pgo_problem* p) {
    pgo_local::Group* G = p->local_group;
    const int par = (int)(p->lc_count & 1);
    for (int q = 0; q < G->world; ++q) if (q != p->rank && G->slot[q].done[par]) HIPCHK(p, hipStreamWaitEvent(p->st, G->slot[q].done[par], 0));
    return PGO_OK;
}
int local_allreduce(pgo_problem* p, double* buf, size_t n, int op) {
    pgo_local::Group* G = p->local_group;
    pgo_local::Group::Slot& me = G->slot[p->rank];
    const int par = (int)(p->lc_count & 1);
    int rc;
    if ((rc = local_pre(p)) != PGO_OK) return rc;
    if (me.stage_cap[par] < n) {
        HIPCHK(p, hipStreamSynchronize(p->st));
        if (me.stage[par]) (void)hipFree(me.stage[par]);
        me.stage[par] = nullptr; me.stage_cap[par] = 0;
        const size_t want = n + n / 4 + 64;
        HIPCHK(p, hipMalloc((void**)&me.stage[par], want * sizeof(double)));
        me.stage_cap[par] = want;
    }
    HIPCHK(p, hipMemcpyAsync(me.stage[par], buf, n * sizeof(double), hipMemcpyDeviceToDevice, p->st));
    HIPCHK(p, hipEventRecord(me.ready[par], p->st));
    me.ptr[par] = me.stage[par];
    if (!G->barrier()) { p->err = "in-process communicator: a rank did not reach the collective (failed, left, or out of step)"; return PGO_ERR_COMM; }
    LocalPeers P{};
    P.n = G->world;
    for (int q = 0; q < G->world; ++q) {
        P.src[q] = G->slot[q].ptr[par]; P.off[q] = 0; P.cnt[q] = (int64_t)n;
        if (q != p->rank) HIPCHK(p, hipStreamWaitEvent(p->st, G->slot[q].ready[par], 0));
    }
    launch_local_reduce(buf, P, (int64_t)n, op, p->st);
    HIPCHK(p, hipEventRecord(me.done[par], p->st));
    ++p->lc_count;
    return PGO_OK;
}
int allreduce(pgo_problem* p, double* buf, size_t n, int op /*0 sum, 2 max*/) {
    if (p->local_ids || p->comm || p->custom_allreduce || p->local_group) { ++p->st_allreduces; p->st_bytes_allreduce += (double)n * sizeof(double); }
    if (p->local_group) return local_allreduce(p, buf, n, op);
    if (p->custom_allreduce) {
        const int rc = p->custom_allreduce(p->custom_ctx, buf, (int64_t)n, op, (void*)p->st);
        if (rc != 0) { p->err = "custom all-reduce callback failed"; return PGO_ERR_COMM; }
        return PGO_OK;
    }
    if (!p->comm) return PGO_OK;
    const int rc = p->nccl.AllReduce(buf, buf, n, /*ncclDouble*/ 8, op, p->comm, p->st);
    if (rc != 0) { p->err = std::string("ncclAllReduce: ") + (p->nccl.GetErrorString ? p->nccl.GetErrorString(rc) : "error"); return PGO_ERR_COMM; }
    return PGO_OK;
}

// The send buffer of the exchange about to be packed (the in-process communicator double-buffers by collective parity and first waits for the peers' reads of that buffer)
int exchange_send_buffer(pgo_problem* p, double** out) {
    int rc;
    if (p->local_group && (rc = local_pre(p)) != PGO_OK) return rc;
    *out = p->d_xsend[p->local_group ? (p->lc_count & 1) : 0].p;
    return PGO_OK;
}
// Neighbour exchange of `K` doubles per row: rows [plan.send_off[q], plan.send_off[q+1]) of `sendbuf` go to rank q, rows [recv_off[q], recv_off[q+1]) of `recvbuf` come from it.
// RCCL: one group of ncclSend / ncclRecv pairs (point-to-point over the xGMI link of each pair).  In-process communicator: one kernel reading the peers' send buffers.
// Caller-supplied collective: its exchange callback, or — without one — an all-reduce of a zero-padded buffer that holds every pair's segment (correct, world x the bytes).
int neighbor_exchange(pgo_problem* p, const pgo_mg::ExchangePlan& X, int K, const double* sendbuf, double* recvbuf) {
    const int W = p->world, r = p->rank;
    ++p->st_exchanges; p->st_bytes_neighbour += (double)X.n_send() * K * sizeof(double);
    if (p->local_group) {
        pgo_local::Group* G = p->local_group;
        pgo_local::Group::Slot& me = G->slot[r];
        const int par = (int)(p->lc_count & 1);
        HIPCHK(p, hipEventRecord(me.ready[par], p->st));
        me.ptr[par] = sendbuf; me.send_off[par] = X.send_off.data();
        if (!G->barrier()) { p->err = "in-process communicator: a rank did not reach the exchange (failed, left, or out of step)"; return PGO_ERR_COMM; }
        LocalPeers P{};
        int np = 0;
        for (int q = 0; q < W; ++q) {
            const int64_t cnt = (X.recv_off[(size_t)q + 1] - X.recv_off[(size_t)q]) * K;
            if (q == r || cnt == 0) continue;
            const int64_t* so = G->slot[q].send_off[par];
            if ((so[r + 1] - so[r]) * K != cnt) { G->abort(); p->err = "in-process communicator: the ranks' exchange plans disagree"; return PGO_ERR_COMM; }
            HIPCHK(p, hipStreamWaitEvent(p->st, G->slot[q].ready[par], 0));
            P.src[np] = G->slot[q].ptr[par] + so[r] * K; P.off[np] = X.recv_off[(size_t)q] * K; P.cnt[np] = cnt; ++np;
        }
        P.n = np;
        if (np > 0) launch_local_copy(recvbuf, P, p->st);
        HIPCHK(p, hipEventRecord(me.done[par], p->st));
        ++p->lc_count;
        return PGO_OK;
    }
    if (p->custom_allreduce && p->custom_exchange) {
        p->x_off_send.resize((size_t)W + 1); p->x_off_recv.resize((size_t)W + 1);
        for (int q = 0; q <= W; ++q) { p->x_off_send[(size_t)q] = X.send_off[(size_t)q] * K; p->x_off_recv[(size_t)q] = X.recv_off[(size_t)q] * K; }
        const int rc = p->custom_exchange(p->custom_ctx, sendbuf, p->x_off_send.data(), recvbuf, p->x_off_recv.data(), (void*)p->st);
        if (rc != 0) { p->err = "custom exchange callback failed"; return PGO_ERR_COMM; }
        return PGO_OK;
    }
    // PGO_EXCHANGE_VIA_ALLREDUCE=1 (read once; a production switch, not a debug hook): RCCL's point-to-point path is bypassed as well — the safety net for a node where
    // ncclSend / ncclRecv misbehave (this repo's send / receive path has never run between two physical GPUs)
    static const bool via_allreduce = []() { const char* e = std::getenv("PGO_EXCHANGE_VIA_ALLREDUCE"); return e && e[0] == '1' && e[1] == 0; }();
    if (p->custom_allreduce || (p->comm && via_allreduce)) {
        // emulation: [src][dst] segments in one buffer; this rank fills row `r`, the all-reduce fills the rest, column `r` is what it receives
        std::vector<int64_t> off((size_t)W * W + 1, 0);
        for (int i = 0; i < W * W; ++i) off[(size_t)i + 1] = off[(size_t)i] + X.pair_cnt[(size_t)i] * K;
        const size_t total = (size_t)off[(size_t)W * W];
        if (total == 0) return PGO_OK;
        HIPCHK(p, p->d_tmp.ensure(total));
        HIPCHK(p, hipMemsetAsync(p->d_tmp.p, 0, total * sizeof(double), p->st));
        for (int q = 0; q < W; ++q) { const int64_t cnt = (X.send_off[(size_t)q + 1] - X.send_off[(size_t)q]) * K; if (cnt > 0) HIPCHK(p, hipMemcpyAsync(p->d_tmp.p + off[(size_t)r * W + q], sendbuf + X.send_off[(size_t)q] * K, (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, p->st)); }
        int rca;
        if ((rca = allreduce(p, p->d_tmp.p, total, 0)) != PGO_OK) return rca;
        for (int q = 0; q < W; ++q) { const int64_t cnt = (X.recv_off[(size_t)q + 1] - X.recv_off[(size_t)q]) * K; if (cnt > 0) HIPCHK(p, hipMemcpyAsync(recvbuf + X.recv_off[(size_t)q] * K, p->d_tmp.p + off[(size_t)q * W + r], (size_t)cnt * sizeof(double), hipMemcpyDeviceToDevice, p->st)); }
        return PGO_OK;
    }
    if (!p->comm) return PGO_OK;
    if (!p->nccl.Send || !p->nccl.Recv || !p->nccl.GroupStart || !p->nccl.GroupEnd) { p->err = "librccl lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd"; return PGO_ERR_COMM; }
    int rc = p->nccl.GroupStart();
    for (int q = 0; q < W && rc == 0; ++q) {
        if (q == r) continue;
        const int64_t ns = (X.send_off[(size_t)q + 1] - X.send_off[(size_t)q]) * K, nr = (X.recv_off[(size_t)q + 1] - X.recv_off[(size_t)q]) * K;
        if (ns > 0) rc = p->nccl.Send(sendbuf + X.send_off[(size_t)q] * K, (size_t)ns, /*ncclDouble*/ 8, q, p->comm, p->st);
        if (rc == 0 && nr > 0) rc = p->nccl.Recv(recvbuf + X.recv_off[(size_t)q] * K, (size_t)nr, 8, q, p->comm, p->st);
    }
    const int rc_end = p->nccl.GroupEnd();
    if (rc == 0) rc = rc_end;
    if (rc != 0) { p->err = std::string("ncclSend / ncclRecv: ") + (p->nccl.GetErrorString ? p->nccl.GetErrorString(rc) : "error"); return PGO_ERR_COMM; }
    return PGO_OK;
}

// Multi-GPU exchange of the keyframes' rows: sums, over the ranks sharing them, the rows of one or two keyframe-indexed device arrays (k1 + k2 doubles per keyframe).  Every rank
// sends its partial rows of the keyframes it shares with a peer to that peer and adds what it receives in ascending rank order (pgo_mg_host.hpp: build_fine_plan): all ranks
// end up with the same bits.  Keyframes touched by a single rank never travel.  `stop` (device flag): a stopped PCG sends zeros and keeps its rows.
int exchange_rows(pgo_problem* p, double* a1, int k1, double* a2, int k2, const int32_t* stop = nullptr) {
    if (!p->local_ids) return PGO_OK;
    const pgo_mg::FinePlan& F = p->fine_plan;
    const int K = k1 + k2;
    int rc;
    double* sb = nullptr;
    if ((rc = exchange_send_buffer(p, &sb)) != PGO_OK) return rc;
    launch_gather_rows(sb, a1, k1, a2, k2, F.x.n_send(), p->d_fp_send.p, stop, p->st);
    if ((rc = neighbor_exchange(p, F.x, K, sb, p->d_xrecv.p)) != PGO_OK) return rc;
    launch_sum_rows(p->d_xrecv.p, a1, k1, a2, k2, (int64_t)F.sh_loc.size(), p->d_fp_shloc.p, p->d_fp_sumptr.p, p->d_fp_sumsrc.p, stop, p->st);
    return PGO_OK;
}
// ... and of the multigrid's level vectors: the rows of one or two vectors of level `l + 1` this rank owns and a peer reads go to that peer, the rows it reads come in
int exchange_level(pgo_problem* p, int l, double* v1, double* v2, const int32_t* stop, const double* dinv) {
    if (!p->local_ids || (size_t)l >= p->lvl_plan.size() || !p->lvl_plan[(size_t)l].plan) return PGO_OK;
    const pgo_problem::LevelPlanDev& L = p->lvl_plan[(size_t)l];
    int rc;
    double* sb = nullptr;
    if ((rc = exchange_send_buffer(p, &sb)) != PGO_OK) return rc;
    if (dinv) {      // x = v1, r = v2: only r travels, x = Dinv r is formed on receipt (pointwise; every rank holds the level's Dinv)
        launch_gather_rows(sb, v2, 6, nullptr, 0, L.plan->n_send(), L.send_idx, stop, p->st);
        if ((rc = neighbor_exchange(p, *L.plan, 6, sb, p->d_xrecv.p)) != PGO_OK) return rc;
        launch_scatter_rows_dinv(p->d_xrecv.p, v2, v1, dinv, L.plan->n_recv(), L.recv_idx, stop, p->st);
        return PGO_OK;
    }
    const int K = v2 ? 12 : 6;
    launch_gather_rows(sb, v1, 6, v2, v2 ? 6 : 0, L.plan->n_send(), L.send_idx, stop, p->st);
    if ((rc = neighbor_exchange(p, *L.plan, K, sb, p->d_xrecv.p)) != PGO_OK) return rc;
    launch_scatter_rows(p->d_xrecv.p, v1, 6, v2, v2 ? 6 : 0, L.plan->n_recv(), L.recv_idx, stop, p->st);
    return PGO_OK;
}
// ... and of the multigrid's SET-UP (distributed set-up, round 6): 6x6 blocks listed by slot (K doubles each: 36, or 18 for an fp32 block).  Copy: every block has one producer.
// Sum: the parts of a block formed on several ranks are added, in ascending rank order, on every rank that needs it (pgo_mg_host.hpp: BlockPlan).  A plan with nothing to send
// anywhere (pair_cnt, the same on all ranks) is skipped by all of them.
static bool plan_is_empty(const pgo_mg::ExchangePlan& X) { for (int64_t c : X.pair_cnt) if (c) return false; return true; }
int exchange_blocks_copy(pgo_problem* p, const pgo_mg::ExchangePlan& X, const int32_t* send_idx, const int32_t* recv_idx, double* arr, int K) {
    if (plan_is_empty(X)) return PGO_OK;
    int rc;
    double* sb = nullptr;
    if ((rc = exchange_send_buffer(p, &sb)) != PGO_OK) return rc;
    launch_gather_rows(sb, arr, K, nullptr, 0, X.n_send(), send_idx, nullptr, p->st);
    if ((rc = neighbor_exchange(p, X, K, sb, p->d_xrecv.p)) != PGO_OK) return rc;
    launch_scatter_rows(p->d_xrecv.p, arr, K, nullptr, 0, X.n_recv(), recv_idx, nullptr, p->st);
    return PGO_OK;
}
int exchange_blocks_sum(pgo_problem* p, const pgo_mg::BlockPlan& B, const pgo_problem::SetupPlanDev& D, double* arr) {
    if (plan_is_empty(B.x)) return PGO_OK;
    int rc;
    double* sb = nullptr;
    if ((rc = exchange_send_buffer(p, &sb)) != PGO_OK) return rc;
    launch_gather_rows(sb, arr, 36, nullptr, 0, B.x.n_send(), D.val_send, nullptr, p->st);
    if ((rc = neighbor_exchange(p, B.x, 36, sb, p->d_xrecv.p)) != PGO_OK) return rc;
    launch_sum_rows(p->d_xrecv.p, arr, 36, nullptr, 0, (int64_t)B.dst.size(), D.val_dst, D.val_sum_ptr, D.val_sum_src, nullptr, p->st);
    return PGO_OK;
}
// all-reduce of a host vector (graph build: rare, sizes up to a few tens of MB)
int host_allreduce(pgo_problem* p, std::vector<double>& v, int op) {
    if (v.empty()) return PGO_OK;
    HIPCHK(p, p->d_tmp.ensure(v.size()));
    HIPCHK(p, hipMemcpyAsync(p->d_tmp.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
    int rc;
    if ((rc = allreduce(p, p->d_tmp.p, v.size(), op)) != PGO_OK) return rc;
    HIPCHK(p, hipMemcpyAsync(v.data(), p->d_tmp.p, v.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

// keyframe-indexed device array of this handle (k doubles per keyframe) -> the caller's array over ALL keyframes, complete on every rank
// (multi-GPU: each keyframe is contributed by its owner; keyframes no rank touches come back as zeros)
int nodes_to_global(pgo_problem* p, const double* dev, int k, double* host_global) {
    if (!p->local_ids) {
        HIPCHK(p, hipMemcpyAsync(host_global, dev, (size_t)p->N * k * sizeof(double), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        return PGO_OK;
    }
    std::vector<double> loc((size_t)p->N * k), glob((size_t)p->N_global * k, 0.0);
    HIPCHK(p, hipMemcpyAsync(loc.data(), dev, loc.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    for (int64_t l = 0; l < p->N; ++l) if (p->h_own[l] != 0.0) std::copy(loc.begin() + l * k, loc.begin() + (l + 1) * k, glob.begin() + (size_t)p->l2g[l] * k);
    HIPCHK(p, p->d_tmp.ensure(glob.size()));
    HIPCHK(p, hipMemcpyAsync(p->d_tmp.p, glob.data(), glob.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
    int rc;
    if ((rc = allreduce(p, p->d_tmp.p, glob.size(), 0)) != PGO_OK) return rc;
    HIPCHK(p, hipMemcpyAsync(host_global, p->d_tmp.p, glob.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}
// the caller's array over all keyframes -> this handle's keyframes on the device
int nodes_from_global(pgo_problem* p, const double* host_global, int k, double* dev) {
    if (!p->local_ids) {
        HIPCHK(p, hipMemcpyAsync(dev, host_global, (size_t)p->N * k * sizeof(double), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        return PGO_OK;
    }
    std::vector<double> loc((size_t)p->N * k);
    for (int64_t l = 0; l < p->N; ++l) std::copy(host_global + (size_t)p->l2g[l] * k, host_global + (size_t)(p->l2g[l] + 1) * k, loc.begin() + l * k);
    HIPCHK(p, hipMemcpyAsync(dev, loc.data(), loc.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

double* part(pgo_problem* p, int k) { return p->d_part.p + (size_t)k * p->n_part; }

// K1 (+ regularisers) at state `which`; cost lands in d_scal[S_COST], d_scal[S_PRIOR_COST]
int run_k1(pgo_problem* p, int which, bool want_j) {
    int np = 0;
    launch_k1(p->G, p->d_pose[which].p, p->d_swv[which].p, want_j, part(p, 0), &np, p->st);
    if (np > 0) launch_reduce(part(p, 0), np, 0, p->d_scal.p + S_COST, p->st);
    else HIPCHK(p, hipMemsetAsync(p->d_scal.p + S_COST, 0, sizeof(double), p->st));
    launch_prior(p->G, p->d_pose[which].p, want_j, p->d_scal.p + S_PRIOR_COST, p->st);
    return PGO_OK;
}

int read_scalars(pgo_problem* p, double* h) {
    // edge-local sums [S_COST..S_SW_XNORM2] are summed over ranks; the projected-gradient norm takes the max
    // (max), the keyframe sums S_STEP2 / S_XNORM2 are owner-weighted partial sums on every rank
    int rc;
    if ((rc = allreduce(p, p->d_scal.p, 5, 0)) != PGO_OK) return rc;
    if ((rc = allreduce(p, p->d_scal.p + S_GMAX, 1, 2)) != PGO_OK) return rc;
    if ((rc = allreduce(p, p->d_scal.p + S_STEP2, 2, 0)) != PGO_OK) return rc;
    HIPCHK(p, hipMemcpyAsync(h, p->d_scal.p, S_N * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

// linearise at the current state: K1 + K2 (+ all-reduce of diagonal blocks and gradient), norms
int linearize(pgo_problem* p, double* cost_out) {
    int rc;
    if ((rc = run_k1(p, p->cur, true)) != PGO_OK) return rc;
    launch_k2(p->G, p->L, !p->built_mf, p->st, p->built_mf ? &p->F : nullptr);
    ++p->lin_epoch;
    if (p->built_mf) launch_mf_compact(p->G, p->F, p->d_pose[p->cur].p, p->d_swv[p->cur].p, p->st);
    if ((rc = exchange_rows(p, p->L.Hd, 36, p->L.g, 6)) != PGO_OK) return rc;   // diagonal blocks + gradient of shared keyframes
    if (!p->scale_ready) { launch_scale_init(p->G, p->L, p->Sc, p->opt.jacobi_scaling, p->st); p->scale_ready = true; }
    int np = 0;
    launch_state_norms(p->G, p->L, p->d_pose[p->cur].p, p->d_swv[p->cur].p, part(p, 1), part(p, 2), part(p, 3), &np, p->st);
    launch_reduce(part(p, 1), np, 0, p->d_scal.p + S_XNORM2, p->st);
    launch_reduce(part(p, 2), np, 0, p->d_scal.p + S_SW_XNORM2, p->st);
    launch_reduce(part(p, 3), np, 1, p->d_scal.p + S_GMAX, p->st);
    HIPCHK(p, hipMemsetAsync(p->d_scal.p + S_MODEL, 0, 2 * sizeof(double), p->st));
    HIPCHK(p, hipMemsetAsync(p->d_scal.p + S_STEP2, 0, sizeof(double), p->st));   // not produced here; keeps the summed slot finite
    double h[S_N];
    if ((rc = read_scalars(p, h)) != PGO_OK) return rc;
    *cost_out = 0.5 * (h[S_COST] + h[S_PRIOR_COST]);
    p->x_norm = std::sqrt(h[S_XNORM2] + h[S_SW_XNORM2]);
    p->gmax = h[S_GMAX];
    return PGO_OK;
}

static int build_mg(pgo_problem* p);
static int regroup_if_moved(pgo_problem* p, const double* sv, bool in_solve);
static int regroup_start(pgo_problem* p);
// c = w_p / w of the smoothed prolongators (Dinv holds w D^-1)
double mg_cs(const pgo_problem* p) {
    const double om = p->opt.mg_omega > 0.0 && p->opt.mg_omega <= 1.0 ? p->opt.mg_omega : 0.9;
    const double wp = p->opt.mg_prolongation_damping > 0.0 && p->opt.mg_prolongation_damping < 0.85 ? p->opt.mg_prolongation_damping : 0.6;
    return wp / om;
}
static const MgLevelDev* mg_fine_view(const pgo_problem* p) { return p->mg_fine ? &p->mg_fineT : nullptr; }      // smoothed keyframe transition: what launch_mg_apply restricts and prolongs with
double mg_scale(const pgo_problem* p) { return p->opt.mg_correction_scale >= 1.0 && p->opt.mg_correction_scale <= 4.0 ? p->opt.mg_correction_scale : 1.0; }

struct CgResult { int iterations; bool breakdown; double rel_residual; bool converged; };

// Several ranks: which exchange the multigrid cycle needs at one of launch_mg_apply's hook points (pgo_internal.hpp: MgExchangeHook) — the plan (index of the level whose vectors
// travel) and the one or two vectors; false: none.  Level `lv` (1-based) is "distributed" when its kernels run on the owner's rows only; otherwise every rank runs all its rows.
//   point 0, down-sweep of lv (lv = n_levels: the dense solve):  a distributed level reads x (with an explicit transfer operator also r) on the halo of its rows; a level every
//            rank runs completely needs r and x complete — a gather — when what produced them ran on owned rows only (the level below is distributed, or lv = 1: the restriction
//            from the keyframes covers the rank's own aggregates)
//   point 1, up-sweep of a distributed lv:  plain transition: xt of lv on the halo, unless the level above wrote all of it (a level every rank runs completely, or the dense
//            solve, prolongs into every child it holds a valid x for: own rows + halo); explicit operator: xf of lv + 1 on the columns of R^T, when that level is distributed
//   point 2, prolongation to the keyframes:  xf of level 1 at the aggregates of every keyframe the rank touches, when level 1 is distributed
bool mg_exchange_at(pgo_problem* p, int point, int lv, int* plan, double** v1, double** v2, const double** dinv /* non-null result: only r (*v2) travels, x (*v1) = Dinv r is formed on receipt */) {
    const int nl = p->M.n_levels;
    auto dist = [&](int level) { return level >= 1 && level < nl && (size_t)(level - 1) < p->mg_dist.size() && p->mg_dist[(size_t)level - 1] != 0; };
    auto expl = [&](int level) { return level >= 1 && level < nl && p->mg_levels[level - 1].smoothed && p->mg_levels[level - 1].rt_valf != nullptr; };
    *v1 = nullptr; *v2 = nullptr; *plan = -1; *dinv = nullptr;
    if (p->world <= 1 || p->lvl_plan.empty()) return false;
    if (point == 0) {
        if (lv == nl) { if (nl == 1 || dist(nl - 1)) { *plan = nl - 1; *v1 = p->K.rc; return true; } return false; }
        MgLevelDev& A = p->mg_levels[lv - 1];
        if (dist(lv)) { *plan = lv - 1; *v1 = A.x; if (expl(lv)) { *v2 = A.r; *dinv = A.Dinv; } return true; }
        if (lv == 1 || dist(lv - 1)) { *plan = lv - 1; *v1 = A.x; *v2 = A.r; *dinv = A.Dinv; return true; }
        return false;
    }
    if (point == 1) {
        if (!dist(lv)) return false;
        if (expl(lv)) { if (dist(lv + 1)) { *plan = lv; *v1 = p->mg_levels[lv].xf; return true; } return false; }
        if (dist(lv + 1)) { *plan = lv - 1; *v1 = p->mg_levels[lv - 1].xt; return true; }
        return false;
    }
    if (point == 2) { if (nl >= 2 && dist(1)) { *plan = nl; *v1 = p->mg_levels[0].xf; return true; } return false; }      // (the prolongation's own plan: a subset of level 1's halo)
    return false;
}
struct MgHookCtx { pgo_problem* p; const int32_t* stop; };
int mg_exchange_hook(void* ctx, int point, int level) {
    MgHookCtx* c = static_cast<MgHookCtx*>(ctx);
    int plan; double* v1; double* v2; const double* dinv;
    if (!mg_exchange_at(c->p, point, level, &plan, &v1, &v2, &dinv)) return PGO_OK;
    return exchange_level(c->p, plan, v1, v2, c->stop, dinv);
}
// z += s P V(P^T r) on several ranks: the cycle's kernels on this rank's share of every level, the exchanges their reads need in between
int mg_apply_ranks(pgo_problem* p, bool inside_iteration) {
    MgHookCtx hc{p, inside_iteration ? p->C.flags : nullptr};
    MgExchangeHook hook{&hc, mg_exchange_hook};
    int hrc = PGO_OK;
    launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz, mg_scale(p), inside_iteration, p->st, false, mg_cs(p), nullptr, &hook, &hrc);
    return hrc;
}


// One GPU, matrix-free matvec, tolerance not below 1e-11: the PCG runs in its single-reduction (Chronopoulos-Gear) form — matvec w = A u with the partials of u.w, then ONE
// vector kernel whose head re-reduces u.w and r.u together (pgo_kernels.hip: sr_head).  Decided by the options alone, so every phase of a paused PCG runs the same form.
// The two-level method: its FUSED three-kernel iteration has a single-reduction form of its own (launch_mf_apply_dot_live_coarse + launch_cg_update_restrict_sr) and runs it under the
// same gates (tolerance >= 1e-11, <= 150 000 keyframes); only its unfused form — aggregates too large for the update kernel's groups — stays classic.
bool single_reduction(const pgo_problem* p) {
    // (the two-level method: only its fused three-kernel iteration has a single-reduction form; its unfused form — aggregates too large for the update kernel's groups — stays classic)
    const bool two_level = p->coarse_active && !p->mg_active;
    // ... and only where the iteration is latency-bound: the form trades one partial-sum head (~4.5 us) for 96 more bytes per keyframe and iteration, which costs more than the
    // head from ~130 000 keyframes on — measured +1.4 % on C3 (100k) and +2...+7 % on 12k-60k-keyframe graphs, but -1.2 % on C4 (200k) and -1.7 % on C5 (1M)
    // (profiles/r05_single_reduction_graph_types.txt, r05_option_ab_c4_c5.txt)
    constexpr int64_t SINGLE_REDUCTION_MAX_KEYFRAMES = 150000;
    return p->opt.cg_single_reduction != 0 && !p->local_ids && p->built_mf && p->opt.cg_rel_tolerance >= 1e-11 && p->N_global <= SINGLE_REDUCTION_MAX_KEYFRAMES &&
           (!two_level || coarse_group_keyframes(p->K) > 0);
}

// rel_tol: relative tolerance of this phase.  resume_from >= 0: continue the stopped PCG at that iteration index with the new tolerance
// (device state x, r, z, p and the partial sums are those of `resume_from` completed iterations).
int run_pcg(pgo_problem* p, CgResult* res, bool warm, double rel_tol, int resume_from, bool switch_now = false) {
    const pgo_options& o = p->opt;
    int rc0;
    const double tol2 = rel_tol * rel_tol;
    if (resume_from >= 0) {
        launch_cg_set_tolerance(p->C, tol2, p->st);
    } else if (warm) {
        // after a rejected step the system keeps H and only the damping grows: start from the previous solution (q = A x first)
        if (p->built_mf) launch_mf_apply(p->G, p->F, p->Sc, p->C, p->C.x, p->C.q, p->st);
        else launch_apply_operator(p->G, p->C, p->C.x, p->C.q, p->st);
        if ((rc0 = exchange_rows(p, p->C.q, 6, nullptr, 0)) != PGO_OK) return rc0;
    }
    // Multi-GPU: the PCG runs in Chronopoulos-Gear form (pgo_kernels.hip): both dot products of an iteration — gamma = r.u (owner-weighted partials of the previous update) and
    // delta = u.A u (rank-local partials) — are known right after the matvec: ONE 2-double all-reduce per iteration, beside the neighbour exchange of the shared rows of w = A u.
    const bool multi = p->local_ids;
    // two-level preconditioner in three kernels per iteration (prolongation inside the matvec, restriction inside the update, r.(P y) from the dense solve):
    // the update kernel's r.z partials take `fused_parts` slots, the solve's C.extra_rz slots behind them
    const bool fused_coarse = !multi && p->coarse_active && !p->mg_active && p->built_mf && coarse_group_keyframes(p->K) > 0;
    const int fused_parts = fused_coarse ? coarse_update_grid(p->G, p->K) : 0;
    if (fused_coarse) p->C.extra_rz = coarse_solve_grid(p->K);
    else if (!p->mg_active) p->C.extra_rz = 0;
    // several ranks, PCG start: r = b (- A x), u = M^-1 r, p = s = 0; part_rz <- owner-weighted partials of gamma_0 (summed over ranks with the first iteration's scalars),
    // part_pq <- partials of b.D^-1 b, summed over ranks here once: the reference norm of the stopping test.  With the multigrid: the distributed cycle (mg_apply_ranks).
    auto start_multi = [&](int warm_i) -> int {
        int rcs;
        const int g = launch_cg_init_vectors(p->G, p->C, warm_i, p->st);
        if (p->mg_active && (rcs = mg_apply_ranks(p, false)) != PGO_OK) return rcs;      // z += P0 V(P0^T r): the restriction covers the rank's own aggregates (all their keyframes are local)
        double* bb = p->C.scal + 12;
        launch_reduce(p->C.part_pq, g, 0, bb, p->st);
        if ((rcs = allreduce(p, bb, 1, 0)) != PGO_OK) return rcs;
        launch_cgcg_scalars_init(p->C, bb, tol2, p->st);
        return PGO_OK;
    };
    if (resume_from < 0) {
        if (!multi && (p->coarse_active || p->mg_active)) {
            // z = D^-1 r + P Ac^-1 P^T r (or the multigrid cycle): the coarse term is added to z and to the r.z partials before the scalars are formed
            int g = launch_cg_init_vectors(p->G, p->C, warm ? 1 : 0, p->st);
            const int g_bb = g;      // the slots of part_pq that hold the partials of b.D^-1 b
            if (p->mg_active) launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz, mg_scale(p), false, p->st, false, mg_cs(p), mg_fine_view(p));
            else launch_coarse_apply(p->G, p->C, p->K, p->C.r, p->C.z, p->C.part_rz, false, p->st);
            if (fused_coarse) {    // z is complete here: the slots the fused kernels will use beyond the start-up kernels' stay zero for this parity
                HIPCHK(p, hipMemsetAsync(p->C.part_rz + g, 0, (size_t)(fused_parts + p->C.extra_rz - g) * sizeof(double), p->st));
                g = fused_parts;
            }
            launch_cg_init_scalars(p->C, g, g_bb, tol2, p->st);
        } else if (!multi) launch_cg_init(p->G, p->C, warm ? 1 : 0, tol2, p->st);
        else if ((rc0 = start_multi(warm ? 1 : 0)) != PGO_OK) return rc0;
    }
    int k = resume_from >= 0 ? resume_from : 0;
    int32_t hflags[3] = {0, 0, 0};
    double hscal[3] = {0, 0, 0};
    int every = 2;
    auto chunk_length = [&]() {
        int e = std::max(2, o.cg_check_every) & ~1;   // even: the r/p ping-pong parity repeats from chunk to chunk
        // captured chunks stay at <= 72 kernel nodes (rocprofv3 7.2 crashes while a graph of 120 nodes is captured under --kernel-trace; 80 are fine): five kernels
        // per iteration with the coarse space in its unfused form -> 12 iterations, three in the fused form -> 24
        if (p->coarse_active && !multi) e = std::min(e, fused_coarse ? 24 : 12);
        int n_sm = 0;
        for (int l = 0; l < p->M.n_levels; ++l) n_sm += (p->mg_levels[l].smoothed && !p->mg_levels[l].rt_valf) ? 1 : 0;      // two more kernels per cycle for every level whose smoothed prolongator is applied implicitly (none with the explicit transfer operator)
        if (p->mg_active && multi) e = std::min(e, std::max(2, (72 / (6 * p->M.n_levels + 12)) & ~1));      // (every exchange is a pack kernel, the transfer and an unpack kernel)
        if (p->mg_active && !multi) e = std::max(2, (72 / (2 * p->M.n_levels + 3 + 2 * n_sm)) & ~1);   // at most 2 n_levels + 1 cycle kernels + matvec + update per iteration (one less with the restriction inside the update)
        return e;
    };
    every = chunk_length();
    int rc;
    const bool sr = single_reduction(p);
    auto one_iteration = [&](int kk) -> int {
        if (sr && fused_coarse) {      // two-level method: w = A (z_bj + P y), update + restriction with the one reduction point, dense solve (y, coarse part of r.u)
            const int pending = kk > 0 ? 1 : 0;      // (iteration 0: the PCG start has left the complete u in C.z)
            launch_mf_apply_dot_live_coarse(p->G, p->F, p->Sc, p->C, p->K, pending, p->st);
            launch_cg_update_restrict_sr(p->G, p->C, p->K, kk, kk == 0 ? 1 : 0, pending, mf_grid_size(p->F), p->st);
            launch_coarse_solve_dot(p->K, p->C.flags, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE + fused_parts, p->st);
            return PGO_OK;
        }
        if (sr) {      // matvec (no head: it only asks whether the PCG has stopped), update with the iteration's one reduction point, [the multigrid cycle]
            launch_mf_apply_dot_live(p->G, p->F, p->Sc, p->C, p->st);
            const int n_pq = mf_grid_size(p->F), first = kk == 0 ? 1 : 0;      // (first: also when a PCG that stopped before its first update is resumed — p = s = 0 still)
            const bool mg_restrict_fused = p->mg_active && p->M.blk_tab != nullptr;
            if (mg_restrict_fused) launch_cg_update_mg_sr(p->G, p->C, p->M, p->mg_levels, p->K, kk, first, n_pq, p->st);
            else launch_cg_update_sr(p->G, p->C, kk, first, n_pq, p->st);
            if (p->mg_active) launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE, mg_scale(p), true, p->st, mg_restrict_fused, mg_cs(p), mg_fine_view(p));
            return PGO_OK;
        }
        if (multi) {
            const int g = cg_grid_size(p->G);
            int g_pq = g;
            if (p->built_mf) { launch_mf_apply_dot(p->G, p->F, p->Sc, p->C, p->C.z, p->C.q, p->st); g_pq = mf_grid_size(p->F); }   // w = A_r u and the partials of u.w in one kernel
            else { launch_apply_operator(p->G, p->C, p->C.z, p->C.q, p->st); launch_cgcg_dots(p->G, p->C, p->st); }
            // The iteration's exchanges: the partial rows of w of the keyframes this rank shares go to the ranks sharing them (one group of sends / receives), the parts are
            // summed in ascending rank order; [delta, gamma] by ONE all-reduce of two doubles.  Then the update; with the multigrid the distributed cycle.
            int r2;
            launch_cg_reduce2_live(p->C, p->C.part_pq, g_pq, p->C.part_rz, g, p->d_xscal.p, p->st);
            if ((r2 = exchange_rows(p, p->C.q, 6, nullptr, 0, p->C.flags)) != PGO_OK) return r2;
            if ((r2 = allreduce(p, p->d_xscal.p, 2, 0)) != PGO_OK) return r2;
            launch_cgcg_update(p->G, p->C, kk, kk == 0 ? 1 : 0, p->st, nullptr, nullptr, p->d_xscal.p);   // (first: also when a PCG that stopped before its first update is resumed: p = s = 0 still)
            ++p->st_pcg_iterations;
            if (p->mg_active && (r2 = mg_apply_ranks(p, true)) != PGO_OK) return r2;      // u = D^-1 r + P0 V(P0^T r)
            return PGO_OK;
        }
        if (fused_coarse) {
            launch_mf_spmv_coarse(p->G, p->F, p->Sc, p->C, p->K, kk, tol2, fused_parts, kk > 0 ? 1 : 0, p->st);
            launch_cg_update_restrict(p->G, p->C, p->K, kk, mf_grid_size(p->F), p->st);
            launch_coarse_solve_dot(p->K, p->C.flags, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE + fused_parts, p->st);
            return PGO_OK;
        }
        int n_pq = cg_grid_size(p->G);
        if (p->built_mf) { launch_mf_spmv(p->G, p->F, p->Sc, p->C, kk, tol2, p->st); n_pq = mf_grid_size(p->F); }
        else launch_cg_spmv(p->G, p->C, kk, tol2, p->st);
        const bool mg_restrict_fused = p->mg_active && p->M.blk_tab != nullptr;      // the vector update also restricts the new residual to level 1
        if (mg_restrict_fused) launch_cg_update_mg(p->G, p->C, p->M, p->mg_levels, p->K, kk, n_pq, p->st);
        else launch_cg_update(p->G, p->C, kk, n_pq, p->st);
        // the new residual is in the OTHER r buffer, its r.z partials in the other parity's slots
        if (p->mg_active) launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, (kk & 1) ? p->C.r : p->C.r2, p->C.z, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE, mg_scale(p), true, p->st, mg_restrict_fused, mg_cs(p), mg_fine_view(p));
        else if (p->coarse_active)
            launch_coarse_apply(p->G, p->C, p->K, (kk & 1) ? p->C.r : p->C.r2, p->C.z, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE, true, p->st);
        return PGO_OK;
    };
    // hipGraph: capture one chunk (iterations 2 .. 2+every-1: no `first` kernel, even start) once per graph build and preconditioner, and replay it
    // Several ranks: a chunk holding RCCL's all-reduce can be captured as well (RCCL supports stream capture); not with a caller-supplied collective (a host callback).
    // Opt-in (PGO_RCCL_GRAPH=1): it could only be tried with a 1-rank communicator on the 1-GPU development boxes.
    static const bool rccl_graph = []() { const char* e = std::getenv("PGO_RCCL_GRAPH"); return e && e[0] == '1'; }();
    const bool want_graph = o.cg_use_graph && !p->cg_graph_failed && (!p->local_ids || (p->comm != nullptr && p->custom_allreduce == nullptr && rccl_graph));
    // Capture + instantiation cost about a millisecond: a PCG pays it only once it has run `graph_after` iterations eagerly (a graph that is rebuilt for every
    // solve — the reference's sessions: one new loop edge, one solve — and converges in a few hundred iterations never does; eager launches keep up with
    // 5-8 us kernels: measured 18.5 vs 19.4 ms at 300 keyframes, 64.0 vs 64.6 ms at 3000)
    const int graph_after = debug_graph_after();
    auto ensure_graph = [&](bool may_capture) {
        const int mode = p->mg_active ? 2 : p->coarse_active ? 1 : 0;
        pgo_problem::CapturedChunk& cc = p->cg_chunk[mode];
        if (!want_graph || p->cg_graph_failed || (cc.exec != nullptr && cc.epoch == p->build_epoch && cc.len == every && cc.scale == mg_scale(p) && cc.sr == sr)) { p->cg_graph = want_graph && !p->cg_graph_failed ? cc.exec : nullptr; return; }
        if (!may_capture) { p->cg_graph = nullptr; return; }
        if (cc.exec) { (void)hipGraphExecDestroy(cc.exec); cc.exec = nullptr; }
        hipGraph_t gr = nullptr;
        bool ok = hipStreamBeginCapture(p->st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            for (int j = 0; j < every; ++j) (void)one_iteration(2 + j);
            ok = hipStreamEndCapture(p->st, &gr) == hipSuccess && gr != nullptr;
        }
        const double t_inst = now_s();
        if (ok) ok = hipGraphInstantiate(&cc.exec, gr, nullptr, nullptr, 0) == hipSuccess;
        if (o.verbosity > 1) std::fprintf(stderr, "[pgo] PCG chunk of %d iterations (preconditioner %d) captured, instantiated in %.2f ms\n", every, mode, (now_s() - t_inst) * 1e3);
        if (gr) (void)hipGraphDestroy(gr);
        if (!ok) { cc.exec = nullptr; p->cg_graph_failed = true; (void)hipGetLastError(); }
        else { cc.epoch = p->build_epoch; cc.len = every; cc.scale = mg_scale(p); cc.sr = sr; }
        p->cg_graph = cc.exec;
    };
    ensure_graph(k >= graph_after);
    // Chunks of `every` iterations; the convergence flag of chunk j is read (pinned memory + event) only AFTER chunk j+1 has been
    // enqueued, so the GPU never drains while the host polls.  A chunk enqueued after convergence is a string of early-exit kernels.
    int n_chunks = 0, waited = -1;
    int ex_k0 = -1; double ex_rz0 = 0.0;      // first polled (iteration, r.z) of this run: base of the convergence-rate estimate
    bool done = false;
    // END GAME (round 5, one GPU).  A chunk enqueued past convergence is a string of early-exit kernels (~2 us each: 100-150 us per stopped PCG with a chunk in flight, more
    // than a tenth of a session-sized PCG).  The polled r.z values give the convergence rate; once the predicted remaining iterations fall below two chunks the host stops
    // running ahead: it enqueues what the prediction asks for (+15 % + 4 iterations: an early-exit iteration costs a quarter of a host round trip), eagerly, and polls at once.
    // Chunk lengths depend on the device's own r.z values alone — the PCG's iterates do not depend on how its iterations are cut into chunks.
    const bool end_game = o.cg_end_game != 0 && !multi;
    bool eg_tight = false, eg_have = false; int eg_next = every; int eg_k = 0; double eg_rz = 0.0;
    auto eg_snapshot = [&]() { if (end_game) launch_cg_poll(p->C, p->poll[2].flags, p->poll[2].scal, p->st); eg_have = false; };      // (read only after a later poll's event: stream order)
    auto eg_update = [&](int slot) {      // a completed poll: new rate estimate from the last two points, length of the next chunk
        if (!end_game) return;
        const int kk = p->poll[slot].flags[2];
        const double rz = p->poll[slot].scal[1], bb = p->poll[slot].scal[0];
        if (!eg_have) { eg_k = p->poll[2].flags[2]; eg_rz = p->poll[2].scal[1]; eg_have = true; }
        eg_tight = false; eg_next = every;
        if (rz > 0.0 && bb > 0.0 && eg_rz > 0.0 && kk > eg_k && rz < eg_rz) {
            const double lr = std::log(rz / eg_rz) / (double)(kk - eg_k);
            const double need = std::log(tol2 * bb / rz);
            const double left = need < 0.0 ? need / lr - (double)(k - kk) : 0.0;      // iterations still to run beyond what is already enqueued
            if (left < 2.0 * (double)every) {
                eg_tight = true;
                const int want = (int)std::ceil(std::max(left, 0.0) * 1.15 + 4.0);
                eg_next = std::max(2, std::min(every, (want + 1) & ~1));
            }
        }
        if (kk > eg_k) { eg_k = kk; eg_rz = rz; }
    };
    auto enqueue_poll = [&](int slot) -> int {
#ifdef PGO_POLL_BY_COPY
        HIPCHK(p, hipMemcpyAsync(p->poll[slot].flags, p->C.flags, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipMemcpyAsync(p->poll[slot].scal, p->C.scal, 3 * sizeof(double), hipMemcpyDeviceToHost, p->st));
#else
        launch_cg_poll(p->C, p->poll[slot].flags, p->poll[slot].scal, p->st);
#endif
        HIPCHK(p, hipEventRecord(p->poll_ev[slot], p->st));
        return PGO_OK;
    };
    // block-Jacobi -> multigrid inside one system: operators built now, PCG restarted from the current iterate (`so_far` iterations are booked as cg_extra)
    auto switch_to_mg = [&](int so_far) -> int {
        int rcs;
        if ((rcs = build_mg(p)) != PGO_OK) return rcs;
        if (!p->mg_active) { p->mg_failed = true; return PGO_OK; }
        p->cg_extra += so_far;
        if (p->built_mf) launch_mf_apply(p->G, p->F, p->Sc, p->C, p->C.x, p->C.q, p->st);
        else launch_apply_operator(p->G, p->C, p->C.x, p->C.q, p->st);
        if (multi) {
            if ((rcs = exchange_rows(p, p->C.q, 6, nullptr, 0)) != PGO_OK) return rcs;
            if ((rcs = start_multi(1)) != PGO_OK) return rcs;
        } else {
            const int g = launch_cg_init_vectors(p->G, p->C, 1, p->st);
            launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz, mg_scale(p), false, p->st, false, mg_cs(p), mg_fine_view(p));
            launch_cg_init_scalars(p->C, g, g, tol2, p->st);
        }
        k = 0; n_chunks = 0; waited = -1;
        every = chunk_length();
        eg_tight = false; eg_next = every; eg_have = false; eg_snapshot();
        ensure_graph(true);      // a system that needed the switch is a long one
        return PGO_OK;
    };
    // a system predicted hard whose step has survived the first early-rejection pause (lm_step): the multigrid takes over from the iterate the pause left
    if (switch_now && resume_from >= 0 && p->mg_built && !p->mg_active && !p->mg_failed && (rc = switch_to_mg(resume_from)) != PGO_OK) return rc;
    eg_next = every;
    if (end_game && (resume_from >= 0 || k == 0)) eg_snapshot();
    while (k < o.cg_max_iterations && !done) {
        if (eg_tight && n_chunks > 0 && waited < n_chunks - 1) {      // end game: the chunk in flight is waited for before anything else is enqueued
            HIPCHK(p, hipEventSynchronize(p->poll_ev[(n_chunks - 1) & 1]));
            waited = n_chunks - 1;
            if (p->poll[waited & 1].flags[0]) { done = true; break; }
            eg_update(waited & 1);
        }
        // (a phase that only has to reach an early-rejection pause's loose tolerance is a matter of a few iterations: its first chunk is short, the rate estimate takes over from there)
        const int first_short = end_game && n_chunks == 0 && rel_tol >= 5e-3 ? std::min(every, 8) : every;
        const int chunk = std::min(eg_tight ? eg_next : first_short, o.cg_max_iterations - k);
        if (want_graph && !p->cg_graph_failed && !p->cg_graph && k >= graph_after && (k & 1) == 0) ensure_graph(true);
        if (k >= 2 && chunk == every && want_graph && p->cg_graph && (k & 1) == 0) {
            HIPCHK(p, hipGraphLaunch(p->cg_graph, p->st));
            k += every;
        } else {
            // iterations 0,1 run eagerly (iteration 0 has its own kernel arguments); an odd resume index takes one eager iteration to realign
            const int n = k == 0 ? std::min(2, chunk) : ((k & 1) ? 1 : chunk);
            const bool startup = k == 0 || (k & 1);
            for (int j = 0; j < n; ++j, ++k) if ((rc = one_iteration(k)) != PGO_OK) return rc;
            if (startup && k < o.cg_max_iterations) continue;     // no host poll after the start-up iterations
        }
        if ((rc = enqueue_poll(n_chunks & 1)) != PGO_OK) return rc;
        // the first two chunks are polled immediately (short solves finish there); afterwards one chunk stays in flight
        const int check = (n_chunks < 2 || eg_tight) ? n_chunks : n_chunks - 1;
        if (check > waited) {
            HIPCHK(p, hipEventSynchronize(p->poll_ev[check & 1]));
            waited = check;
            if (p->poll[check & 1].flags[0]) done = true;
            else eg_update(check & 1);
            // A system without a prediction (the first of a solve, the first after rejected steps) need not burn mg_switch_iterations block-Jacobi iterations to be
            // recognised as hard: the polled r.z values give its convergence rate, and a system that would need >= the start threshold in total at that rate (and at
            // least twice what it has done) switches now.  Depends on the solve's own data alone; several ranks: r.z and the reference norm are all-reduced values, every
            // rank sees the same bits and takes the same branch.
            if (!done && p->mg_built && !p->mg_active && !p->mg_failed && !p->mg_start_deferred && o.mg_switch_iterations > 0 && k < p->mg_switch_at) {
                const int kk = p->poll[check & 1].flags[2];
                const double rz = p->poll[check & 1].scal[1], bb = p->poll[check & 1].scal[0];
                if (rz > 0.0 && bb > 0.0) {
                    if (ex_k0 < 0) { if (kk >= 24) { ex_k0 = kk; ex_rz0 = rz; } }
                    else if (kk >= 96 && kk > ex_k0) {
                        const double lr = std::log(rz / ex_rz0) / (double)(kk - ex_k0);                        // log reduction per iteration (negative while converging)
                        const double need = std::log(o.cg_rel_tolerance * o.cg_rel_tolerance * bb / rz);      // what is left down to the final tolerance (negative)
                        const double total = lr < 0.0 ? (double)kk + need / lr : 1e30;
                        if (total >= 1.75 * (double)o.mg_switch_iterations && total >= 2.0 * (double)kk) p->mg_switch_at = std::min(p->mg_switch_at, k);
                    }
                }
            }
        }
        ++n_chunks;
        // Hybrid preconditioning: most LM systems (small trust regions, steps about to be rejected) are solved by block-Jacobi in a few
        // hundred cheap iterations; one that is not done after mg_switch_iterations is a hard one, and from there the multigrid (4x fewer
        // iterations or better at ~3x the price) takes over: operators built now, PCG restarted from the current iterate.
        if (!done && p->mg_built && !p->mg_active && !p->mg_failed && k >= p->mg_switch_at && k < o.cg_max_iterations) {     // (several ranks: every quantity tested here is the same on all of them)
            HIPCHK(p, hipMemcpyAsync(hflags, p->C.flags, sizeof(hflags), hipMemcpyDeviceToHost, p->st));
            HIPCHK(p, hipStreamSynchronize(p->st));
            if (hflags[0]) { done = true; (void)enqueue_poll(n_chunks & 1); ++n_chunks; break; }
            if ((rc = switch_to_mg(hflags[2])) != PGO_OK) return rc;
        }
    }
    if (n_chunks > 0) {   // the state after the LAST enqueued chunk is the final one (kernels past convergence do nothing)
        HIPCHK(p, hipEventSynchronize(p->poll_ev[(n_chunks - 1) & 1]));
        std::memcpy(hflags, p->poll[(n_chunks - 1) & 1].flags, sizeof(hflags));
        std::memcpy(hscal, p->poll[(n_chunks - 1) & 1].scal, sizeof(hscal));
    } else {
        HIPCHK(p, hipMemcpyAsync(hflags, p->C.flags, sizeof(hflags), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipMemcpyAsync(hscal, p->C.scal, sizeof(hscal), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
    }
    res->converged = hflags[0] != 0 && hflags[1] == 0;
    if (!hflags[0] && !multi) {   // iteration cap reached: one more convergence test so that scal[1] holds the last r.z (x is already final)
        launch_cg_set_tolerance(p->C, 1e300, p->st);
        if (sr) { if ((rc = one_iteration(k)) != PGO_OK) return rc; }      // (its update's head finds r.u below the tolerance: scal[1] <- r.u, nothing else moves)
        else if (fused_coarse) launch_mf_spmv_coarse(p->G, p->F, p->Sc, p->C, p->K, k, 1e300, fused_parts, k > 0 ? 1 : 0, p->st);
        else if (p->built_mf) launch_mf_spmv(p->G, p->F, p->Sc, p->C, k, 1e300, p->st);
        else launch_cg_spmv(p->G, p->C, k, 1e300, p->st);
        HIPCHK(p, hipMemcpyAsync(hflags, p->C.flags, sizeof(hflags), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipMemcpyAsync(hscal, p->C.scal, sizeof(hscal), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
    }
    res->iterations = hflags[2];
    res->breakdown = hflags[1] != 0;
    res->rel_residual = hscal[0] > 0 ? std::sqrt(std::max(0.0, hscal[1]) / hscal[0]) : 0.0;
    return PGO_OK;
}

// Coarse operator of the two-level preconditioner for the system just built: Ac = P^T A P (deterministic assembly) and its dense inverse
// (blocked Gauss-Jordan kernels).  A coarse operator that is not numerically positive definite leaves the coarse space off for this iteration.
static int build_coarse(pgo_problem* p) {
    p->coarse_active = false;
    const double t_coarse0 = now_s();
    // Where it pays: always when the aggregates are small (the coarse space is then a sizeable fraction of the problem: graphs up to
    // ~64 x coarse_aggregates keyframes), otherwise only at large trust regions, where the slow modes are the long wavelengths
    // (measured: scripts/gpu_coarse_ab.py).
    if (!p->coarse_built || p->opt.coarse_aggregates <= 0) return PGO_OK;
    if (p->coarse_mode == 2) {
        // dropped at a smaller trust region: the long wavelengths it removes dominate more and more as the radius grows, so it gets another
        // comparison once the radius is 9x (two accepted steps) beyond the one it lost at — at most twice per solve
        if (p->coarse_retests >= 2 || p->coarse_skip_all || !(p->radius >= 9.0 * p->coarse_drop_radius)) return PGO_OK;   // (eligibility by aggregate size / coarse_min_radius is checked below)
        ++p->coarse_retests; p->coarse_mode = 0;
    }
    if (!(p->K.m <= 64 || (p->radius >= p->opt.coarse_min_radius && p->K.m <= 1024))) return PGO_OK;   // aggregates of thousands of keyframes are too coarse to help
    if (p->coarse_geometry_epoch != p->lin_epoch) {          // the aggregates' centroids follow the poses of the current linearisation
        launch_coarse_geometry(p->G, p->K, p->d_pose[p->cur].p, p->st);
        p->coarse_geometry_epoch = p->lin_epoch;
    }
    launch_coarse_assemble(p->G, p->L, p->Sc, p->C, p->K, p->st);
    int32_t* fail = p->d_cinfo.p;
    HIPCHK(p, hipMemsetAsync(fail, 0, sizeof(int32_t), p->st));
    launch_coarse_invert(p->K, p->d_cscr.p, fail, p->st);
    if (debug_break_coarse()) launch_coarse_negate(p->K, p->st);
    int32_t h = 1;
    HIPCHK(p, hipMemcpyAsync(&h, fail, sizeof(h), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    p->coarse_active = h == 0;
    if (p->opt.verbosity > 1) std::fprintf(stderr, "[pgo] coarse operator assembled and inverted in %.3f ms (since the start of build_coarse)\n", (now_s() - t_coarse0) * 1e3);
    if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] coarse space: %d aggregates of %d keyframes, %d blocks, radius %.1e: %s\n", p->K.n_agg, p->K.m, p->K.n_blk, p->radius, h == 0 ? "on" : "coarse operator not positive definite -> off");
    return PGO_OK;
}

// Multigrid operators of the system just built: Galerkin products level by level, block-Jacobi inverses, dense inverse of the coarsest level.
// A block that is not numerically positive definite leaves the multigrid off for this LM iteration (plain block-Jacobi).
// Regroup: the hierarchy was built from the switch values of its time (0.99 everywhere at the first solve of a graph).  A few LM steps later the solver has
// switched the outliers off, and aggregates of the levels above level 1 that such a loop closure held together are no rigid pieces any more: measured on C3, the
// late systems need 305 / 367 / 454 multigrid iterations with the hierarchy of the start against 156 / 187 / 249 with one built from the final switch values.  So when
// multigrid operators are about to be built and the switch values have moved far from the hierarchy's (switchable edges that moved by > 0.5 in s^2 make up more than mg_regroup_fraction of ALL
// edges), the levels above level 1 are matched again along the couplings alive NOW (the keyframes' level-1 aggregates, matched along relative-pose edges only, and the
// level-1 structure are cached: pgo_mg::BuildCache) — at most twice per solve.  Several ranks: the count is all-reduced, every rank regroups at the same LM step.
// how many switchable edges have moved by > 0.5 in s^2 since the hierarchy was matched, against ALL residual blocks (what counts is how much of the coupling structure
// changed: C4 has 2 % loop closures — no regroup pays there); summed over the ranks
static int regroup_count(pgo_problem* p, const double* sv, std::vector<double>& cnt, double moved_by = 0.5) {
    const int64_t Es = p->swe.size();
    cnt.assign(2, 0.0);
    for (int64_t e = 0; e < Es; ++e) { const double w = sv[p->swe.sw[e]] * sv[p->swe.sw[e]]; if (std::fabs(w - p->mg_sw_built[e]) > moved_by) cnt[0] += 1.0; }
    cnt[1] = (double)(Es + p->rel.size());
    return p->local_ids ? host_allreduce(p, cnt, 0) : PGO_OK;
}
// A regroup is TRANSACTIONAL: the hierarchy in place is replaced only by one that coarsened; when the matching along the current couplings stalls (build_hierarchy
// gives up above 0.85 nodes per node, or runs out of levels) the installed hierarchy stays — with the new switch record, so that the same failing attempt is not
// repeated at every later check — instead of the handle silently falling back to plain block-Jacobi for the rest of its life.
static int regroup_commit(pgo_problem* p, MgPrepared& Q) {
    if (!Q.ok) {
        if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] multigrid: the regrouped hierarchy does not coarsen -> the one in place stays\n");
        p->mg_sw_built.swap(Q.sw_built);
        return PGO_OK;
    }
    int rc;
    HIPCHK(p, hipStreamSynchronize(p->st));
    if ((rc = mg_install(p, Q)) != PGO_OK) return rc;
    ++p->build_epoch;      // captured PCG chunks hold pointers into the old pools
    return PGO_OK;
}
static int regroup_if_moved(pgo_problem* p, const double* sv /* host: the caller's switch array */, bool in_solve) {
    std::vector<double> cnt;
    int rc;
    if ((rc = regroup_count(p, sv, cnt, in_solve ? 0.5 : 0.0)) != PGO_OK) return rc;
    // inside a solve: once the moved edges are a sizeable part of the coupling structure.  At the START of a solve: whenever ANY switch differs from the record — the
    // hierarchy a solve starts with is then a function of the graph and of the solve's own start values alone, whatever earlier solves of the handle left behind
    // (pgo.h: "no per-handle history"; tests/test_gpu_determinism.py solves from state A after a solve that regrouped and compares with a fresh handle, bit for bit).
    if (in_solve ? !(cnt[0] > p->opt.mg_regroup_fraction * cnt[1]) : !(cnt[0] > 0.0)) return PGO_OK;
    const double t0 = now_s();
    mg_job_cancel(p);
    MgPrepared Q;
    if ((rc = mg_prepare(p, sv, Q)) != PGO_OK) return rc;
    if ((rc = regroup_commit(p, Q)) != PGO_OK) return rc;
    if (in_solve) ++p->mg_regroups;
    if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] multigrid: regrouped %s (%.0f switchable edges of %.0f edges moved), %.1f ms\n", in_solve ? "inside the solve" : "for the new start", cnt[0], cnt[1], (now_s() - t0) * 1e3);
    return PGO_OK;
}
static bool regroup_allowed(const pgo_problem* p) {
    // (not during the first three LM iterations: the switches of outliers — and of inliers far from the odometry guess, which recover — are still falling then: measured on C3,
    // 17 % of the switchable edges have moved after the first step, and a regroup there is paid twice)
    return p->opt.mg_regroup_fraction > 0.0 && p->mg_built && p->S > 0 && p->mg_regroups < 2 && p->iteration >= 3 && (int64_t)p->mg_sw_built.size() == p->swe.size();
}
// several ranks: the regroup happens where multigrid operators are about to be built, synchronously (its host half holds collectives) — every rank at the same LM step
static int maybe_regroup(pgo_problem* p) {
    if (!regroup_allowed(p) || p->iteration <= 3) return PGO_OK;
    std::vector<double> sv((size_t)p->S);
    HIPCHK(p, hipMemcpyAsync(sv.data(), p->d_swv[p->cur].p, sv.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return regroup_if_moved(p, sv.data(), true);
}
// One GPU: the HOST half of a regroup (≈25 ms for C3: matching of the upper levels, structures of the smoothed transition, pooled arrays) starts on a worker thread right
// after the accepted step that moved the switches far enough, and is installed where multigrid operators are next built (regroup_install) — on C3 that is a dozen cheap
// block-Jacobi LM steps later, so the solve never waits for it.  Which step starts it and which step installs it depend on the solve's own history only.
static int regroup_start(pgo_problem* p) {
    if (p->local_ids || p->mg_job_running || !regroup_allowed(p)) return PGO_OK;
    std::vector<double> sv((size_t)p->S);
    HIPCHK(p, hipMemcpyAsync(sv.data(), p->d_swv[p->cur].p, sv.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    std::vector<double> cnt;
    int rc;
    if ((rc = regroup_count(p, sv.data(), cnt)) != PGO_OK) return rc;
    if (!(cnt[0] > p->opt.mg_regroup_fraction * cnt[1])) return PGO_OK;
    ++p->mg_regroups;
    p->mg_job_out.reset(new MgPrepared());
    p->mg_job_out->moved = cnt[0]; p->mg_job_out->of_edges = cnt[1];
    p->mg_job_running = true; p->rc_job = PGO_OK;
    MgPrepared* Q = p->mg_job_out.get();
    MgPrepared* old_image = p->mg_job_old.release();
    try {
        p->mg_job = std::thread([p, Q, old_image, sv]() { delete old_image; p->rc_job = mg_prepare(p, sv.data(), *Q); });
    } catch (...) {      // no thread to be had (the C-ABI never throws): the same work on this one
        delete old_image;
        p->rc_job = mg_prepare(p, sv.data(), *Q);
    }
    return PGO_OK;
}
static int regroup_install(pgo_problem* p) {
    if (!p->mg_job_running) return PGO_OK;
    const double t0 = now_s();
    if (p->mg_job.joinable()) p->mg_job.join();
    p->mg_job_running = false;
    std::unique_ptr<MgPrepared> Q = std::move(p->mg_job_out);
    if (p->rc_job != PGO_OK || !Q) return p->rc_job;
    const double waited = (now_s() - t0) * 1e3;
    int rc;
    if ((rc = regroup_commit(p, *Q)) != PGO_OK) return rc;
    // The host image is NOT freed here: it was allocated by the worker thread (an mmap-backed malloc arena), and returning ~100 MB of it to the system from this thread
    // costs 5 ms of munmap plus a ~10 ms stall of the next kernels (measured: MMU-notifier invalidations reach the GPU's address space).  It is kept until the next
    // regroup's worker (or pgo_destroy) drops it, off the solve's critical path.
    p->mg_job_old = std::move(Q);
    MgPrepared* Qk = p->mg_job_old.get();
    if (p->opt.verbosity > 0) std::fprintf(stderr, "[pgo] multigrid: regrouped inside the solve (%.0f switchable edges of %.0f edges moved): host half %.1f ms on a worker thread, waited %.1f ms, installed in %.1f ms\n",
                                           Qk->moved, Qk->of_edges, Qk->host_ms, waited, (now_s() - t0) * 1e3 - waited);
    return PGO_OK;
}

// Several ranks, distributed set-up (round 6): the operators of the current LM system with every DISTRIBUTED level formed by its rows' owners.
//   level 1:  every rank's part of the Galerkin product from its own edges and owned keyframes (as before), then — instead of the all-reduce of ALL of level 1's blocks — the
//             parts of the blocks two ranks share go to the ranks that need them (BlockPlan: summed in ascending rank order)
//   level l distributed:  block-Jacobi inverses, the fp32 copy, the smoother's safety estimate (the whole level's eight power steps on the owners' rows: the iterate's halo before
//             every step, one 3-double all-reduce of the norms and the failure flag — a failed block counts for all ranks); a smoothed transition above it: Dinv of the halo rows (the cycle forms x = Dinv r on receipt), Ps on its own rows, the rows of Ps its rows of W = A Ps
//             multiply from their owners, W and R^T = Ps - Dinv W on its own rows, the blocks of R whose coarse row is another rank's to that rank, its rows' part of Ps^T W to
//             the needers; a plain transition: P^T A P on its own rows (children are the parent's rank's), the blocks above the diagonal also to the column's owner
//   the first level every rank runs completely:  formed like that by its rows' owners, gathered by all; from there on every rank forms the same small levels and the dense inverse
// Nothing here is replicated that grows with the graph: under weak scaling a rank's set-up stays its share + the small top.
static int build_mg_ranks(pgo_problem* p, double omega, int32_t* fail, bool hoff_valid, bool kernels_only = false /* pgo_time_kernel(8): this rank's kernels without the exchanges (the numbers are then meaningless) */) {
    const int fw = p->mg_first_whole;
    int rc;
    launch_mg_galerkin0(p->G, p->L, p->Sc, p->C, p->M, p->mg_levels, p->st, hoff_valid);
    if (!kernels_only && (rc = exchange_blocks_sum(p, p->mg_setup.val[0], p->su_plan[0], p->mg_levels[0].val)) != PGO_OK) return rc;
    for (int l = 0; l < fw; ++l) {
        MgLevelDev& A = p->mg_levels[l];
        MgLevelDev& B = p->mg_levels[l + 1];
        launch_mg_level_inverses(A, omega, fail, p->st);
        {   // the smoother's safety estimate: the whole level's power method, the iterate's halo exchanged before every step, the norms (and the failure flag) summed over the ranks
            launch_mg_power_init(A, p->st);
            double* v = A.x; double* w = A.xt;
            for (int it = 0; it < 8; ++it) {
                if (!kernels_only && (rc = exchange_level(p, l, v, nullptr, nullptr, nullptr)) != PGO_OK) return rc;
                launch_mg_power_step(A, v, w, omega, p->st);
                std::swap(v, w);
            }
            launch_mg_power_sums(A, w, v, fail, p->d_xscal.p + 8, p->st);      // (v: 8 steps, w: 7 steps)
            if (!kernels_only && (rc = allreduce(p, p->d_xscal.p + 8, 3, 0)) != PGO_OK) return rc;
            launch_mg_power_finish(p->d_xscal.p + 8, fail, A.xf, p->st);
            launch_mg_level_rescale(A, A.xf, omega, p->st);
        }
        if (A.smoothed) {
            const pgo_problem::LevelPlanDev& LP = p->lvl_plan[(size_t)l];
            if (!kernels_only && LP.plan && (rc = exchange_blocks_copy(p, *LP.plan, LP.send_idx, LP.recv_idx, A.Dinv, 36)) != PGO_OK) return rc;
            launch_mg_transition_ps(A, mg_cs(p), p->st);
            if (!kernels_only && (rc = exchange_blocks_copy(p, p->mg_setup.ps[(size_t)l], p->su_plan[(size_t)l].ps_send, p->su_plan[(size_t)l].ps_recv, A.ps_val, 36)) != PGO_OK) return rc;
            launch_mg_transition_w(A, p->st);
            if (!kernels_only && (rc = exchange_blocks_copy(p, p->mg_setup.rv[(size_t)l], p->su_plan[(size_t)l].rv_send, p->su_plan[(size_t)l].rv_recv, reinterpret_cast<double*>(A.r_valf), 18)) != PGO_OK) return rc;
            launch_mg_transition_product(A, B, p->st);
        } else if (l + 1 == fw) {      // the first level every rank runs completely: its own rows here, the rest by the gather below
            MgLevelDev Bo = B;
            Bo.su_row0 = p->mg_fw_row0; Bo.su_row1 = p->mg_fw_row1; Bo.su_blk0 = p->mg_fw_blk0; Bo.su_blk1 = p->mg_fw_blk1;
            launch_mg_level_galerkin(A, Bo, p->st);
        } else launch_mg_level_galerkin(A, B, p->st);
        if (!kernels_only && (rc = exchange_blocks_sum(p, p->mg_setup.val[(size_t)l + 1], p->su_plan[(size_t)l + 1], B.val)) != PGO_OK) return rc;
    }
    launch_mg_assemble_rest(p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p), fw);
    return PGO_OK;
}

static int build_mg(pgo_problem* p) {
    p->mg_active = false;
    if (!p->mg_built) return PGO_OK;
    { int rci; if ((rci = mg_init_finish(p)) != PGO_OK) return rci; }      // the hierarchy of a fresh graph build is installed where it is first needed
    if (!p->mg_built) return PGO_OK;
    const double t_build0 = now_s();
    { int rcr; if ((rcr = p->local_ids ? maybe_regroup(p) : regroup_install(p)) != PGO_OK) return rcr; }
    if (!p->mg_built) return PGO_OK;
    if (p->opt.verbosity > 1) std::fprintf(stderr, "[pgo] multigrid: build_mg past the regroup at %.2f ms\n", (now_s() - t_build0) * 1e3);
    int rcm;
    if (p->mg_geometry_epoch != p->lin_epoch) {              // the aggregates' centroids follow the poses of the current linearisation
        if (p->local_ids) {     // a level-1 node's keyframes live on several ranks: owner-weighted position sums, one all-reduce, then as on one GPU
            launch_mg_geometry0_sum(p->G, p->M, p->mg_levels, p->d_pose[p->cur].p, p->st);
            if ((rcm = allreduce(p, p->mg_levels[0].pos, (size_t)p->M.n1 * 3, 0)) != PGO_OK) return rcm;
            launch_mg_geometry_finish(p->G, p->M, p->mg_levels, p->d_pose[p->cur].p, p->st);
        } else launch_mg_geometry(p->G, p->M, p->mg_levels, p->d_pose[p->cur].p, p->st);
        p->mg_geometry_epoch = p->lin_epoch;
    }
    int32_t* fail = p->d_cinfo.p;
    HIPCHK(p, hipMemsetAsync(fail, 0, sizeof(int32_t), p->st));
    const double omega = p->opt.mg_omega > 0.0 && p->opt.mg_omega <= 1.0 ? p->opt.mg_omega : 0.9;
    // level 1's Galerkin product reads J1^T J2 of every edge: the block-CSR solver has them from K2; under the matrix-free solver they are formed here, once per linearisation
    // that builds multigrid operators (an edge-parallel pass whose Jacobian loads coalesce, ~60 us on C3 — the wavefront-per-block product gathering K1's Jacobians itself,
    // twelve strided loads per lane and contribution, took 0.9 ms)
    bool hoff_valid = !p->built_mf;
    if (p->built_mf && p->d_Hoff.cap >= (size_t)(p->G.rel.Epad + p->G.sw.Epad) * 36) {
        if (p->hoff_epoch != p->lin_epoch) { p->L.Hoff = p->d_Hoff.p; launch_k2_offdiag(p->G, p->L, p->st); p->hoff_epoch = p->lin_epoch; }
        hoff_valid = true;
    }
    if (p->local_ids && p->mg_first_whole > 0) {
        if ((rcm = build_mg_ranks(p, omega, fail, hoff_valid)) != PGO_OK) return rcm;
    } else if (p->local_ids) {  // level 1 = the sum of the ranks' Galerkin products (each edge lives on one rank, each diagonal block is its owner's); the levels above are replicated
        launch_mg_galerkin0(p->G, p->L, p->Sc, p->C, p->M, p->mg_levels, p->st, hoff_valid);
        if ((rcm = allreduce(p, p->mg_levels[0].val, (size_t)p->mg_levels[0].nnzb * 36, 0)) != PGO_OK) return rcm;
        launch_mg_assemble_rest(p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p));
    } else if (p->opt.verbosity > 1) {
        HIPCHK(p, hipStreamSynchronize(p->st)); std::fprintf(stderr, "[pgo] multigrid: geometry done at %.2f ms\n", (now_s() - t_build0) * 1e3);
        launch_mg_galerkin0(p->G, p->L, p->Sc, p->C, p->M, p->mg_levels, p->st, hoff_valid);
        HIPCHK(p, hipStreamSynchronize(p->st)); std::fprintf(stderr, "[pgo] multigrid: galerkin0 done at %.2f ms\n", (now_s() - t_build0) * 1e3);
        launch_mg_assemble_rest(p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p));
    } else if (p->mg_fine) {      // smoothed keyframe transition: level 1 = Ps_0^T A Ps_0 from the keyframe level's own blocks
        launch_mg_assemble_fine(p->G, p->L, p->Sc, p->C, p->mg_fineF, p->mg_fineT, p->mg_levels[0], omega, fail, p->st, mg_cs(p), hoff_valid, p->d_pose[p->cur].p);
        launch_mg_assemble_rest(p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p));
    } else launch_mg_assemble(p->G, p->L, p->Sc, p->C, p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p), hoff_valid);
    if (p->opt.verbosity > 1) { HIPCHK(p, hipStreamSynchronize(p->st)); std::fprintf(stderr, "[pgo] multigrid: level operators done at %.2f ms\n", (now_s() - t_build0) * 1e3); }
    launch_coarse_invert(p->K, p->d_cscr.p, fail, p->st);
    if (debug_break_coarse()) launch_coarse_negate(p->K, p->st);
    int32_t h = 1;
    HIPCHK(p, hipMemcpyAsync(&h, fail, sizeof(h), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    p->mg_active = h == 0;
    // level 1's up-sweep kernel also prolongs to the keyframes; its workgroups (at most MAX_PARTIALS, each taking every gridDim-th tile) put their r.z partials behind the update kernel's
    // (measured: 1 114 tiles on 1 024 workgroups — C4 — lose 3 % to the ragged second trip against the separate prolongation kernel; 3 907 tiles — C5 — gain 3.5 %)
    const int t1 = p->mg_levels[0].tiles;
    p->C.extra_rz = (p->mg_active && !p->local_ids && !p->mg_fine && p->M.n_levels >= 2 && (t1 <= MAX_PARTIALS || t1 >= 2 * MAX_PARTIALS)) ? std::min<int>(t1, MAX_PARTIALS) : 0;
    if (p->opt.verbosity > 0 && h != 0) std::fprintf(stderr, "[pgo] multigrid: a coarse block is not positive definite at radius %.1e -> off for this iteration\n", p->radius);
    if (p->opt.verbosity > 1) std::fprintf(stderr, "[pgo] multigrid: operators of LM iteration %d built in %.2f ms\n", p->iteration, (now_s() - t_build0) * 1e3);
    return PGO_OK;
}

int build_system(pgo_problem* p, bool* ok) {
    int rc;
    HIPCHK(p, hipMemsetAsync(p->d_flags.p + 4, 0, sizeof(int32_t), p->st));
    launch_build_rows(p->G, p->L, p->Sc, p->C, p->radius, 1 /*one GPU: this handle adds Hd, g and the damping; multi-GPU: the keyframe's owner (G.own)*/, p->built_mf ? p->d_lam.p : nullptr, p->st);
    if ((rc = exchange_rows(p, p->C.Dtot, 36, p->C.b, 6)) != PGO_OK) return rc;   // reduced diagonal + rhs of shared keyframes
    launch_invert_rows(p->G, p->C, p->d_flags.p + 4, p->st);
    int32_t fail = 0;
    HIPCHK(p, hipMemcpyAsync(&fail, p->d_flags.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    if (p->local_ids) {      // a block that fails on one rank makes the step invalid on all of them (the ranks must take the same branch: collectives follow)
        std::vector<double> f(1, fail ? 1.0 : 0.0);
        if ((rc = host_allreduce(p, f, 2)) != PGO_OK) return rc;
        fail = f[0] != 0.0;
    }
    *ok = fail == 0;
    p->mg_active = false; p->mg_failed = false; p->C.extra_rz = 0; p->mg_start_deferred = false;
    // block-Jacobi-equivalent iterations this system is expected to need: those of the last fully solved system of this solve x sqrt(radius ratio); 0 = no prediction
    p->cg_predicted = (p->cg_prev_radius > 0.0 && p->radius > 0.0) ? p->cg_prev_equiv * std::sqrt(p->radius / p->cg_prev_radius) : 0.0;
    if (*ok && p->mg_built) {
        // Which preconditioner the PCG of this LM system starts with.  Block-Jacobi iterations grow like sqrt(radius) from one accepted step
        // to the next, so the previous step of this solve predicts this one (a multigrid iteration counts as 4 block-Jacobi ones: it costs
        // ~2.5x and saves 4x or more on hard systems):  predicted >= 1.75 x mg_switch_iterations -> multigrid (from the first iteration, or after the prelude below);
        // predicted easier than that -> block-Jacobi, and the in-flight switch of run_pcg waits for twice the prediction (switching 400
        // iterations into a system that needs 520 throws the work away); no prediction (first step, after a rejected one) -> block-Jacobi with
        // the switch at mg_switch_iterations.  Depends on this solve's own history only.
        double predicted = 0.0;
        // (round 3, with the smoothed cycle: start factors 1.0 - 2.25, waiting factors 1.5 - 2.0 and switch points 200 - 600 all within +-2 % on C3 and C4)
        // (with the deferred start and the regroup off the critical path, session 2 of round 3: start 1.0 - 2.25 x wait 1.5 / 2.0 on C3 0.311 - 0.333 s, C4 1.498 - 1.542 s; 1.75 / 2.0 is the best pair on both)
        const double start_factor = 1.75, wait_factor = 2.0;
        if (p->cg_prev_radius > 0.0 && p->radius > 0.0) predicted = p->cg_prev_equiv * std::sqrt(p->radius / p->cg_prev_radius);
        p->mg_switch_at = p->opt.mg_switch_iterations;
        if (predicted > 0.0 && predicted < start_factor * (double)p->opt.mg_switch_iterations) p->mg_switch_at = std::max(p->opt.mg_switch_iterations, (int)(wait_factor * predicted));
        // A system predicted hard gets the multigrid from the first iteration — unless the step can still be rejected early: most steps that ARE rejected follow a long
        // accepted one at a large radius, i.e. exactly the systems predicted hard, and block-Jacobi reaches the first pause (cg_early_tolerance, a few dozen iterations)
        // for a fraction of what the operators cost (C3, step 4: 32 ms for a step thrown away at 21 iterations).  Then the build waits for the pause (lm_step).
        const bool hard = p->opt.mg_switch_iterations <= 0 || predicted >= start_factor * (double)p->opt.mg_switch_iterations;
        // ... and only where a rejection is in the air: the previous step was rejected (rejections come in streaks: the radius shrinks over several steps), or the last accepted
        // step's relative decrease fell below 0.8 — the quadratic model is losing its grip (C3's and C4's first rejected steps follow rho = 0.67 and 0.62; the accepted hard steps
        // of both follow rho >= 0.89, and a prelude there is 74 block-Jacobi iterations the multigrid would not have needed: 3 ms x 5 on C3, 5 ms x 12 on C4)
        const bool rejection_likely = p->reuse_diagonal || p->last_rho < 0.8;
        p->mg_start_deferred = hard && rejection_likely && p->opt.mg_switch_iterations > 0 && p->opt.cg_early_tolerance > p->opt.cg_rel_tolerance;
        if (p->mg_start_deferred) {      // ... but not for long: a step that has not reached the pause within the prelude is a hard one that stays (late C3 systems need ~300 block-Jacobi iterations to 1e-2)
            const int prelude = 72;      // three chunks (measured on C3 / C4, 20 steps: 48 -> 0.392 / 1.500 s — C3's rejected step 4 needs 53 —, 72 -> 0.322 / 1.515 s, 96 -> 0.324 / 1.524 s)
            p->mg_switch_at = std::min(p->mg_switch_at, prelude);
        }
        if (hard && !p->mg_start_deferred && (rc = build_mg(p)) != PGO_OK) return rc;
    }
    else if (*ok && (rc = build_coarse(p)) != PGO_OK) return rc;
    return PGO_OK;
}

const char* step_reason_text(int r) {
    static const char* const t[] = {"ok", "REJ(rho)", "REJ(pause)", "INVALID(factorization)", "INVALID(breakdown)", "INVALID(model)", "CONVERGED"};
    return r >= 0 && r < 7 ? t[r] : "?";
}

void log_iter(pgo_problem* p, const pgo_iteration& it) {
    if (p->sum.num_logged < PGO_MAX_ITERATION_LOG) p->sum.iterations[p->sum.num_logged++] = it;
    if (p->opt.verbosity > 0)
        std::fprintf(stderr, "[pgo] it %3d cost %.12e dcost %.3e rho %.3e |step| %.3e radius %.3e cg %d (%.1e) %s %.2f ms\n", it.iteration, it.cost, it.cost_change,
                     it.relative_decrease, it.step_norm, it.trust_region_radius, it.cg_iterations, it.cg_residual, step_reason_text(it.reason), it.seconds * 1e3);
}

void terminate(pgo_problem* p, int type, const char* msg) {
    p->terminated = true;
    p->sum.termination_type = type;
    std::snprintf(p->sum.message, sizeof(p->sum.message), "%s", msg);
}

int solve_begin(pgo_problem* p, const double* quat, const double* t, const double* sw, int64_t N, int64_t S) {
    if (!quat || !t || N <= 0 || S < 0 || (S > 0 && !sw)) { p->err = "null state array or bad size"; return PGO_ERR_INVALID_ARG; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    p->t_begin = now_s();
    mg_job_cancel(p);
    const bool rebuild = p->graph_dirty || p->priors_dirty || N != p->N_global || S != p->S;
    if (rebuild) { if ((rc = build_graph(p, N, S, sw)) != PGO_OK) return rc; }
    else if ((rc = mg_init_finish(p)) != PGO_OK) return rc;      // (an unchanged graph whose hierarchy no solve has needed yet: installed, then compared with this solve's start values)
    if (!rebuild && p->opt.mg_regroup_fraction > 0.0 && p->mg_built && S > 0 && sw && (int64_t)p->mg_sw_built.size() == p->swe.size()) {
        // the hierarchy of an unchanged graph was built (or regrouped inside the last solve) for other switch values than this solve starts from: the levels above level 1
        // are rebuilt for the start values whenever ANY switch differs from the record (regroup_if_moved, moved_by = 0) — a synchronous mg_prepare + mg_install, tens of
        // milliseconds on C3 — so that repeated solves from the same state stay bitwise identical whatever the handle solved before.  What this costs in practice: a session's
        // next trigger has a NEW graph (one more loop edge: full rebuild anyway); only a re-solve of an unchanged graph from other switch values pays it.  (The matching
        // depends on the switch values continuously — coupling strengths order the heavy-edge matching — so "nearly the same switches" is not a safe reason to keep a hierarchy.)
        // Exception, stated: when the rebuilt hierarchy does not coarsen the one in place stays (regroup_commit) with the new switch record; the starting hierarchy then
        // depends on the handle's history.  No graph of the test suite or of profiles/ reaches that branch at a solve's start.
        if ((rc = regroup_if_moved(p, sw, false)) != PGO_OK) return rc;
    }
    // upload in the reference layout (multi-GPU: only this rank's keyframes), repack on the device
    double* io = p->d_io.p;
    const int64_t Nl = p->N;
    if (p->local_ids) { p->h_init_q.assign(quat, quat + (size_t)N * 4); p->h_init_t.assign(t, t + (size_t)N * 3); }
    if ((rc = nodes_from_global(p, quat, 4, io)) != PGO_OK) return rc;
    if ((rc = nodes_from_global(p, t, 3, io + (size_t)Nl * 4)) != PGO_OK) return rc;
    p->cur = 0;
    launch_pack_pose(io, io + (size_t)Nl * 4, p->d_pose[0].p, Nl, p->st);
    if (S > 0) {
        HIPCHK(p, hipMemcpyAsync(p->d_swv[0].p, sw, (size_t)S * sizeof(double), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemcpyAsync(p->d_swv[1].p, p->d_swv[0].p, (size_t)S * sizeof(double), hipMemcpyDeviceToDevice, p->st));
    }
    HIPCHK(p, hipStreamSynchronize(p->st));
    p->t_device0 = now_s();
    std::memset(&p->sum, 0, sizeof(p->sum));
    p->in_solve = true; p->terminated = false; p->scale_ready = false; p->have_prev_step = false;
    p->st_exchanges = p->st_allreduces = p->st_pcg_iterations = 0; p->st_bytes_neighbour = p->st_bytes_allreduce = 0.0;
    p->coarse_retests = 0; p->coarse_drop_radius = 0.0;
    p->cg_prev_equiv = 0.0; p->cg_prev_radius = 0.0; p->mg_regroups = 0; p->last_rho = 1.0;
    if (p->coarse_skip > 0) { p->coarse_mode = 2; p->coarse_skip_all = true; --p->coarse_skip; }
    else { p->coarse_mode = (p->coarse_keep_streak % 4 != 0) ? 1 : 0; p->coarse_skip_all = false; }
    p->radius = p->opt.initial_trust_region_radius; p->decrease_factor = 2.0; p->reuse_diagonal = false; p->iteration = 0; p->invalid = 0;
    p->sum.termination_type = PGO_NO_CONVERGENCE;
    if ((rc = linearize(p, &p->x_cost)) != PGO_OK) return rc;
    p->sum.initial_cost = p->x_cost;
    p->sum.final_cost = p->x_cost;
    if (!std::isfinite(p->x_cost)) { terminate(p, PGO_FAILURE, "initial cost is not finite"); return PGO_OK; }
    pgo_iteration it{};
    it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = p->x_cost; it.gradient_max_norm = p->gmax; it.trust_region_radius = p->radius;
    it.seconds = now_s() - p->t_device0;
    log_iter(p, it);
    return PGO_OK;
}

int lm_step(pgo_problem* p, int ignore_termination, int* done) {
    if (!p->in_solve) { p->err = "pgo_lm_step before pgo_solve_begin"; return PGO_ERR_STATE; }
    const pgo_options& o = p->opt;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    if (p->terminated && !ignore_termination) { if (done) *done = 1; return PGO_OK; }
    if (p->sum.termination_type == PGO_FAILURE && p->terminated) { if (done) *done = 1; return PGO_OK; }
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (!ignore_termination) {
        if (p->iteration >= o.max_num_iterations) { terminate(p, PGO_NO_CONVERGENCE, "Maximum number of iterations reached."); if (done) *done = 1; return PGO_OK; }
        if (p->gmax <= o.gradient_tolerance) { terminate(p, PGO_CONVERGENCE, "Gradient tolerance reached."); if (done) *done = 1; return PGO_OK; }
        if (p->radius < o.min_trust_region_radius) { terminate(p, PGO_CONVERGENCE, "Minimum trust region radius reached."); if (done) *done = 1; return PGO_OK; }
    }
    const double t0 = now_s();
    ++p->iteration;
    pgo_iteration it{};
    it.iteration = p->iteration; it.trust_region_radius = p->radius;
    if (!p->reuse_diagonal) launch_lm_diag(p->G, p->L, p->Sc, o.min_lm_diagonal, o.max_lm_diagonal, p->st);
    bool ok = true;
    if ((rc = build_system(p, &ok)) != PGO_OK) return rc;
    const double t_built = now_s();
    int why_invalid = ok ? PGO_STEP_ACCEPTED : PGO_STEP_INVALID_FACTORIZATION;      // pgo_iteration.reason of an invalid step
    int precond_used = PGO_PRECOND_BLOCK_JACOBI;
    CgResult cg{0, false, 0.0, false};
    p->cg_extra = 0;
    const int nxt = p->cur ^ 1;
    double h[S_N] = {0};
    // candidate point x (+) delta, its cost, the model cost change and the step norms -> h[]
    auto evaluate_candidate = [&]() -> int {
        int np = 0, np2 = 0, r2;
        launch_model_change(p->G, p->L, p->Sc, p->C.x, p->d_delta_s.p, part(p, 4), &np, p->st);
        launch_reduce(part(p, 4), np, 0, p->d_scal.p + S_MODEL, p->st);
        launch_plus(p->G, p->d_pose[p->cur].p, p->d_swv[p->cur].p, p->C.x, p->d_delta_s.p, p->d_pose[nxt].p, p->d_swv[nxt].p, part(p, 1), part(p, 2), &np2, p->st);
        launch_reduce(part(p, 1), np2, 0, p->d_scal.p + S_STEP2, p->st);
        launch_reduce(part(p, 2), np2, 0, p->d_scal.p + S_SW_STEP2, p->st);
        if ((r2 = run_k1(p, nxt, false)) != PGO_OK) return r2;
        HIPCHK(p, hipMemsetAsync(p->d_scal.p + S_SW_XNORM2, 0, 2 * sizeof(double), p->st));   // SW_XNORM2, GMAX unused here
        HIPCHK(p, hipMemsetAsync(p->d_scal.p + S_XNORM2, 0, sizeof(double), p->st));
        return read_scalars(p, h);
    };
    bool evaluated = false;
    if (ok) {
        // The PCG pauses at up to two intermediate tolerances (cg_early_tolerance > cg_mid_tolerance > cg_rel_tolerance).  A rejected step
        // only changes the trust-region radius (Ceres StepRejected), so a step that is already clearly bad at a pause
        // (relative_decrease below the stage's threshold, and neither convergence test would fire) is rejected without paying for the
        // remaining decades; otherwise the same PCG resumes towards the next tolerance.
        struct Stage { double tol, reject_rho; };
        Stage stages[2]; int n_stages = 0;
        // A pause costs one candidate evaluation (~0.1 ms: eight small launches and a host sync) and pays only when a step is rejected on a system whose PCG is expensive.  The
        // reference's own sessions (hundreds to a few thousand keyframes, steps accepted almost throughout, PCGs of 20-100 iterations at ~14 us) only pay: measured 20.8 -> 17.1 ms
        // on a 400-keyframe trigger, 63.2 -> 60.3 ms at 3 000.  So below CG_PAUSE_MIN_KEYFRAMES the pauses are armed by the first rejected step of the solve (a rejection is
        // usually followed by more: the radius shrinks in several steps) — a rule that depends on the solve's own history only.
        constexpr int64_t CG_PAUSE_MIN_KEYFRAMES = 20000;
        // ... and (round 5) in proportion to what they can save.  A pause costs ~0.25 ms (candidate evaluation, host round trips, the PCG's restart out of its hipGraph), and a
        // system whose step is ACCEPTED pays it for nothing: 13 of C3's 20 steps, 2.8 % of its headline.
        //   * both pauses where a rejection is in the air — the rule build_system defers the multigrid by: the previous step was rejected (rejections come in streaks) or the last
        //     accepted step's relative decrease fell below 0.8 (C3's and C4's first rejected steps follow rho = 0.67 and 0.62);
        //   * the FIRST pause alone, as cheap insurance, where the system is expensive enough for one wasted solve to outweigh dozens of pauses: predicted block-Jacobi-equivalent
        //     iterations x keyframes >= 5.6e7, i.e. a solve of >= ~20 ms (a pause pair is 0.5 ms; a block-Jacobi iteration costs ~36 us per 100 000 keyframes).  rho does NOT
        //     predict every rejection: C5's step 8 follows rho = 0.97 and is rejected with rho = -2.0 — 1.87 s of PCG thrown away against 0.25 s with the pause
        //     (profiles/r05_pause_rule.txt); a system without a prediction counts as mg_switch_iterations iterations;
        //   * none elsewhere.  The PCG's own iterates do not depend on where it pauses.
        const bool rejection_likely = p->reuse_diagonal || p->last_rho < 0.8;
        const double predicted_its = p->cg_predicted > 0.0 ? p->cg_predicted : (double)(o.mg_switch_iterations > 0 ? o.mg_switch_iterations : 400);
        const bool expensive = predicted_its * (double)p->N_global >= 5.6e7;
        const bool armed = p->N_global >= CG_PAUSE_MIN_KEYFRAMES || p->sum.num_unsuccessful_steps > 0;
        const bool pauses = armed && (rejection_likely || p->opt.cg_pause_always != 0);
        const bool early_only = armed && !pauses && expensive;
        if ((pauses || early_only) && o.cg_early_tolerance > o.cg_rel_tolerance) stages[n_stages++] = Stage{o.cg_early_tolerance, o.cg_early_reject_rho};
        if (pauses && o.cg_mid_tolerance > o.cg_rel_tolerance && (n_stages == 0 || o.cg_mid_tolerance < stages[0].tol)) stages[n_stages++] = Stage{o.cg_mid_tolerance, o.cg_mid_reject_rho};
        const bool warm = o.cg_warm_start != 0 && p->have_prev_step && p->reuse_diagonal;
        if ((rc = run_pcg(p, &cg, warm, n_stages ? stages[0].tol : o.cg_rel_tolerance, -1)) != PGO_OK) return rc;
        for (int sidx = 0; sidx < n_stages && !cg.breakdown && !evaluated; ++sidx) {
            if ((rc = evaluate_candidate()) != PGO_OK) return rc;
            const double mc = -h[S_MODEL];
            const double cand = 0.5 * (h[S_COST] + h[S_PRIOR_COST]);
            const double dc = p->x_cost - cand;
            const double sn = std::sqrt(h[S_STEP2] + h[S_SW_STEP2]);
            const bool clear_reject = mc > 0.0 && std::isfinite(mc) && std::isfinite(cand) && dc / mc < stages[sidx].reject_rho &&
                                      sn > o.parameter_tolerance * (p->x_norm + o.parameter_tolerance) && std::fabs(dc) > o.function_tolerance * p->x_cost;
            if (clear_reject) evaluated = true;
            else {
                const bool to_mg = p->mg_start_deferred && !p->mg_active;
                p->mg_start_deferred = false;
                if ((rc = run_pcg(p, &cg, false, sidx + 1 < n_stages ? stages[sidx + 1].tol : o.cg_rel_tolerance, cg.iterations, to_mg)) != PGO_OK) return rc;
            }
        }
        // The coarse space pays by a large factor or not at all (it can even cost iterations on chains that odometry weights cut into
        // many loose pieces), so once per solve — at the first full-accuracy step that used it — plain block-Jacobi gets the SAME
        // iteration budget on the same system: if it does not converge within it the coarse space stays for the rest of
        // the solve, otherwise it is dropped.  The test costs at most as many iterations as the coarse run took.
        if (p->coarse_active && p->coarse_mode == 0 && !evaluated && !cg.breakdown && cg.converged) {
            HIPCHK(p, p->d_tmp.ensure((size_t)p->N * 6));
            HIPCHK(p, hipMemcpyAsync(p->d_tmp.p, p->C.x, (size_t)p->N * 6 * sizeof(double), hipMemcpyDeviceToDevice, p->st));
            const int saved_cap = p->opt.cg_max_iterations;
            // an iteration with the coarse space costs 2.6-2.8x a plain one (three more kernels at the latency floor, measured from 200 to
            // 20k keyframes): equal TIME budgets
            p->opt.cg_max_iterations = std::max(3 * cg.iterations, 2 * (std::max(2, p->opt.cg_check_every) & ~1));
            p->coarse_active = false;
            CgResult plain{0, false, 0.0, false};
            rc = run_pcg(p, &plain, false, o.cg_rel_tolerance, -1);
            p->opt.cg_max_iterations = saved_cap;
            if (rc != PGO_OK) return rc;
            if (plain.converged && !plain.breakdown) {      // block-Jacobi alone is at least as fast here
                // lost although it needed clearly fewer iterations: worth another comparison at a larger radius; lost without even that
                // (the aggregates' rigid modes are not this graph's slow modes): no more comparisons in this solve
                if ((double)plain.iterations < 1.2 * (double)cg.iterations) p->coarse_retests = 2;
                p->coarse_mode = 2; cg.iterations += plain.iterations; p->coarse_drop_radius = p->radius;
            }
            else {
                p->coarse_mode = 1; p->coarse_active = true; p->coarse_backoff = 0;
                HIPCHK(p, hipMemcpyAsync(p->C.x, p->d_tmp.p, (size_t)p->N * 6 * sizeof(double), hipMemcpyDeviceToDevice, p->st));
                cg.iterations += plain.iterations;
            }
        }
        // A breakdown under the multigrid (its cycle was not positive definite on this system — a smoother at its stability limit) or under the two-level method (its
        // dense coarse inverse is applied rounded to fp32: at large trust-region radii the coarse operator's condition number exceeds what fp32 resolves, and the rounded
        // inverse need not be positive definite) is not the system's fault: the same system is solved again by plain block-Jacobi before the step may count as invalid.
        // Ceres' exact factorisation never turns a solvable step into an invalid one (reference src/PoseGraphSLAM.cpp:1903).
        precond_used = p->mg_active ? PGO_PRECOND_MULTIGRID : (p->coarse_active ? PGO_PRECOND_TWO_LEVEL : PGO_PRECOND_BLOCK_JACOBI);
        if (cg.breakdown && (p->mg_active || p->coarse_active) && !evaluated) {
            if (o.verbosity > 0) std::fprintf(stderr, "[pgo] %s: PCG breakdown at radius %.1e after %d iterations (preconditioner not positive definite) -> block-Jacobi for this system\n",
                                              p->mg_active ? "multigrid" : "two-level method", p->radius, cg.iterations);
            if (p->mg_active) { p->mg_active = false; p->mg_failed = true; }
            p->coarse_active = false;      // (this system only: build_coarse decides again for the next one)
            p->C.extra_rz = 0;
            p->cg_extra += cg.iterations;
            ++p->sum.pcg_retries;
            precond_used = PGO_PRECOND_BLOCK_JACOBI | PGO_PRECOND_RETRIED;
            // The iterate the broken-down PCG stopped at is a valid starting point (x_k with r_k = b - A x_k; a breakdown leaves x untouched): warm start.  Should that one
            // break down as well (a NaN that reached x), the system is solved from zero.
            if ((rc = run_pcg(p, &cg, true, o.cg_rel_tolerance, -1)) != PGO_OK) return rc;
            if (cg.breakdown) { p->cg_extra += cg.iterations; if ((rc = run_pcg(p, &cg, false, o.cg_rel_tolerance, -1)) != PGO_OK) return rc; }
        }
        p->have_prev_step = !cg.breakdown;
        if (cg.breakdown) { ok = false; why_invalid = PGO_STEP_INVALID_BREAKDOWN; }
        // block-Jacobi-equivalent work of this system, for the next system's choice of preconditioner (build_system)
        if (!evaluated && !cg.breakdown) {
            const double equiv = p->mg_levels[0].smoothed ? 8.0 : 4.0;     // block-Jacobi iterations one multigrid iteration stands for on a hard system
            p->cg_prev_equiv = (double)p->cg_extra + (p->mg_active ? equiv : 1.0) * (double)cg.iterations; p->cg_prev_radius = p->radius;
        }
    }
    it.cg_iterations = cg.iterations + p->cg_extra; it.cg_residual = cg.rel_residual;
    const double t_solved = now_s();
    it.seconds_system = t_built - t0; it.seconds_pcg = t_solved - t_built;
    it.cg_iterations_multigrid = p->mg_active ? cg.iterations : 0; it.single_reduction = ok && single_reduction(p) ? 1 : 0;
    if (o.verbosity > 1) std::fprintf(stderr, "[pgo] it %3d PCG: %d iterations%s after %d with block-Jacobi; system + preconditioner %.3f ms, PCG %.3f ms\n", p->iteration, cg.iterations, p->mg_active ? " with the multigrid" : "", p->cg_extra, (t_built - t0) * 1e3, (t_solved - t_built) * 1e3);
    p->sum.cg_iterations += cg.iterations + p->cg_extra;
    if (p->mg_active) p->sum.cg_iterations_multigrid += cg.iterations;     // iterations before an in-flight switch (cg_extra) ran with block-Jacobi
    if (ok) {
        if (!evaluated && (rc = evaluate_candidate()) != PGO_OK) return rc;
        it.seconds_evaluate = now_s() - t_solved;
        if (o.verbosity > 1) std::fprintf(stderr, "[pgo] it %3d candidate evaluated in %.3f ms\n", p->iteration, it.seconds_evaluate * 1e3);
        it.model_cost_change = -h[S_MODEL];
        if (!(it.model_cost_change > 0.0) || !std::isfinite(it.model_cost_change)) { ok = false; why_invalid = PGO_STEP_INVALID_MODEL; }
    }
    it.preconditioner = precond_used;
    if (!ok) {
        // HandleInvalidStep
        it.step_is_valid = 0; it.cost = p->x_cost; it.gradient_max_norm = p->gmax; it.reason = why_invalid;
        ++p->invalid; ++p->sum.num_unsuccessful_steps;
        if (p->invalid >= o.max_num_consecutive_invalid_steps && !ignore_termination) {
            terminate(p, PGO_FAILURE, "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps.");
            it.seconds = now_s() - t0; log_iter(p, it);
            if (done) *done = 1;
            return PGO_OK;
        }
        p->radius *= 0.5; p->reuse_diagonal = true;   // LevenbergMarquardtStrategy::StepIsInvalid
        it.seconds = now_s() - t0; log_iter(p, it);
        if (done) *done = 0;
        return PGO_OK;
    }
    p->invalid = 0;
    it.step_is_valid = 1;
    const double cand_cost = 0.5 * (h[S_COST] + h[S_PRIOR_COST]);
    it.step_norm = std::sqrt(h[S_STEP2] + h[S_SW_STEP2]);
    it.cost_change = p->x_cost - cand_cost;
    it.relative_decrease = it.cost_change / it.model_cost_change;
    bool stop = false;
    if (!ignore_termination) {
        if (it.step_norm <= o.parameter_tolerance * (p->x_norm + o.parameter_tolerance)) { terminate(p, PGO_CONVERGENCE, "Parameter tolerance reached."); stop = true; }
        else if (std::fabs(it.cost_change) <= o.function_tolerance * p->x_cost) { terminate(p, PGO_CONVERGENCE, "Function tolerance reached."); stop = true; }
    }
    if (stop) {
        it.reason = PGO_STEP_CONVERGED;
        it.cost = p->x_cost; it.gradient_max_norm = p->gmax; it.seconds = now_s() - t0; log_iter(p, it);
        if (done) *done = 1;
        return PGO_OK;
    }
    if (std::isfinite(cand_cost) && it.relative_decrease > o.min_relative_decrease) {
        // HandleSuccessfulStep
        p->cur = nxt;
        double c = 0;
        const double t_lin = now_s();
        if ((rc = linearize(p, &c)) != PGO_OK) return rc;
        it.seconds_linearize = now_s() - t_lin;
        if (o.verbosity > 1) std::fprintf(stderr, "[pgo] it %3d linearised in %.3f ms\n", p->iteration, it.seconds_linearize * 1e3);
        p->x_cost = c;
        it.step_is_successful = 1;
        p->radius = p->radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));   // StepAccepted
        p->radius = std::min(o.max_trust_region_radius, p->radius);
        p->decrease_factor = 2.0; p->reuse_diagonal = false;
        p->last_rho = it.relative_decrease;
        it.reason = PGO_STEP_ACCEPTED;
        ++p->sum.num_successful_steps;
        if ((rc = regroup_start(p)) != PGO_OK) return rc;     // the switches have moved: does the hierarchy above level 1 still fit them?
    } else {
        p->radius = p->radius / p->decrease_factor; p->decrease_factor *= 2.0; p->reuse_diagonal = true;   // StepRejected
        it.reason = evaluated ? PGO_STEP_REJECTED_AT_PAUSE : PGO_STEP_REJECTED_RHO;
        ++p->sum.num_unsuccessful_steps;
    }
    it.cost = p->x_cost; it.gradient_max_norm = p->gmax; it.seconds = now_s() - t0;
    log_iter(p, it);
    p->sum.final_cost = p->x_cost;
    if (done) *done = 0;
    return PGO_OK;
}

int solve_end(pgo_problem* p, double* quat, double* t, double* sw, pgo_summary* out) {
    if (!p->in_solve) { p->err = "pgo_solve_end before pgo_solve_begin"; return PGO_ERR_STATE; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const double t_dev = now_s();
    p->sum.num_iterations = p->iteration;
    p->sum.final_cost = p->x_cost;
    p->sum.seconds_device = t_dev - p->t_device0;
    if (p->sum.termination_type != PGO_FAILURE && quat && t) {
        // single write-back at the very end (reference relies on this: src/PoseGraphSLAM.cpp:1894-1903)
        double* io = p->d_io.p;
        launch_unpack_pose(p->d_pose[p->cur].p, io, io + (size_t)p->N * 4, p->N, p->st);
        const int64_t Ng = p->N_global;
        std::vector<double> hq((size_t)Ng * 4), ht((size_t)Ng * 3), hs((size_t)p->S);
        if (!p->local_ids) {
            HIPCHK(p, hipMemcpyAsync(hq.data(), io, hq.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
            HIPCHK(p, hipMemcpyAsync(ht.data(), io + (size_t)p->N * 4, ht.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
        } else {
            // every keyframe is written by its owner into a zeroed array over all keyframes; one all-reduce replicates the result
            HIPCHK(p, p->d_tmp.ensure((size_t)Ng * 7));
            HIPCHK(p, hipMemsetAsync(p->d_tmp.p, 0, (size_t)Ng * 7 * sizeof(double), p->st));
            launch_scatter_owned_pose(io, io + (size_t)p->N * 4, p->N, p->d_l2g.p, p->d_own.p, p->d_tmp.p, p->d_tmp.p + (size_t)Ng * 4, p->st);
            if ((rc = allreduce(p, p->d_tmp.p, (size_t)Ng * 7, 0)) != PGO_OK) return rc;
            HIPCHK(p, hipMemcpyAsync(hq.data(), p->d_tmp.p, hq.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
            HIPCHK(p, hipMemcpyAsync(ht.data(), p->d_tmp.p + (size_t)Ng * 4, ht.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
            HIPCHK(p, hipStreamSynchronize(p->st));
            for (int64_t g = 0; g < Ng; ++g) if (!p->h_touched_any[g]) {   // keyframes without any residual block: the values given to solve_begin
                std::copy(p->h_init_q.begin() + g * 4, p->h_init_q.begin() + g * 4 + 4, hq.begin() + g * 4); std::copy(p->h_init_t.begin() + g * 3, p->h_init_t.begin() + g * 3 + 3, ht.begin() + g * 3);
            }
        }
        if (p->S > 0) {
            if (p->local_ids) {
                // every switch is owned by the rank holding its edge: sum (owned ? value : 0) and the owner count
                std::vector<double> own((size_t)p->S * 2, 0.0), cur((size_t)p->S);
                HIPCHK(p, hipMemcpyAsync(cur.data(), p->d_swv[p->cur].p, (size_t)p->S * sizeof(double), hipMemcpyDeviceToHost, p->st));
                HIPCHK(p, hipStreamSynchronize(p->st));
                for (int64_t i = 0; i < p->S; ++i) if (p->h_sw_used[i]) { own[i] = cur[i]; own[p->S + i] = 1.0; }
                HIPCHK(p, p->d_tmp.ensure((size_t)p->S * 2));
                HIPCHK(p, hipMemcpyAsync(p->d_tmp.p, own.data(), own.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
                if ((rc = allreduce(p, p->d_tmp.p, own.size(), 0)) != PGO_OK) return rc;
                HIPCHK(p, hipMemcpyAsync(own.data(), p->d_tmp.p, own.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
                HIPCHK(p, hipStreamSynchronize(p->st));
                for (int64_t i = 0; i < p->S; ++i) hs[i] = own[p->S + i] > 0.5 ? own[i] : (sw ? sw[i] : cur[i]);
            } else {
                HIPCHK(p, hipMemcpyAsync(hs.data(), p->d_swv[p->cur].p, (size_t)p->S * sizeof(double), hipMemcpyDeviceToHost, p->st));
            }
        }
        HIPCHK(p, hipStreamSynchronize(p->st));
        std::memcpy(quat, hq.data(), hq.size() * sizeof(double));
        std::memcpy(t, ht.data(), ht.size() * sizeof(double));
        if (sw && p->S > 0) std::memcpy(sw, hs.data(), hs.size() * sizeof(double));
    }
    // (a solve that kept it: the next three solves of this handle use it without the comparison)
    // a solve in which the coarse space lost every comparison: the following solves of this handle (incremental triggers on the same kind
    // of graph) skip it, 1, 3, 7, 15 solves at a time, before comparing again; one win resets the back-off
    if (p->coarse_mode == 2 && !p->coarse_skip_all) { p->coarse_backoff = std::min(2 * p->coarse_backoff + 1, 15); p->coarse_skip = p->coarse_backoff; }
    if (p->coarse_mode == 1) ++p->coarse_keep_streak; else if (p->coarse_mode == 2 && !p->coarse_skip_all) p->coarse_keep_streak = 0;
    mg_job_cancel(p);      // a regroup nobody needed any more: dropped (the hierarchy in place keeps its own switch record)
    if ((rc = mg_init_finish(p)) != PGO_OK) return rc;      // a fresh graph's hierarchy that this solve never needed: installed now, for the handle's next solves
    p->sum.seconds_total = now_s() - p->t_begin;
    if (out) *out = p->sum;
    p->in_solve = false;
    return PGO_OK;
}

int add_edges(pgo_problem* p, HostClass& H, int64_t n, const int32_t* c1, const int32_t* c2, const double* T, const double* w, const int32_t* sw) {
    if (n < 0 || (n > 0 && (!c1 || !c2 || !T))) { p->err = "null edge array"; return PGO_ERR_INVALID_ARG; }
    for (int64_t k = 0; k < n; ++k) if (c1[k] < 0 || c2[k] < 0 || c1[k] == c2[k] || (sw && sw[k] < 0)) { p->err = "negative index or self edge"; return PGO_ERR_INVALID_ARG; }
    mg_job_cancel(p);      // (the worker reads the edge lists)
    mg_init_drop(p);
    const size_t base = H.c1.size();
    H.c1.insert(H.c1.end(), c1, c1 + n);
    H.c2.insert(H.c2.end(), c2, c2 + n);
    if (sw) H.sw.insert(H.sw.end(), sw, sw + n);
    H.meas.resize((base + n) * 8);
    for (int64_t k = 0; k < n; ++k) meas_from_matrix(T + 16 * k, w ? w[k] : 1.0, &H.meas[(base + k) * 8]);
    p->graph_dirty = true;
    return PGO_OK;
}

}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

int32_t pgo_abi_version(void) { return PGO_ABI_VERSION; }
int64_t pgo_abi_sizeof(int32_t which) { return which == 0 ? (int64_t)sizeof(pgo_options) : which == 1 ? (int64_t)sizeof(pgo_iteration) : which == 2 ? (int64_t)sizeof(pgo_summary) : 0; }

void pgo_options_init(pgo_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->mg_dist_min_rows = 8192;
    o->mg_fine_filter = 0;
    o->mg_dist_setup = 1;
    o->max_num_iterations = 10;          // src/PoseGraphSLAM.cpp:1272
    o->linear_solver = PGO_LINEAR_PCG_MATRIX_FREE;
    o->jacobi_scaling = 1;
    o->max_num_consecutive_invalid_steps = 5;
    o->initial_trust_region_radius = 1e4;
    o->max_trust_region_radius = 1e16;
    o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3;
    o->min_lm_diagonal = 1e-6;
    o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6;
    o->gradient_tolerance = 1e-10;
    o->parameter_tolerance = 1e-8;
    o->cg_max_iterations = 50000;   // chain-like graphs need 5-15k iterations per step at large trust regions; capping them costs parity
    o->cg_check_every = 25;
    o->cg_warm_start = 1;
    o->cg_use_graph = 1;
    o->cg_early_tolerance = 1e-2;
    o->cg_early_reject_rho = -0.5;
    o->cg_mid_tolerance = 1e-4;
    o->cg_mid_reject_rho = -0.05;
    o->coarse_aggregates = 768;
    o->coarse_min_radius = 1e7;
    o->mg_min_keyframes = 5000;
    o->mg_min_keyframes_switchable = 5000;
    o->mg_omega = 0.9;
    o->mg_correction_scale = 1.0;
    o->mg_first_passes = 3;
    o->mg_passes = 0;
    o->mg_dense_max_nodes = 512;
    o->mg_switch_iterations = 400;
    o->mg_loop_discount = 3.0;
    o->mg_regroup_fraction = 0.02;
    o->mg_prolongation_damping = 0.6;
    o->mg_smoothed_levels = -1;
    o->cg_rel_tolerance = 3e-10;    // keeps the 10-iteration chi^2 of C3 within 1e-8 of the independent CPU trajectory whatever the preconditioner schedule (1e-9: 1e-7)
    o->device_id = -1;
    o->verbosity = 0;
    o->cg_single_reduction = 1;
    o->cg_pause_always = 0;
    o->mg_smoothed_fine = -1;
    o->mg_explicit_transfer = 1;
    o->cg_end_game = 1;
}

int pgo_create(pgo_problem** out, const pgo_options* opts) {
    if (!out) return PGO_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PGO_ERR_NO_DEVICE;
    pgo_problem* p = new (std::nothrow) pgo_problem();
    if (!p) return PGO_ERR_OUT_OF_MEMORY;
    if (opts) p->opt = *opts; else pgo_options_init(&p->opt);
    int dev = p->opt.device_id;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) { delete p; return PGO_ERR_NO_DEVICE; }
    p->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&p->st, hipStreamNonBlocking) != hipSuccess) { delete p; return PGO_ERR_NO_DEVICE; }
    std::memset(&p->sum, 0, sizeof(p->sum));
    if (hipHostMalloc((void**)&p->poll, 3 * sizeof(pgo_problem::Poll), hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&p->poll_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->poll_ev[1], hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(p->st); delete p; return PGO_ERR_OUT_OF_MEMORY; }
    std::memset(p->poll, 0, 3 * sizeof(pgo_problem::Poll));
    // One-time costs of the process belong here, not in the first trigger: the first device allocation and the first kernel launch of the library (its code object goes to the
    // device).  Failures here are not errors (the solve reports its own).  (A captured + instantiated graph would also take the first hipGraphInstantiate of the process off the
    // first long PCG — 9.4 ms against 0.2 ms for later ones — but a capture in one thread makes a concurrent synchronous hipMemcpy of ANOTHER thread fail with
    // hipErrorStreamCaptureImplicit on this runtime, thread-local mode or not: handles are created concurrently by callers that run one rank per thread.)
    {
        double* w = nullptr;
        if (hipMalloc((void**)&w, 4096) == hipSuccess) {
            (void)hipMemsetAsync(w, 0, 4096, p->st);
            launch_reduce(w, 0, 0, w + 8, p->st);
            (void)hipStreamSynchronize(p->st);
            (void)hipFree(w);
            // the runtime's copy paths by size class (pageable host memory, both directions) set up their staging on first use: measured, the first wake-up of a session
            // 27.3 -> 19.6 ms with these copies done here
            double* big = nullptr;
            if (hipMalloc((void**)&big, (size_t)4 << 20) == hipSuccess) {
                std::vector<char> host((size_t)4 << 20, 0);
                for (size_t bytes : {(size_t)1 << 10, (size_t)16 << 10, (size_t)32 << 10, (size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
                    (void)hipMemcpyAsync(big, host.data(), bytes, hipMemcpyHostToDevice, p->st);
                    (void)hipMemcpyAsync(host.data(), big, bytes, hipMemcpyDeviceToHost, p->st);
                    (void)hipStreamSynchronize(p->st);
                }
                (void)hipFree(big);
            }
            (void)hipGetLastError();
        }
    }
    *out = p;
    return PGO_OK;
}

int pgo_destroy(pgo_problem* p) {
    if (!p) return PGO_ERR_INVALID_ARG;
    mg_job_cancel(p);
    mg_init_drop(p);
    (void)hipSetDevice(p->device);
    if (p->comm && p->nccl.CommDestroy) p->nccl.CommDestroy(p->comm);
    (void)hipStreamSynchronize(p->st);
    for (auto& cc : p->cg_chunk) if (cc.exec) (void)hipGraphExecDestroy(cc.exec);
    if (p->poll) (void)hipHostFree(p->poll);
    for (int i = 0; i < 2; ++i) if (p->poll_ev[i]) (void)hipEventDestroy(p->poll_ev[i]);
    p->d_rc1.release(); p->d_rc2.release(); p->d_sc1.release(); p->d_sc2.release(); p->d_sidx.release(); p->d_bsr_col.release();
    p->d_rmeas.release(); p->d_smeas.release(); p->d_rwin.release(); p->d_swin.release(); p->d_prior.release();
    p->d_inc_rowptr.release(); p->d_inc.release(); p->d_bsr_rowptr.release(); p->d_node_free.release();
    p->d_Jr.release(); p->d_Js.release(); p->d_Jp.release(); p->d_Hd_g.release(); p->d_Hoff.release(); p->d_c.release(); p->d_hss.release(); p->d_gs.release();
    p->d_scale_p.release(); p->d_scale_s.release(); p->d_diag_p.release(); p->d_diag_s.release(); p->d_a_inv.release();
    p->d_val.release(); p->d_Lf.release(); p->d_Dtot_b.release(); p->d_cgvec.release(); p->d_part.release(); p->d_cgpart.release();
    p->d_flags.release(); p->d_scal.release(); p->d_pose[0].release(); p->d_pose[1].release(); p->d_swv[0].release(); p->d_swv[1].release();
    p->d_delta_s.release(); p->d_io.release(); p->d_tmp.release(); p->d_vio.release(); p->d_vio_idx.release(); p->d_vio_meas.release();
    p->d_mg_f64.release(); p->d_mg_i32.release(); p->d_mg_i64.release();
    p->d_ccen.release(); p->d_cd.release(); p->d_cAc.release(); p->d_crc.release(); p->d_cblk_ptr.release(); p->d_ccontrib.release(); p->d_cblk_ab.release(); p->d_cagg_free.release(); p->d_cinfo.release(); p->d_cscr.release(); p->d_cAcf.release();
    p->d_l2g.release(); p->d_fp_send.release(); p->d_fp_shloc.release(); p->d_fp_sumptr.release(); p->d_fp_sumsrc.release(); p->d_own.release(); p->d_xsend[0].release(); p->d_xsend[1].release(); p->d_xrecv.release(); p->d_xscal.release();
    p->d_einc.release(); p->d_einc_slot.release(); p->d_node_rng.release(); p->d_tile_inc0.release(); p->d_einc_other.release();
    p->d_tile_node0.release(); p->d_tile_sw0.release(); p->d_node_prior.release(); p->d_rec.release(); p->d_lam.release();
    (void)hipStreamDestroy(p->st);
    delete p;
    return PGO_OK;
}

int pgo_set_options(pgo_problem* p, const pgo_options* o) {
    if (!p || !o) return PGO_ERR_INVALID_ARG;
    const int dev = p->opt.device_id;
    mg_job_cancel(p);
    mg_init_drop(p);
    if (o->linear_solver != p->opt.linear_solver) p->graph_dirty = true;
    // the preconditioner hierarchies are part of the device graph build
    if (o->mg_min_keyframes != p->opt.mg_min_keyframes || o->mg_min_keyframes_switchable != p->opt.mg_min_keyframes_switchable || o->mg_first_passes != p->opt.mg_first_passes || o->mg_passes != p->opt.mg_passes ||
        o->mg_dense_max_nodes != p->opt.mg_dense_max_nodes || o->coarse_aggregates != p->opt.coarse_aggregates || o->mg_smoothed_levels != p->opt.mg_smoothed_levels || o->mg_loop_discount != p->opt.mg_loop_discount ||
        o->mg_explicit_transfer != p->opt.mg_explicit_transfer || o->mg_smoothed_fine != p->opt.mg_smoothed_fine || o->mg_dist_min_rows != p->opt.mg_dist_min_rows || o->mg_dist_setup != p->opt.mg_dist_setup ||
        o->mg_fine_filter != p->opt.mg_fine_filter) p->graph_dirty = true;
    p->opt = *o;
    p->opt.device_id = dev;   // the device binding is fixed at create
    return PGO_OK;
}

int pgo_reserve(pgo_problem* p, int64_t n_nodes, int64_t n_edges) {
    if (!p || n_nodes < 0 || n_edges < 0) return PGO_ERR_INVALID_ARG;
    if ((size_t)n_edges <= p->rel.c1.capacity() && (size_t)n_edges <= p->rel.c2.capacity() && (size_t)n_edges * 8 <= p->rel.meas.capacity()) return PGO_OK;      // nothing moves
    // the edge arrays are about to be reallocated: the hierarchy workers (a fresh graph's, a regroup's) read them — same rule as every other mutating entry point
    if (p->in_solve) { p->err = "pgo_reserve inside a solve (between pgo_solve_begin and pgo_solve_end)"; return PGO_ERR_STATE; }
    mg_job_cancel(p);
    mg_init_drop(p);
    p->rel.c1.reserve(n_edges); p->rel.c2.reserve(n_edges); p->rel.meas.reserve((size_t)n_edges * 8);
    return PGO_OK;
}

int pgo_add_relpose_edges(pgo_problem* p, int64_t n, const int32_t* c1, const int32_t* c2, const double* T, const double* w) {
    if (!p) return PGO_ERR_INVALID_ARG;
    if (n > 0 && !w) { p->err = "weight array required for relative-pose edges"; return PGO_ERR_INVALID_ARG; }
    return add_edges(p, p->rel, n, c1, c2, T, w, nullptr);
}
int pgo_add_switchable_edges(pgo_problem* p, int64_t n, const int32_t* c1, const int32_t* c2, const double* T, const double* w, const int32_t* sw) {
    if (!p) return PGO_ERR_INVALID_ARG;
    if (n > 0 && !sw) { p->err = "switch index array required"; return PGO_ERR_INVALID_ARG; }
    return add_edges(p, p->swe, n, c1, c2, T, w, sw);
}
int pgo_set_node_regularizers(pgo_problem* p, int64_t n, const int32_t* node, const double* target, const double* weight) {
    if (!p || n < 0 || (n > 0 && (!node || !target || !weight))) return PGO_ERR_INVALID_ARG;
    std::vector<PriorDev> v((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        if (node[k] < 0) { p->err = "negative regulariser node"; return PGO_ERR_INVALID_ARG; }
        const double* T = target + 16 * k;
        PriorDev& P = v[k];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) P.Rf[r * 3 + c] = T[c * 4 + r];
        P.tf[0] = T[12]; P.tf[1] = T[13]; P.tf[2] = T[14];
        eigen_matrix_to_quat(P.Rf, P.qf);
        P.w = weight[k]; P.node = node[k]; P.pad_ = 0;
    }
    mg_job_cancel(p);
    mg_init_drop(p);
    p->priors.swap(v);
    p->priors_dirty = true;
    return PGO_OK;
}
int pgo_set_nodes_constant(pgo_problem* p, int64_t n, const int32_t* node) {
    if (!p || n < 0 || (n > 0 && !node)) return PGO_ERR_INVALID_ARG;
    for (int64_t k = 0; k < n; ++k) if (node[k] < 0) return PGO_ERR_INVALID_ARG;
    mg_job_cancel(p);      // (the worker reads h_node_free / constant_nodes)
    mg_init_drop(p);
    p->constant_nodes.insert(p->constant_nodes.end(), node, node + n);
    p->graph_dirty = true;
    return PGO_OK;
}
// ---- graph construction from the resident VIO poses (K0) ----
int pgo_set_vio_poses(pgo_problem* p, int64_t first, int64_t n, const double* w_M) {
    if (!p || first < 0 || n < 0 || (n > 0 && !w_M)) return PGO_ERR_INVALID_ARG;
    if (first > p->n_vio) { p->err = "VIO poses must be appended contiguously"; return PGO_ERR_INVALID_ARG; }
    if (n == 0) return PGO_OK;
    HIPCHK(p, hipSetDevice(p->device));
    const int64_t need = first + n;
    if ((size_t)need * 16 > p->d_vio.cap) {       // grow geometrically, keep the old poses
        ScopedBuf<double> bigger;
        HIPCHK(p, bigger.ensure((size_t)std::max<int64_t>(need + need / 2, 1024) * 16));
        if (p->n_vio > 0) HIPCHK(p, hipMemcpyAsync(bigger.p, p->d_vio.p, (size_t)p->n_vio * 16 * sizeof(double), hipMemcpyDeviceToDevice, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        std::swap(p->d_vio.p, bigger.p); std::swap(p->d_vio.cap, bigger.cap);
    }
    HIPCHK(p, hipMemcpyAsync(p->d_vio.p + (size_t)first * 16, w_M, (size_t)n * 16 * sizeof(double), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    p->n_vio = std::max(p->n_vio, need);
    return PGO_OK;
}
int pgo_num_vio_poses(const pgo_problem* p, int64_t* n) { if (!p || !n) return PGO_ERR_INVALID_ARG; *n = p->n_vio; return PGO_OK; }

int pgo_add_odometry_edges_from_vio(pgo_problem* p, const int32_t* set_id, int64_t u_begin, int64_t u_end, int32_t f_max, int32_t use_yaw_weight, int64_t* n_added) {
    if (!p || u_begin < 0 || u_end < u_begin || f_max < 1) return PGO_ERR_INVALID_ARG;
    if (u_end > p->n_vio) { p->err = "odometry edges requested beyond the resident VIO poses"; return PGO_ERR_INVALID_ARG; }
    if (p->in_solve) { p->err = "graph construction inside a solve"; return PGO_ERR_STATE; }
    mg_job_cancel(p);      // (a regroup's worker left behind by a failed solve reads the edge lists)
    mg_init_drop(p);
    std::vector<int32_t> c1, c2;
    c1.reserve((size_t)(u_end - u_begin) * f_max); c2.reserve(c1.capacity());
    for (int64_t u = u_begin; u < u_end; ++u)
        for (int f = 1; f <= f_max; ++f) {
            if (u - f < 0) continue;                                           // (:1588-1591)
            if (set_id && (set_id[u] < 0 || set_id[u - f] < 0)) continue;      // dead zone (:1583-1586)
            c1.push_back((int32_t)u); c2.push_back((int32_t)(u - f));
        }
    const int64_t n = (int64_t)c1.size();
    if (n_added) *n_added = n;
    if (n == 0) return PGO_OK;
    HIPCHK(p, hipSetDevice(p->device));
    DBuf<int32_t>& d_c = p->d_vio_idx;            // c1 then c2 (kept across calls)
    DBuf<double>& d_meas = p->d_vio_meas;
    HIPCHK(p, d_c.ensure((size_t)2 * n)); HIPCHK(p, d_meas.ensure((size_t)8 * n));
    HIPCHK(p, hipMemcpyAsync(d_c.p, c1.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(d_c.p + n, c2.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    launch_vio_odometry(n, d_c.p, d_c.p + n, p->d_vio.p, use_yaw_weight, d_meas.p, p->st);
    HostClass& H = p->rel;
    const size_t base = H.c1.size();
    H.meas.resize((base + n) * 8);
    // (the second wake-up of a process spends ~8 ms inside this copy call — runtime-internal, once; a pinned staging buffer of our own does not change it: measured)
    HIPCHK(p, hipMemcpyAsync(&H.meas[base * 8], d_meas.p, (size_t)8 * n * sizeof(double), hipMemcpyDeviceToHost, p->st));
    const hipError_t e = hipStreamSynchronize(p->st);
    if (e != hipSuccess) { H.meas.resize(base * 8); p->err = hipGetErrorString(e); return PGO_ERR_HIP; }
    H.c1.insert(H.c1.end(), c1.begin(), c1.end());
    H.c2.insert(H.c2.end(), c2.begin(), c2.end());
    p->graph_dirty = true;
    return PGO_OK;
}

int pgo_initial_guess_from_vio(pgo_problem* p, int64_t n_left, const double* left, const int32_t* left_of_node, int64_t u_begin, int64_t u_end, double* quat, double* t) {
    if (!p || n_left < 0 || u_begin < 0 || u_end < u_begin) return PGO_ERR_INVALID_ARG;
    const int64_t cnt = u_end - u_begin;
    if (cnt == 0) return PGO_OK;
    if (!left_of_node || !quat || !t || (n_left > 0 && !left)) return PGO_ERR_INVALID_ARG;
    if (u_end > p->n_vio) { p->err = "initial guesses requested beyond the resident VIO poses"; return PGO_ERR_INVALID_ARG; }
    bool any = false;
    for (int64_t i = 0; i < cnt; ++i) {
        if (left_of_node[i] >= n_left) { p->err = "left-matrix selector out of range"; return PGO_ERR_INVALID_ARG; }
        any |= left_of_node[i] >= 0;
    }
    if (!any) return PGO_OK;
    HIPCHK(p, hipSetDevice(p->device));
    ScopedBuf<double> d_left, d_q, d_t;
    ScopedBuf<int32_t> d_sel;
    HIPCHK(p, d_left.ensure((size_t)n_left * 16)); HIPCHK(p, d_q.ensure((size_t)cnt * 4)); HIPCHK(p, d_t.ensure((size_t)cnt * 3)); HIPCHK(p, d_sel.ensure((size_t)cnt));
    HIPCHK(p, hipMemcpyAsync(d_left.p, left, (size_t)n_left * 16 * sizeof(double), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(d_sel.p, left_of_node, (size_t)cnt * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    // untouched keyframes keep the caller's values: seed the staging buffers with them
    HIPCHK(p, hipMemcpyAsync(d_q.p, quat + u_begin * 4, (size_t)cnt * 4 * sizeof(double), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(d_t.p, t + u_begin * 3, (size_t)cnt * 3 * sizeof(double), hipMemcpyHostToDevice, p->st));
    launch_vio_initial_guess(u_begin, cnt, d_left.p, d_sel.p, p->d_vio.p, d_q.p, d_t.p, p->st);
    HIPCHK(p, hipMemcpyAsync(quat + u_begin * 4, d_q.p, (size_t)cnt * 4 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipMemcpyAsync(t + u_begin * 3, d_t.p, (size_t)cnt * 3 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

int pgo_get_relpose_edge_records(const pgo_problem* p, int64_t first, int64_t n, int32_t* c1, int32_t* c2, double* record8) {
    if (!p || first < 0 || n < 0 || first + n > p->rel.size()) return PGO_ERR_INVALID_ARG;
    if (c1) std::copy(p->rel.c1.begin() + first, p->rel.c1.begin() + first + n, c1);
    if (c2) std::copy(p->rel.c2.begin() + first, p->rel.c2.begin() + first + n, c2);
    if (record8) std::copy(p->rel.meas.begin() + first * 8, p->rel.meas.begin() + (first + n) * 8, record8);
    return PGO_OK;
}
int pgo_num_relpose_edges(const pgo_problem* p, int64_t* n) { if (!p || !n) return PGO_ERR_INVALID_ARG; *n = p->rel.size(); return PGO_OK; }
int pgo_num_switchable_edges(const pgo_problem* p, int64_t* n) { if (!p || !n) return PGO_ERR_INVALID_ARG; *n = p->swe.size(); return PGO_OK; }
int pgo_num_regularizers(const pgo_problem* p, int64_t* n) { if (!p || !n) return PGO_ERR_INVALID_ARG; *n = (int64_t)p->priors.size(); return PGO_OK; }

int pgo_solve_begin(pgo_problem* p, const double* q, const double* t, const double* sw, int64_t N, int64_t S) {
    if (!p) return PGO_ERR_INVALID_ARG;
    return solve_begin(p, q, t, sw, N, S);
}
// a failed step or write-back: no regroup worker outlives it (it reads host arrays the caller may change next), and a stream capture a failing launch left open is ended
static void after_failure(pgo_problem* p) {
    mg_job_cancel(p);
    mg_init_drop(p);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (p->st && hipStreamIsCapturing(p->st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(p->st, &g); if (g) (void)hipGraphDestroy(g); p->cg_graph_failed = true; }
    (void)hipGetLastError();
}
int pgo_lm_step(pgo_problem* p, int32_t ignore_termination, int32_t* done) {
    if (!p) return PGO_ERR_INVALID_ARG;
    int d = 0;
    const int rc = lm_step(p, ignore_termination, &d);
    if (rc != PGO_OK) after_failure(p);
    if (done) *done = d;
    return rc;
}
int pgo_solve_end(pgo_problem* p, double* q, double* t, double* sw, pgo_summary* s) {
    if (!p) return PGO_ERR_INVALID_ARG;
    const int rc = solve_end(p, q, t, sw, s);
    if (rc != PGO_OK) { after_failure(p); p->in_solve = false; }
    return rc;
}
int pgo_solve(pgo_problem* p, double* q, double* t, double* sw, int64_t N, int64_t S, pgo_summary* s) {
    if (!p) return PGO_ERR_INVALID_ARG;
    int rc = solve_begin(p, q, t, sw, N, S);
    if (rc != PGO_OK) return rc;
    int done = p->terminated ? 1 : 0;
    while (!done) { rc = lm_step(p, 0, &done); if (rc != PGO_OK) { after_failure(p); p->in_solve = false; return rc; } }
    rc = solve_end(p, q, t, sw, s);
    if (rc != PGO_OK) { after_failure(p); p->in_solve = false; }
    return rc;
}

int pgo_evaluate(pgo_problem* p, const double* q, const double* t, const double* sw, int64_t N, int64_t S, double* cost, double* residuals, double* gradient) {
    if (!p) return PGO_ERR_INVALID_ARG;
    const pgo_summary keep = p->sum;
    int rc = solve_begin(p, q, t, sw, N, S);   // upload + K1 + K2 + norms at the given point
    if (rc != PGO_OK) return rc;
    p->in_solve = false;
    if (cost) *cost = p->x_cost;
    const int64_t Er = p->G.rel.E, Es = p->G.sw.E, Eg = p->G.n_prior;
    if (residuals) {
        const int64_t total = 6 * Er + 7 * Es + 6 * Eg;
        HIPCHK(p, p->d_tmp.ensure(std::max<int64_t>(total, 1)));
        launch_unpack_k1(p->G, 0, 0, Er, p->d_tmp.p, nullptr, nullptr, nullptr, p->st);
        launch_unpack_k1(p->G, 1, 0, Es, p->d_tmp.p + 6 * Er, nullptr, nullptr, nullptr, p->st);
        launch_unpack_k1(p->G, 2, 0, Eg, p->d_tmp.p + 6 * Er + 7 * Es, nullptr, nullptr, nullptr, p->st);
        HIPCHK(p, hipMemcpyAsync(residuals, p->d_tmp.p, total * sizeof(double), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
    }
    if (gradient) {
        std::vector<double> gs((size_t)std::max<int64_t>(Es, 1));
        // constant keyframes have no gradient entry; zero them on the device copy before it is spread over all keyframes
        std::vector<double> gl((size_t)p->N * 6);
        HIPCHK(p, hipMemcpyAsync(gl.data(), p->L.g, gl.size() * sizeof(double), hipMemcpyDeviceToHost, p->st));
        if (Es) HIPCHK(p, hipMemcpyAsync(gs.data(), p->L.gs, Es * sizeof(double), hipMemcpyDeviceToHost, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        for (int64_t n = 0; n < p->N; ++n) if (!p->h_node_free[n]) for (int c = 0; c < 6; ++c) gl[6 * n + c] = 0.0;
        HIPCHK(p, p->d_io.ensure(gl.size()));
        HIPCHK(p, hipMemcpyAsync(p->d_io.p, gl.data(), gl.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
        if ((rc = nodes_to_global(p, p->d_io.p, 6, gradient)) != PGO_OK) return rc;
        for (int64_t i = 0; i < S; ++i) gradient[6 * N + i] = 0.0;
        for (int64_t e = 0; e < Es; ++e) gradient[6 * N + p->swe.sw[e]] = gs[e];   // multi-GPU: each rank reports the switches of its own edges
    }
    p->sum = keep;
    return PGO_OK;
}

int pgo_get_jacobian_blocks(pgo_problem* p, int32_t kind, int64_t first, int64_t count, double* J1, double* J2, double* dr_ds) {
    if (!p || kind < 0 || kind > 2 || first < 0 || count < 0) return PGO_ERR_INVALID_ARG;
    if (p->graph_dirty) { p->err = "no linearisation available"; return PGO_ERR_STATE; }
    const int64_t E = kind == 0 ? p->G.rel.E : kind == 1 ? p->G.sw.E : p->G.n_prior;
    if (first + count > E) return PGO_ERR_INVALID_ARG;
    if (count == 0) return PGO_OK;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    HIPCHK(p, p->d_tmp.ensure((size_t)count * 79));
    double* d1 = p->d_tmp.p; double* d2 = d1 + count * 36; double* ds = d2 + count * 36;
    launch_unpack_k1(p->G, kind, first, count, nullptr, d1, kind == 2 ? nullptr : d2, kind == 1 ? ds : nullptr, p->st);
    if (J1) HIPCHK(p, hipMemcpyAsync(J1, d1, count * 36 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    if (J2 && kind != 2) HIPCHK(p, hipMemcpyAsync(J2, d2, count * 36 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    if (dr_ds && kind == 1) HIPCHK(p, hipMemcpyAsync(dr_ds, ds, count * 7 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

int pgo_get_normal_blocks(pgo_problem* p, double* diag, double* grad, double* offdiag, double* sw_c, double* sw_hss, double* sw_gs) {
    if (!p) return PGO_ERR_INVALID_ARG;
    if (p->graph_dirty) { p->err = "no linearisation available"; return PGO_ERR_STATE; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const int64_t Er = p->G.rel.E, Es = p->G.sw.E;
    if (diag && (rc = nodes_to_global(p, p->L.Hd, 36, diag)) != PGO_OK) return rc;
    if (grad && (rc = nodes_to_global(p, p->L.g, 6, grad)) != PGO_OK) return rc;
    if (offdiag && p->built_mf) {   // the matrix-free solver never forms J1^T J2: compute it for the parity hook only
        HIPCHK(p, p->d_Hoff.ensure((size_t)(p->G.rel.Epad + p->G.sw.Epad) * 36));
        p->L.Hoff = p->d_Hoff.p;
        launch_k2(p->G, p->L, true, p->st);
    }
    if (offdiag) {
        if (Er) HIPCHK(p, hipMemcpyAsync(offdiag, p->L.Hoff, Er * 36 * sizeof(double), hipMemcpyDeviceToHost, p->st));
        if (Es) HIPCHK(p, hipMemcpyAsync(offdiag + Er * 36, p->L.Hoff + (size_t)p->G.rel.Epad * 36, Es * 36 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    }
    if (sw_c && Es) HIPCHK(p, hipMemcpyAsync(sw_c, p->L.c, Es * 12 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    if (sw_hss && Es) HIPCHK(p, hipMemcpyAsync(sw_hss, p->L.hss, Es * sizeof(double), hipMemcpyDeviceToHost, p->st));
    if (sw_gs && Es) HIPCHK(p, hipMemcpyAsync(sw_gs, p->L.gs, Es * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

int pgo_manifold_plus(pgo_problem* p, int64_t n, const double* quat, const double* t, const double* delta, double* quat_out, double* t_out) {
    if (!p || n < 0 || (n > 0 && (!quat || !delta || !quat_out))) return PGO_ERR_INVALID_ARG;
    if (n == 0) return PGO_OK;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const bool with_t = t != nullptr && t_out != nullptr;
    ScopedBuf<double> d;
    HIPCHK(p, d.ensure((size_t)n * 20));
    double* dq = d.p; double* dd = dq + 4 * n; double* dqo = dd + 6 * n; double* dt = dqo + 4 * n; double* dto = dt + 3 * n;
    HIPCHK(p, hipMemcpyAsync(dq, quat, (size_t)n * 4 * sizeof(double), hipMemcpyHostToDevice, p->st));
    HIPCHK(p, hipMemcpyAsync(dd, delta, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, p->st));
    if (with_t) HIPCHK(p, hipMemcpyAsync(dt, t, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, p->st));
    launch_manifold_plus(n, dq, with_t ? dt : nullptr, dd, dqo, with_t ? dto : nullptr, p->st);
    HIPCHK(p, hipMemcpyAsync(quat_out, dqo, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    if (with_t) HIPCHK(p, hipMemcpyAsync(t_out, dto, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

int pgo_apply_normal_operator(pgo_problem* p, const double* x, double* y) {
    if (!p || !x || !y) return PGO_ERR_INVALID_ARG;
    if (!p->in_solve) { p->err = "pgo_apply_normal_operator needs an open solve (pgo_solve_begin)"; return PGO_ERR_STATE; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const pgo_options& o = p->opt;
    if (!p->reuse_diagonal) launch_lm_diag(p->G, p->L, p->Sc, o.min_lm_diagonal, o.max_lm_diagonal, p->st);
    bool ok = true;
    if ((rc = build_system(p, &ok)) != PGO_OK) return rc;
    // x and y are arrays over ALL keyframes (multi-GPU: this rank applies its part to its keyframes, shared rows are summed)
    HIPCHK(p, p->d_io.ensure((size_t)p->N * 12));
    double* xin = p->d_io.p; double* yout = p->d_io.p + (size_t)p->N * 6;
    if ((rc = nodes_from_global(p, x, 6, xin)) != PGO_OK) return rc;
    if (p->built_mf) launch_mf_apply(p->G, p->F, p->Sc, p->C, xin, yout, p->st);
    else launch_apply_operator(p->G, p->C, xin, yout, p->st);
    if ((rc = exchange_rows(p, yout, 6, nullptr, 0)) != PGO_OK) return rc;
    return nodes_to_global(p, yout, 6, y);
}

// ---- multi-GPU ----
static int load_rccl(Rccl& r, std::string& err) {
    if (r.h) return PGO_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) { err = std::string("dlopen(librccl): ") + dlerror(); return PGO_ERR_COMM; }
    r.GetUniqueId = (int (*)(void*))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, Rccl::Uid, int))dlsym(r.h, "ncclCommInitRank");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.h, "ncclAllReduce");
    r.CommDestroy = (int (*)(void*))dlsym(r.h, "ncclCommDestroy");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(r.h, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(r.h, "ncclRecv");
    r.GroupStart = (int (*)())dlsym(r.h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.h, "ncclGroupEnd");
    r.GetErrorString = (const char* (*)(int))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { err = "librccl: missing symbols"; return PGO_ERR_COMM; }
    return PGO_OK;
}
static Rccl g_rccl_for_id;

int pgo_comm_get_unique_id(uint8_t id[PGO_COMM_ID_BYTES]) {
    if (!id) return PGO_ERR_INVALID_ARG;
    std::string err;
    if (load_rccl(g_rccl_for_id, err) != PGO_OK) return PGO_ERR_COMM;
    Rccl::Uid u;
    if (g_rccl_for_id.GetUniqueId(&u) != 0) return PGO_ERR_COMM;
    std::memcpy(id, u.b, PGO_COMM_ID_BYTES);
    return PGO_OK;
}
int pgo_comm_init(pgo_problem* p, int32_t rank, int32_t world, const uint8_t id[PGO_COMM_ID_BYTES]) {
    if (!p || !id || world < 1 || rank < 0 || rank >= world) return PGO_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    if ((rc = load_rccl(p->nccl, p->err)) != PGO_OK) return rc;
    Rccl::Uid u;
    std::memcpy(u.b, id, PGO_COMM_ID_BYTES);
    void* comm = nullptr;
    const int nrc = p->nccl.CommInitRank(&comm, world, u, rank);
    if (nrc != 0) { p->err = std::string("ncclCommInitRank: ") + (p->nccl.GetErrorString ? p->nccl.GetErrorString(nrc) : "error"); return PGO_ERR_COMM; }
    mg_job_cancel(p); mg_init_drop(p);
    p->comm = comm; p->rank = rank; p->world = world;
    p->graph_dirty = true;   // keyframe participation is the union over ranks
    return PGO_OK;
}
int pgo_comm_init_custom(pgo_problem* p, int32_t rank, int32_t world, pgo_allreduce_fn fn, void* ctx) {
    if (!p || !fn || world < 1 || rank < 0 || rank >= world) return PGO_ERR_INVALID_ARG;
    mg_job_cancel(p); mg_init_drop(p);
    p->custom_allreduce = fn; p->custom_ctx = ctx; p->rank = rank; p->world = world;
    p->graph_dirty = true;
    return PGO_OK;
}
int pgo_comm_set_exchange(pgo_problem* p, pgo_exchange_fn fn) {
    if (!p || !p->custom_allreduce) return PGO_ERR_INVALID_ARG;      // (belongs to a communicator set up by pgo_comm_init_custom)
    p->custom_exchange = fn;
    return PGO_OK;
}
int pgo_local_group_create(int32_t world, void** group) {
    if (!group || world < 1 || world > pgo_local::MAX_RANKS) return PGO_ERR_INVALID_ARG;
    pgo_local::Group* G = new (std::nothrow) pgo_local::Group();
    if (!G) return PGO_ERR_OUT_OF_MEMORY;
    G->world = world;
    *group = G;
    return PGO_OK;
}
int pgo_local_group_abort(void* group) {
    if (!group) return PGO_ERR_INVALID_ARG;
    static_cast<pgo_local::Group*>(group)->abort();
    return PGO_OK;
}
int pgo_local_group_destroy(void* group) {
    if (!group) return PGO_ERR_INVALID_ARG;
    delete static_cast<pgo_local::Group*>(group);
    return PGO_OK;
}
int pgo_comm_init_local(pgo_problem* p, int32_t rank, int32_t world, void* group) {
    pgo_local::Group* G = static_cast<pgo_local::Group*>(group);
    if (!p || !G || world != G->world || rank < 0 || rank >= world) return PGO_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    pgo_local::Group::Slot& me = G->slot[rank];
    if (me.joined) { p->err = "in-process communicator: the rank is taken"; return PGO_ERR_INVALID_ARG; }
    for (int k = 0; k < 2; ++k) {
        HIPCHK(p, hipEventCreateWithFlags(&me.ready[k], hipEventDisableTiming));
        HIPCHK(p, hipEventCreateWithFlags(&me.done[k], hipEventDisableTiming));
    }
    me.device = p->device; me.joined = true;
    mg_job_cancel(p); mg_init_drop(p);
    p->local_group = G; p->lc_count = 0; p->rank = rank; p->world = world;
    p->graph_dirty = true;
    // ranks on other GPUs of this process: their buffers are read over xGMI (peer access); every rank has joined once all have passed this barrier
    if (!G->barrier()) { p->err = "in-process communicator: not all ranks joined"; return PGO_ERR_COMM; }
    for (int q = 0; q < world; ++q) if (q != rank && G->slot[q].device != p->device) { const hipError_t e = hipDeviceEnablePeerAccess(G->slot[q].device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); p->err = "in-process communicator: no peer access between the ranks' GPUs"; G->abort(); return PGO_ERR_COMM; } (void)hipGetLastError(); }
    return PGO_OK;
}
int pgo_get_sharding_stats(pgo_problem* p, pgo_sharding_stats* out) {
    if (!p || !out) return PGO_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    out->world = p->world; out->rank = p->rank;
    if (!p->local_ids || p->graph_dirty) return PGO_OK;
    out->keyframes_local = p->N;
    for (double w : p->h_own) if (w != 0.0) ++out->keyframes_owned;
    out->keyframes_shared = p->n_sh_mine; out->shared_global = p->n_sh_global;
    out->pcg_iterations = p->st_pcg_iterations; out->exchanges = p->st_exchanges; out->allreduces = p->st_allreduces;
    out->bytes_sent_neighbour = p->st_bytes_neighbour; out->bytes_allreduce = p->st_bytes_allreduce;
    const double fine = (double)p->fine_plan.x.n_send() * 48.0 + 16.0;
    out->bytes_sent_per_bj_iteration = fine; out->exchanges_per_bj_iteration = 1;
    out->bytes_round5_per_bj_iteration = (6.0 * (double)p->n_sh_global + 2.0) * 8.0;
    if (p->mg_built && !p->mg_init_pending && p->M.n_levels >= 1) {
        const int nl = p->M.n_levels;
        out->mg_levels = nl; out->mg_levels_distributed = p->mg_levels_distributed;
        out->mg_rows_total = p->mg_rows_total; out->mg_rows_own = p->mg_rows_own; out->mg_blocks_total = p->mg_blocks_total; out->mg_blocks_own = p->mg_blocks_own;
        double bytes = fine; int nx = 1;
        auto count = [&](int point, int lv) { int plan; double* v1; double* v2; const double* dinv; if (mg_exchange_at(p, point, lv, &plan, &v1, &v2, &dinv) && p->lvl_plan[(size_t)plan].plan) { bytes += (double)p->lvl_plan[(size_t)plan].plan->n_send() * (v2 && !dinv ? 96.0 : 48.0); ++nx; } };
        for (int l = 1; l <= nl; ++l) count(0, l);
        for (int l = nl - 1; l >= 1; --l) count(1, l);
        count(2, 1);
        out->bytes_sent_per_mg_iteration = bytes; out->exchanges_per_mg_iteration = nx;
        out->bytes_round5_per_mg_iteration = (6.0 * (double)p->n_sh_global + 2.0 + 6.0 * (double)p->M.n1) * 8.0;
        // the set-up: blocks formed per LM system (level matrices; Ps, W and R^T of smoothed transitions), by all and by this rank; what its exchanges send
        const int fw = p->mg_first_whole;
        for (int l = 0; l + 1 < nl; ++l) {
            const MgLevelDev& A = p->mg_levels[l];
            const int64_t all = A.nnzb + (A.smoothed ? (int64_t)A.n_ps + 2 * (int64_t)A.n_w : 0);
            const int64_t own = l < fw ? (A.su_blk1 - A.su_blk0) + (A.smoothed ? (int64_t)(A.su_ps1 - A.su_ps0) + 2 * (int64_t)(A.su_w1 - A.su_w0) : 0) : all;
            out->mg_setup_blocks_total += all; out->mg_setup_blocks_own += own;
        }
        out->bytes_allreduce_replicated_setup = (double)p->mg_levels[0].nnzb * 288.0;
        out->mg_setup_levels_own_rows = fw; out->mg_setup_exchanges = 1;
        if (fw > 0) {
            double sb = 0.0; int nx = 0;
            auto add = [&](const pgo_mg::ExchangePlan& X, double bytes_per_row) { if (!plan_is_empty(X)) { sb += (double)X.n_send() * bytes_per_row; ++nx; } };
            for (const pgo_mg::BlockPlan& B : p->mg_setup.val) add(B.x, 288.0);
            for (int l = 0; l < fw; ++l) {
                nx += 9; sb += 24.0;      // the level's power method: eight halo exchanges of the iterate + the 3-double all-reduce
                if ((size_t)l < p->lvl_plan.size() && p->lvl_plan[(size_t)l].plan) sb += 8.0 * (double)p->lvl_plan[(size_t)l].plan->n_send() * 48.0;
                if (!p->mg_levels[l].smoothed) continue;
                if ((size_t)l < p->lvl_plan.size() && p->lvl_plan[(size_t)l].plan) add(*p->lvl_plan[(size_t)l].plan, 288.0);
                add(p->mg_setup.ps[(size_t)l], 288.0); add(p->mg_setup.rv[(size_t)l], 144.0);
            }
            out->bytes_sent_per_mg_setup = sb; out->mg_setup_exchanges = nx;
        }
    }
    return PGO_OK;
}
// Diagnostic (tests): sums of squares of what this rank's cycle kernels read of level `level` (1-based) — the same whichever way the set-up ran (pgo_options.mg_dist_setup)
int pgo_mg_level_norms(pgo_problem* p, int32_t level, double* out8) {
    if (!p || !out8) return PGO_ERR_INVALID_ARG;
    for (int k = 0; k < 8; ++k) out8[k] = 0.0;
    if (!p->mg_built || p->mg_init_pending || level < 1 || level > p->M.n_levels || (size_t)(level - 1) >= p->mg_own.size()) { p->err = "pgo_mg_level_norms: no such level (is a hierarchy installed?)"; return PGO_ERR_INVALID_ARG; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const MgLevelDev& A = p->mg_levels[level - 1];
    const pgo_problem::OwnRange& R = p->mg_own[(size_t)level - 1];
    HIPCHK(p, hipStreamSynchronize(p->st));
    auto sq64 = [&](const double* dev, int64_t first, int64_t count, double* out) -> int {
        if (!dev || count <= 0) return PGO_OK;
        std::vector<double> h((size_t)count);
        HIPCHK(p, hipMemcpy(h.data(), dev + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
        long double s = 0.0L; for (double v : h) s += (long double)v * v;
        *out = (double)s; return PGO_OK;
    };
    auto sq32 = [&](const float* dev, int64_t first, int64_t count, double* out) -> int {
        if (!dev || count <= 0) return PGO_OK;
        std::vector<float> h((size_t)count);
        HIPCHK(p, hipMemcpy(h.data(), dev + first, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
        long double s = 0.0L; for (float v : h) s += (long double)v * v;
        *out = (double)s; return PGO_OK;
    };
    const bool sparse = level < p->M.n_levels;
    if ((rc = sq64(A.val, R.blk0 * 36, (R.blk1 - R.blk0) * 36, out8 + 0)) != PGO_OK) return rc;
    if (sparse) {
        if ((rc = sq32(A.valf, R.blk0 * 36, (R.blk1 - R.blk0) * 36, out8 + 1)) != PGO_OK) return rc;
        if ((rc = sq64(A.Dinv, R.row0 * 36, (R.row1 - R.row0) * 36, out8 + 2)) != PGO_OK) return rc;
        if (A.smoothed && A.rt_valf) {
            if ((rc = sq32(A.rt_valf, R.w0 * 36, (R.w1 - R.w0) * 36, out8 + 3)) != PGO_OK) return rc;
            if ((rc = sq32(A.r_valf, R.rT0 * 36, (R.rT1 - R.rT0) * 36, out8 + 4)) != PGO_OK) return rc;
        }
    } else if ((rc = sq64(p->K.Ac, 0, (int64_t)p->K.nc * p->K.nc, out8 + 5)) != PGO_OK) return rc;      // the dense level: its inverse
    return PGO_OK;
}
int pgo_comm_destroy(pgo_problem* p) {
    if (!p) return PGO_ERR_INVALID_ARG;
    p->custom_allreduce = nullptr; p->custom_ctx = nullptr; p->custom_exchange = nullptr;
    if (p->local_group) {      // every rank's stream has drained before any event or staging buffer goes (a peer's kernel may still be reading them)
        pgo_local::Group* G = p->local_group;
        (void)hipStreamSynchronize(p->st);
        (void)G->barrier();
        pgo_local::Group::Slot& me = G->slot[p->rank];
        for (int k = 0; k < 2; ++k) {
            if (me.ready[k]) (void)hipEventDestroy(me.ready[k]);
            if (me.done[k]) (void)hipEventDestroy(me.done[k]);
            if (me.stage[k]) (void)hipFree(me.stage[k]);
            me.ready[k] = me.done[k] = nullptr; me.stage[k] = nullptr; me.stage_cap[k] = 0; me.ptr[k] = nullptr; me.send_off[k] = nullptr;
        }
        me.joined = false;
        p->local_group = nullptr;
    }
    if (p->comm && p->nccl.CommDestroy) { (void)hipStreamSynchronize(p->st); p->nccl.CommDestroy(p->comm); }
    mg_job_cancel(p); mg_init_drop(p);
    p->comm = nullptr; p->rank = 0; p->world = 1;
    p->graph_dirty = true;
    return PGO_OK;
}

// ---- edge sharding policies (host only) ----
int pgo_partition_edges(int32_t policy, int32_t world, int64_t n_nodes, const double* t_xyz, int64_t n_rel, const int32_t* rel_c1, const int32_t* rel_c2,
                        int64_t n_sw, const int32_t* sw_c1, const int32_t* sw_c2, int32_t* node_part, int32_t* rel_rank, int32_t* sw_rank) {
    if (world < 1 || n_nodes < 0 || n_rel < 0 || n_sw < 0 || (n_rel > 0 && (!rel_c1 || !rel_c2 || !rel_rank)) || (n_sw > 0 && (!sw_c1 || !sw_c2 || !sw_rank))) return PGO_ERR_INVALID_ARG;
    if (policy == PGO_PARTITION_CONTIGUOUS) {
        // rank r holds the edges [n r / world, n (r+1) / world) of each class
        for (int cls = 0; cls < 2; ++cls) {
            const int64_t n = cls ? n_sw : n_rel; int32_t* out = cls ? sw_rank : rel_rank;
            for (int r = 0; r < world; ++r) for (int64_t e = (n * r) / world; e < (n * (r + 1)) / world; ++e) out[e] = r;
        }
        return PGO_OK;
    }
    if (policy != PGO_PARTITION_CHAIN && policy != PGO_PARTITION_SPATIAL) return PGO_ERR_INVALID_ARG;
    if (policy == PGO_PARTITION_SPATIAL && n_nodes > 0 && !t_xyz) return PGO_ERR_INVALID_ARG;
    for (int64_t e = 0; e < n_rel; ++e) if (rel_c1[e] < 0 || rel_c1[e] >= n_nodes || rel_c2[e] < 0 || rel_c2[e] >= n_nodes) return PGO_ERR_INVALID_ARG;
    for (int64_t e = 0; e < n_sw; ++e) if (sw_c1[e] < 0 || sw_c1[e] >= n_nodes || sw_c2[e] < 0 || sw_c2[e] >= n_nodes) return PGO_ERR_INVALID_ARG;
    // parts are balanced by the edges they will receive (an edge goes with its later endpoint); keyframes without edges still spread evenly
    std::vector<double> load((size_t)n_nodes, 0.0);
    for (int64_t e = 0; e < n_rel; ++e) load[std::max(rel_c1[e], rel_c2[e])] += 1.0;
    for (int64_t e = 0; e < n_sw; ++e) load[std::max(sw_c1[e], sw_c2[e])] += 1.0;
    for (double& v : load) v += 1e-3;
    std::vector<int32_t> part((size_t)n_nodes, 0);
    if (policy == PGO_PARTITION_CHAIN) {
        std::vector<double> c((size_t)n_nodes);
        double acc = 0.0;
        for (int64_t i = 0; i < n_nodes; ++i) { acc += load[i]; c[i] = acc; }
        for (int64_t i = 0; i < n_nodes; ++i) part[i] = (int32_t)std::min((c[i] - load[i]) * (double)world / acc, (double)(world - 1));
    } else {
        // recursive coordinate bisection: cells [lo, hi) get the keyframes idx[b, e); split along the axis of largest extent at the load quantile
        std::vector<int32_t> idx((size_t)n_nodes), tmp;
        for (int64_t i = 0; i < n_nodes; ++i) idx[i] = (int32_t)i;
        struct Job { int64_t b, e; int lo, hi; };
        std::vector<Job> stack{{0, n_nodes, 0, world}};
        while (!stack.empty()) {
            const Job j = stack.back(); stack.pop_back();
            if (j.hi - j.lo <= 1 || j.e - j.b <= 0) { for (int64_t k = j.b; k < j.e; ++k) part[idx[k]] = j.lo; continue; }
            const int mid = (j.lo + j.hi) / 2;
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
            for (int64_t k = j.b; k < j.e; ++k) for (int a = 0; a < 3; ++a) { const double v = t_xyz[(size_t)idx[k] * 3 + a]; mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v); }
            int axis = 0;
            for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
            std::stable_sort(idx.begin() + j.b, idx.begin() + j.e, [&](int32_t x, int32_t y) { return t_xyz[(size_t)x * 3 + axis] < t_xyz[(size_t)y * 3 + axis]; });
            const int64_t len = j.e - j.b;
            std::vector<double> c((size_t)len);
            double acc = 0.0;
            for (int64_t k = 0; k < len; ++k) { acc += load[idx[j.b + k]]; c[k] = acc; }
            const double target = acc * (double)(mid - j.lo) / (double)(j.hi - j.lo);
            int64_t k = std::lower_bound(c.begin(), c.end(), target) - c.begin();
            if (len > 1) k = std::min(std::max<int64_t>(k, 1), len - 1); else k = len;
            stack.push_back({j.b + k, j.e, mid, j.hi});
            stack.push_back({j.b, j.b + k, j.lo, mid});
        }
    }
    for (int64_t e = 0; e < n_rel; ++e) rel_rank[e] = part[std::max(rel_c1[e], rel_c2[e])];
    for (int64_t e = 0; e < n_sw; ++e) sw_rank[e] = part[std::max(sw_c1[e], sw_c2[e])];
    if (node_part) std::copy(part.begin(), part.end(), node_part);
    return PGO_OK;
}

// ---- measurement helpers ----
int pgo_time_kernel(pgo_problem* p, int32_t which, int32_t launches, double* avg_ms, double* algorithmic_bytes) {
    if (!p || launches <= 0 || !avg_ms) return PGO_ERR_INVALID_ARG;
    if (!p->in_solve) { p->err = "pgo_time_kernel needs an open solve (pgo_solve_begin)"; return PGO_ERR_STATE; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    EventPair ev;
    HIPCHK(p, ev.create());
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    int np = 0;
    const int nxt = p->cur ^ 1;
    const GraphDev& G = p->G;
    double bytes = 0, best_ms = -1.0;
    if (which == 6 || which == 7 || which == 8) {   // one multigrid-preconditioned PCG iteration (6) / its level kernels alone (7) / the kernels of the multigrid's set-up (8), on the current LM system
        if (!p->mg_built || !p->built_mf || (p->local_ids && which == 6)) { p->err = "pgo_time_kernel: this graph has no multigrid hierarchy (mg_min_keyframes) / several ranks: only the level kernels (7) can be timed"; return PGO_ERR_STATE; }
        const pgo_options& o = p->opt;
        if (!p->reuse_diagonal) launch_lm_diag(p->G, p->L, p->Sc, o.min_lm_diagonal, o.max_lm_diagonal, p->st);
        bool ok = true;
        if ((rc = build_system(p, &ok)) != PGO_OK) return rc;
        if (!p->mg_active && (rc = build_mg(p)) != PGO_OK) return rc;
        if (!p->mg_active) { p->err = "pgo_time_kernel: the multigrid operators of this system are not positive definite"; return PGO_ERR_NUMERIC; }
        const int g = launch_cg_init_vectors(p->G, p->C, 0, p->st);
        if (p->local_ids) { if ((rc = mg_apply_ranks(p, false)) != PGO_OK) return rc; launch_cg_set_tolerance(p->C, 0.0, p->st); }      // (one full distributed cycle: every level vector holds finite numbers)
        else {
            launch_mg_apply(p->G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz, mg_scale(p), false, p->st, false, mg_cs(p), mg_fine_view(p));
            launch_cg_init_scalars(p->C, g, g, 0.0, p->st);
        }
    }
    // in-process ranks share the GPU(s) of one process: the timed launches of the ranks take turns (every rank's figure is what its GPU would need on its own)
    const int turns = (p->local_group && (which == 7 || which == 8)) ? p->world : 1;
    for (int turn = 0; turn < turns; ++turn) {
    if (turns > 1) { HIPCHK(p, hipStreamSynchronize(p->st)); if (!p->local_group->barrier()) { p->err = "in-process communicator: a rank left during pgo_time_kernel"; return PGO_ERR_COMM; } if (turn != p->rank) continue; }
    if (which == 5 && single_reduction(p)) {      // (its head needs the u.w partials of a matvec on the CURRENT u: launched back to back it sees stale ones, breaks down and returns early)
        p->err = "pgo_time_kernel(5): the single-reduction update cannot be timed without its matvec; time the iteration (2) and the matvec (4) and subtract"; return PGO_ERR_STATE;
    }
    if (which == 2 || which == 4 || which == 5) {   // a live PCG state to iterate on (tolerance 0: never converges during the timed launches)
        const pgo_options& o = p->opt;
        if (!p->reuse_diagonal) launch_lm_diag(p->G, p->L, p->Sc, o.min_lm_diagonal, o.max_lm_diagonal, p->st);
        bool ok = true;
        if ((rc = build_system(p, &ok)) != PGO_OK) return rc;
        p->mg_active = false; p->coarse_active = false; p->C.extra_rz = 0;   // the timed iteration is the plain block-Jacobi one: no partial-sum slots of a multigrid / two-level solve
        launch_cg_init(p->G, p->C, 0, 0.0, p->st);
    }
    const double N = (double)G.N, E = (double)(G.rel.E + G.sw.E), Es = (double)G.sw.E;
    // one untimed launch first (instruction cache, TLB).  In-process ranks: three timed batches, the fastest counts — the first batch after a solve_begin that regrouped the
    // hierarchy was measured at 3-5x the steady figure on every rank (C5 on 8 ranks: 0.47-0.82 ms, then 0.146-0.168 ms call after call): eight handles' old images going back to
    // the system stall the GPU's address translation for tens of milliseconds
    const int batches = turns > 1 ? 3 : 1;
    for (int rep = 0; rep < 1 + batches; ++rep) {
        const int n = rep == 0 ? 1 : launches;
        if (rep >= 1) HIPCHK(p, hipEventRecord(e0, p->st));
        for (int i = 0; i < n; ++i) {
            switch (which) {
                case 0: launch_k1(G, p->d_pose[p->cur].p, p->d_swv[p->cur].p, true, part(p, 0), &np, p->st); bytes = k1_algorithmic_bytes(G, true); break;
                case 1: launch_k2(G, p->L, !p->built_mf, p->st, p->built_mf ? &p->F : nullptr); bytes = (624.0 * G.rel.E + 688.0 * Es) + 288.0 * E + 336.0 * N + 112.0 * Es; break;
                case 2: case 4: case 5: {   // one PCG iteration (2), its matvec alone (4), its vector update alone (5)
                          const int kk = rep == 0 ? 0 : i + 1;
                          const bool sr = single_reduction(p);      // the form the solver runs on this handle
                          if (sr) {
                              if (which != 5) launch_mf_apply_dot_live(G, p->F, p->Sc, p->C, p->st);
                              if (which != 4) launch_cg_update_sr(G, p->C, kk, kk == 0 ? 1 : 0, mf_grid_size(p->F), p->st);
                          } else {
                              if (which != 5) { if (p->built_mf) launch_mf_spmv(G, p->F, p->Sc, p->C, kk, 0.0, p->st); else launch_cg_spmv(G, p->C, kk, 0.0, p->st); }
                              if (which != 4) launch_cg_update(G, p->C, kk, p->built_mf ? mf_grid_size(p->F) : cg_grid_size(G), p->st);
                          }
                          // Bytes this design moves per iteration, each array once.  Matrix-free matvec: per LANE (a relative-pose edge with both
                          // keyframes in one tile is one lane, every other edge side its own) the compact record (8 double2 planes; 11 for switchable
                          // sides) + 12 B of index data (+ a_inv for switchable sides); per keyframe z and p_prev read, p and q written (4 x 48),
                          // damping 48, side ranges / regulariser index / free flag 13.  Update: r, q, p, x read, r, x, z written (7 x 48), the fp32
                          // block-Jacobi factor 96.  Block-CSR matvec: SURVEY.md 8d's assembled form.
                          const double lanes_rel = (double)(p->mf_pair_lanes + p->mf_rel_side_lanes), lanes_sw = (double)p->mf_sw_lanes;
                          // Single-reduction form: the matvec reads u and writes w (2 x 48 per keyframe instead of 4 x 48); the update reads u, w, p, s, x, r and writes p, s, x, r, u (11 x 48).
                          const double mv = p->built_mf ? lanes_rel * (128.0 + 12.0) + lanes_sw * (128.0 + 12.0 + 8.0) + N * ((sr ? 2.0 : 4.0) * 48.0 + 48.0 + 13.0)
                                                        : 288.0 * (N + 2.0 * E) + 4.0 * (N + 2.0 * E) + N * 4.0 * 48.0;
                          const double up = N * ((sr ? 11.0 : 7.0) * 48.0 + 96.0);
                          bytes = which == 2 ? mv + up : which == 4 ? mv : up;
                          break; }
                case 3: launch_k1(G, p->d_pose[nxt].p, p->d_swv[nxt].p, false, part(p, 5), &np, p->st); bytes = k1_algorithmic_bytes(G, false); break;
                case 6: case 7: {
                          const int kk = rep == 0 ? 0 : i + 1;
                          const bool fused = p->M.blk_tab != nullptr;
                          const bool sr = single_reduction(p);
                          if (which == 6 && sr) {
                              launch_mf_apply_dot_live(G, p->F, p->Sc, p->C, p->st);
                              if (fused) launch_cg_update_mg_sr(G, p->C, p->M, p->mg_levels, p->K, kk, kk == 0 ? 1 : 0, mf_grid_size(p->F), p->st);
                              else launch_cg_update_sr(G, p->C, kk, kk == 0 ? 1 : 0, mf_grid_size(p->F), p->st);
                          } else if (which == 6) {
                              launch_mf_spmv(G, p->F, p->Sc, p->C, kk, 0.0, p->st);
                              if (fused) launch_cg_update_mg(G, p->C, p->M, p->mg_levels, p->K, kk, mf_grid_size(p->F), p->st);
                              else launch_cg_update(G, p->C, kk, mf_grid_size(p->F), p->st);
                          }
                          if (p->local_ids) {      // several ranks: this rank's share of the cycle's kernels, no exchanges (what its GPU computes per cycle)
                              launch_mg_apply(G, p->C, p->M, p->mg_levels, p->K, p->C.r, p->C.z, p->C.part_rz, mg_scale(p), false, p->st, false, mg_cs(p), nullptr);
                              bytes = (double)p->mg_blocks_own * 148.0 + (double)p->mg_rows_own * (288.0 + 24.0 + 8.0 * 48.0 + 16.0) + (double)p->K.nc * (double)p->K.nc * 4.0 + (double)p->K.nc * 16.0;
                              break;
                          }
                          launch_mg_apply(G, p->C, p->M, p->mg_levels, p->K, sr ? p->C.r : ((kk & 1) ? p->C.r : p->C.r2), p->C.z, p->C.part_rz + (size_t)((kk & 1) ^ 1) * RZ_STRIDE, mg_scale(p), true, p->st, which == 6 && fused, mg_cs(p), mg_fine_view(p));
                          // Bytes of this design, each array once per kernel that streams it.  Fine level as in case 2 (+ the restriction's per-keyframe offsets and slot table,
                          // the prolongation's read-modify-write of z, offsets and aggregate index); every sparse coarse level: its fp32 blocks and column indices twice
                          // (down- and up-sweep), Dinv, positions/offsets and its four vectors; the dense level: the fp32 inverse once.
                          const double lanes_rel = (double)(p->mf_pair_lanes + p->mf_rel_side_lanes), lanes_sw = (double)p->mf_sw_lanes;
                          const double fine = lanes_rel * (128.0 + 12.0) + lanes_sw * (128.0 + 12.0 + 8.0) + N * ((sr ? 2.0 : 4.0) * 48.0 + 48.0 + 13.0) + N * ((sr ? 11.0 : 7.0) * 48.0 + 96.0);
                          double cyc = N * (24.0 + 16.0 / 8.0 * 8.0) /* d0 + slot table (restriction) */ + N * (2.0 * 48.0 + 24.0 + 4.0 + 4.0) /* z read + write, d0, agg0, member list (prolongation) */;
                          for (int l = 0; l + 1 < p->M.n_levels; ++l) {
                              const MgLevelDev& A = p->mg_levels[l];
                              if (A.smoothed && A.rt_valf)      // explicit transfer operator: the level's own blocks once (smoothing step), R and R^T once each, Dinv once, r / x / y / xf and the level above's r, x
                                  cyc += (double)A.nnzb * (144.0 + 4.0) + 2.0 * (double)A.n_w * (144.0 + 4.0) + (double)A.n * (288.0 + 24.0 + 8.0 * 48.0 + 16.0) + (double)A.n_next * (288.0 + 2.0 * 48.0 + 8.0);
                              else
                              cyc += (A.smoothed ? 4.0 : 2.0) * (double)A.nnzb * (144.0 + 4.0) + (double)A.n * ((A.smoothed ? 4.0 : 2.0) * 288.0 /* Dinv: smoothing steps */ + 24.0 + (A.smoothed ? 18.0 : 10.0) * 48.0 + 16.0);
                          }
                          cyc += (double)p->K.nc * (double)p->K.nc * 4.0 + (double)p->K.nc * 16.0;
                          bytes = which == 6 ? fine + cyc : cyc;
                          break; }
                case 8: {     // this rank's kernels of one multigrid set-up (operators of an LM system incl. the dense inverse), without the exchanges between them
                          const double omega = p->opt.mg_omega > 0.0 && p->opt.mg_omega <= 1.0 ? p->opt.mg_omega : 0.9;
                          const bool hoff_valid = !p->built_mf || p->hoff_epoch == p->lin_epoch;
                          int32_t* fail = p->d_cinfo.p;
                          if (p->local_ids && p->mg_first_whole > 0) { if ((rc = build_mg_ranks(p, omega, fail, hoff_valid, true)) != PGO_OK) return rc; }
                          else if (p->mg_fine) { launch_mg_assemble_fine(p->G, p->L, p->Sc, p->C, p->mg_fineF, p->mg_fineT, p->mg_levels[0], omega, fail, p->st, mg_cs(p), hoff_valid, p->d_pose[p->cur].p); launch_mg_assemble_rest(p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p)); }
                          else launch_mg_assemble(p->G, p->L, p->Sc, p->C, p->M, p->mg_levels, p->K, omega, fail, p->st, mg_cs(p), hoff_valid);
                          launch_coarse_invert(p->K, p->d_cscr.p, fail, p->st);
                          bytes = 0.0;
                          break; }
                default: return PGO_ERR_INVALID_ARG;
            }
        }
        if (rep >= 1) HIPCHK(p, hipEventRecord(e1, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        if (rep >= 1) { float msb = 0; HIPCHK(p, hipEventElapsedTime(&msb, e0, e1)); if (best_ms < 0.0 || (double)msb < best_ms) best_ms = (double)msb; }
    }
    }
    if (turns > 1 && !p->local_group->barrier()) { p->err = "in-process communicator: a rank left during pgo_time_kernel"; return PGO_ERR_COMM; }
    if (which == 8) { p->mg_active = false; if ((rc = build_mg(p)) != PGO_OK) return rc; }      // (several ranks: the timed kernels ran without their exchanges — the operators are formed again, properly)
    *avg_ms = best_ms / launches;
    if (algorithmic_bytes) *algorithmic_bytes = bytes;
    return PGO_OK;
}
int pgo_time_linearize_kernel(pgo_problem* p, int32_t launches, double* avg_ms, double* bytes) { return pgo_time_kernel(p, 0, launches, avg_ms, bytes); }

int pgo_time_vio_odometry_kernel(pgo_problem* p, int32_t f_max, int32_t launches, double* avg_ms, double* algorithmic_bytes) {
    if (!p || launches <= 0 || !avg_ms || f_max < 1) return PGO_ERR_INVALID_ARG;
    if (p->n_vio < 2) { p->err = "no resident VIO poses"; return PGO_ERR_STATE; }
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    std::vector<int32_t> c;
    for (int64_t u = 0; u < p->n_vio; ++u) for (int f = 1; f <= f_max; ++f) if (u - f >= 0) c.push_back((int32_t)u);
    const int64_t n = (int64_t)c.size();
    for (int64_t u = 0; u < p->n_vio; ++u) for (int f = 1; f <= f_max; ++f) if (u - f >= 0) c.push_back((int32_t)(u - f));
    ScopedBuf<int32_t> d_c; ScopedBuf<double> d_meas;
    HIPCHK(p, d_c.ensure((size_t)2 * n)); HIPCHK(p, d_meas.ensure((size_t)8 * n));
    HIPCHK(p, hipMemcpyAsync(d_c.p, c.data(), (size_t)2 * n * sizeof(int32_t), hipMemcpyHostToDevice, p->st));
    EventPair ev;
    HIPCHK(p, ev.create());
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    launch_vio_odometry(n, d_c.p, d_c.p + n, p->d_vio.p, 1, d_meas.p, p->st);
    HIPCHK(p, hipEventRecord(e0, p->st));
    for (int i = 0; i < launches; ++i) launch_vio_odometry(n, d_c.p, d_c.p + n, p->d_vio.p, 1, d_meas.p, p->st);
    HIPCHK(p, hipEventRecord(e1, p->st));
    HIPCHK(p, hipStreamSynchronize(p->st));
    float ms = 0;
    HIPCHK(p, hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / launches;
    if (algorithmic_bytes) *algorithmic_bytes = 128.0 * (double)p->n_vio + (8.0 + 64.0) * (double)n;   // each pose once + 2 indices + one record per edge
    return PGO_OK;
}

int pgo_dense_spd_inverse(pgo_problem* p, int32_t n, const double* a, double* a_inv, int32_t launches, double* avg_ms) {
    if (!p || n <= 0 || !a || !a_inv || launches < 1) return PGO_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    const int nc = (n + 63) / 64 * 64;
    std::vector<double> h((size_t)nc * nc, 0.0);
    for (int i = 0; i < nc; ++i) {
        if (i < n) std::memcpy(&h[(size_t)i * nc], a + (size_t)i * n, (size_t)n * sizeof(double));
        else h[(size_t)i * nc + i] = 1.0;
    }
    ScopedBuf<double> d_a, d_scr; ScopedBuf<int32_t> d_fail;
    HIPCHK(p, d_a.ensure((size_t)nc * nc)); HIPCHK(p, d_scr.ensure((size_t)nc * 64 + 4096)); HIPCHK(p, d_fail.ensure(1));
    CoarseDev K{}; K.nc = nc; K.Ac = d_a.p;
    EventPair ev;
    HIPCHK(p, ev.create());
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    float total = 0;
    for (int l = 0; l < launches; ++l) {
        HIPCHK(p, hipMemcpyAsync(d_a.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, p->st));
        HIPCHK(p, hipMemsetAsync(d_fail.p, 0, sizeof(int32_t), p->st));
        HIPCHK(p, hipEventRecord(e0, p->st));
        launch_coarse_invert(K, d_scr.p, d_fail.p, p->st);
        HIPCHK(p, hipEventRecord(e1, p->st));
        HIPCHK(p, hipStreamSynchronize(p->st));
        float ms = 0;
        HIPCHK(p, hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    int32_t fail = 1;
    HIPCHK(p, hipMemcpy(&fail, d_fail.p, sizeof(fail), hipMemcpyDeviceToHost));
    HIPCHK(p, hipMemcpy(h.data(), d_a.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) std::memcpy(a_inv + (size_t)i * n, &h[(size_t)i * nc], (size_t)n * sizeof(double));
    if (avg_ms) *avg_ms = (double)total / launches;
    if (fail) { p->err = "matrix is not numerically positive definite"; return PGO_ERR_NUMERIC; }
    return PGO_OK;
}

int pgo_device_synchronize(pgo_problem* p) {
    if (!p) return PGO_ERR_INVALID_ARG;
    int rc;
    if ((rc = set_device(p)) != PGO_OK) return rc;
    if ((rc = mg_init_finish(p)) != PGO_OK) return rc;      // "everything this handle has in flight": the hierarchy worker of a fresh graph build too
    HIPCHK(p, hipStreamSynchronize(p->st));
    return PGO_OK;
}

const char* pgo_strerror(int code) {
    switch (code) {
        case PGO_OK: return "ok";
        case PGO_ERR_INVALID_ARG: return "invalid argument";
        case PGO_ERR_NO_DEVICE: return "no usable HIP device (libpgo has no CPU fallback)";
        case PGO_ERR_HIP: return "HIP runtime error";
        case PGO_ERR_OUT_OF_MEMORY: return "out of device memory";
        case PGO_ERR_STATE: return "call order violated";
        case PGO_ERR_COMM: return "RCCL error";
        case PGO_ERR_NUMERIC: return "non-finite value";
        default: return "unknown error";
    }
}
const char* pgo_last_error(const pgo_problem* p) { return p ? p->err.c_str() : ""; }
#ifndef PGO_SOURCE_SHA256
#define PGO_SOURCE_SHA256 "unknown (not built by _build.py)"
#endif
const char* pgo_build_info(void) { return "libpgo sources sha256:" PGO_SOURCE_SHA256; }

}  // extern "C"
