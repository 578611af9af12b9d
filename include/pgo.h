/*
 * pgo.h — C-ABI of libpgo: the MI355X-native 6-DoF pose-graph Levenberg-Marquardt solver.
 *
 * This is the drop-in boundary for the ONE hot path of mpkuse/solve_keyframe_pose_graph:
 * everything `PoseGraphSLAM::reinit_ceres_problem_onnewloopedge_optimize6DOF()` asks of Ceres
 * (reference src/PoseGraphSLAM.cpp:1251-1950) plus the three cost functors of
 * reference src/CeresResidues.h:19-222.  Each entry point below cites the reference
 * interface it replaces.  Plain C: opaque handle, caller-owned arrays, fp64 + int32/int64.
 * No torch types, no C++ types, never throws, never calls exit().
 *
 * Conventions (identical to the reference's storage, src/PoseGraphSLAM.h:153-175,
 * src/utils/PoseManipUtils.cpp:61-98):
 *   - node orientation: unit quaternion, 4 doubles per node in order x,y,z,w (Eigen coeffs order)
 *   - node translation: 3 doubles per node
 *   - one switch double per switchable edge, addressed by the caller-chosen switch index
 *   - measurements are Eigen `Matrix4d` values, i.e. 16 doubles COLUMN-major, exactly what the
 *     reference hands to `X::Create(const Matrix4d&, double)` (CeresResidues.h:72,129,204)
 *   - tangent ordering inside the library (gradient output of pgo_evaluate):
 *       [dtheta(3), dt(3)] per node, node-major, followed by one entry per switch variable.
 *     dtheta is the Ceres `EigenQuaternionParameterization` increment: q <- [sin|d| d/|d|, cos|d|] (x) q.
 *   - cost is the Ceres convention 0.5 * sum ||r||^2;  chi^2 = 2 * cost.
 *
 * Threading: a handle is single-caller (the reference drives Ceres from one thread, th_slam,
 * src/keyframe_pose_graph_slam_node.cpp:475-477).  pgo_solve writes the caller's arrays exactly
 * once, at the very end (the reference relies on this: src/PoseGraphSLAM.cpp:1894-1903).
 */
#ifndef PGO_H_
#define PGO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGO_VERSION_MAJOR 0
#define PGO_VERSION_MINOR 1

typedef struct pgo_problem pgo_problem; /* opaque; replaces the persistent `ceres::Problem reint_problem`
                                           (reference src/PoseGraphSLAM.h:197) */

/* ---- error codes (0 = OK, negative = error).  Reference: asserts / exit(n) in the driver
 *      (src/PoseGraphSLAM.cpp:201,207,274,1680,1711); Ceres reports through Summary. ---- */
enum {
    PGO_OK = 0,
    PGO_ERR_INVALID_ARG = -1,    /* null pointer, negative count, index out of range */
    PGO_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime error at create */
    PGO_ERR_HIP = -3,            /* a HIP call failed; pgo_last_error() has the text */
    PGO_ERR_OUT_OF_MEMORY = -4,
    PGO_ERR_STATE = -5,          /* call order violated (e.g. pgo_lm_step before pgo_solve_begin) */
    PGO_ERR_COMM = -6,           /* RCCL failure */
    PGO_ERR_NUMERIC = -7         /* non-finite cost/step at the initial point */
};

/* ---- termination types; same meaning as ceres::TerminationType as consumed at
 *      src/PoseGraphSLAM.cpp:1905,1912 ---- */
enum {
    PGO_CONVERGENCE = 0,
    PGO_NO_CONVERGENCE = 1,
    PGO_FAILURE = 2
};

/* ---- linear solver for the LM normal equations (replaces SPARSE_NORMAL_CHOLESKY,
 *      src/PoseGraphSLAM.cpp:1270) ---- */
enum {
    PGO_LINEAR_PCG_BLOCK_JACOBI = 0, /* device PCG on the Schur-reduced pose system, 6x6 block-Jacobi, assembled block-CSR matrix */
    PGO_LINEAR_PCG_MATRIX_FREE = 1   /* same PCG, the matvec evaluated matrix-free from one compact record per edge side (default) */
};

/* Options.  Defaults (pgo_options_init) are the Ceres defaults the reference runs with, plus
 * max_num_iterations = 10 (src/PoseGraphSLAM.cpp:1268-1272).  See SURVEY.md Appendix B. */
typedef struct pgo_options {
    int32_t max_num_iterations;          /* 10   (PoseGraphSLAM.cpp:1272) */
    int32_t linear_solver;               /* PGO_LINEAR_PCG_MATRIX_FREE */
    int32_t jacobi_scaling;              /* 1    (Ceres default) */
    int32_t max_num_consecutive_invalid_steps; /* 5 */
    double initial_trust_region_radius;  /* 1e4  */
    double max_trust_region_radius;      /* 1e16 */
    double min_trust_region_radius;      /* 1e-32 */
    double min_relative_decrease;        /* 1e-3 */
    double min_lm_diagonal;              /* 1e-6 */
    double max_lm_diagonal;              /* 1e32 */
    double function_tolerance;           /* 1e-6 */
    double gradient_tolerance;           /* 1e-10 */
    double parameter_tolerance;          /* 1e-8 */
    /* PCG controls (no Ceres counterpart: Ceres factorises exactly) */
    int32_t cg_max_iterations;           /* 50000: a safety net, not a budget — a capped PCG is an inexact LM step and leaves the exact-solve path */
    int32_t cg_check_every;              /* 25: host polls the device convergence flag every this many iterations (rounded down to even; 12 while the two-level preconditioner is on) */
    double cg_rel_tolerance;             /* 3e-10: stop when ||r||_{M^-1} <= tol * ||b||_{D^-1} (D = block-Jacobi).  The block-Jacobi factors are stored rounded
                                          *       to fp32 (a preconditioner may be anything symmetric positive definite; the arithmetic applying them is fp64).
                                          *       Measured on C3 against the independent CPU trajectory (tests/golden): chi^2 after 10 LM iterations within 9e-8 / 1.3e-7 (with / without the
                                          *       early-rejection pauses: the two runs switch preconditioner at different points) at 1e-9, 3e-8 / 7e-8 at 5e-10, 8e-9 / 1.1e-8 at 3e-10,
                                          *       1.1e-8 / 2.4e-8 at 2e-10; after 20 iterations 1.5e-9 at 1e-9, 1.8e-10 at 3e-10; 3e-10 costs 5 % more PCG iterations than 1e-9 */
    int32_t cg_warm_start;               /* 1: after a rejected step start the PCG from the previous step (same H, larger damping) */
    int32_t cg_use_graph;                /* 1: replay each `cg_check_every`-iteration chunk of the PCG loop as one hipGraph (single GPU) */
    /* Early rejection: a rejected LM step only shrinks the trust region, so the PCG pauses at up to two intermediate tolerances, the
     * candidate is evaluated there, and a step whose relative_decrease is already below the stage's threshold (and which would trip
     * neither convergence test) is rejected without paying for the remaining decades; otherwise the SAME PCG resumes. */
    /* NOT a Ceres rule: with an exact solve Ceres always evaluates the full step.  A step whose relative_decrease at the pause is below the
     * stage's threshold while the converged step would have cleared min_relative_decrease would be rejected here and accepted by Ceres; the
     * thresholds are far enough from min_relative_decrease that no such step has been seen (tests/test_gpu_fuzz.py compares the sequences
     * with the stages on and off over hundreds of graphs), but a caller that wants Ceres' exact decision rule sets both tolerances to 0.
     * The `relative_decrease`, `cost_change` and `step_norm` logged for an early-rejected step are the values at the pause.
     * Graphs below 20 000 keyframes (the reference's sessions: steps accepted almost throughout, short PCGs) arm the pauses only once a step of the solve has been
     * rejected: a pause costs one candidate evaluation, ~0.1 ms (measured: a 400-keyframe trigger 20.8 -> 17.1 ms without them). */
    double cg_early_tolerance;           /* 1e-2: first stage (0 disables the stage) */
    double cg_early_reject_rho;          /* -0.5: threshold of the first stage — far below min_relative_decrease because the step is still crude */
    double cg_mid_tolerance;             /* 1e-4: second stage (0 disables the stage) */
    double cg_mid_reject_rho;            /* -0.05: threshold of the second stage — the step is within ~1e-4 of the final one there */
    /* Two-level preconditioner for large trust regions: block-Jacobi plus the rigid-body modes of `coarse_aggregates` aggregates of
     * consecutive keyframes, the coarse operator formed and inverted densely (blocked Gauss-Jordan kernels) once per LM iteration.  It is used when the
     * aggregates hold <= 64 keyframes each (small and mid-size graphs: 3-20x fewer PCG iterations at every radius) and otherwise for LM
     * iterations whose trust-region radius is >= coarse_min_radius (aggregates up to 1024 keyframes), where the slow modes are the long
     * wavelengths the coarse space removes (10-60x fewer iterations); with large aggregates at small radii it does not pay.  Once per solve (and once more when a coarse space dropped
     * at a small radius becomes eligible at coarse_min_radius) plain block-Jacobi gets the same time budget on the same system; the
     * loser is not used for the rest of the solve; a handle that kept it skips the comparison in its next three solves, one that dropped
     * it skips the coarse space for its next 1, 3, 7, 15 solves.  Aggregate count: at most `coarse_aggregates`, at least 8 keyframes each but not fewer
     * than min(coarse_aggregates / 2, 256) on small graphs; a graph of no more keyframes than that gets one aggregate per keyframe, which makes the
     * coarse inverse the inverse of the reduced system itself (a direct solve refined by the PCG).  Inside the PCG the preconditioner costs one
     * kernel on top of block-Jacobi's two: the restriction rides in the vector update, the prolongation in the next matvec, and the dense solve
     * (its inverse streamed in fp32) also yields the coarse part of r.z.  The solution of each step is the same to the PCG tolerance.  Single GPU only. */
    int32_t coarse_aggregates;           /* 768 (coarse dimension <= 4608: 170 MB dense inverse + its fp32 copy; graphs up to 4096 keyframes use <= 512); 0 disables */
    /* Aggregation multigrid for large graphs (single GPU): z = D^-1 r + P V(P^T r) — block-Jacobi on the keyframes plus one V(1,1)
     * cycle over a hierarchy of rigid aggregates (dtheta_i = dtheta_a, dt_i = dt_a - 2 [d_i]x dtheta_a).  Level 1 groups up to
     * 2^mg_first_passes keyframes along RELATIVE-POSE (odometry) edges only — a switchable loop closure may be an outlier the solver is about
     * to switch off, and aggregates held together by such an edge cost 2-3x the iterations once it is off (measured on C3); the levels above
     * match whole groups along their summed couplings, odometry and loop closures alike (heavy-edge matching, 2^mg_passes nodes per
     * aggregate).  Block-Jacobi smoothing with damping mg_omega on every level, every coarse correction scaled by mg_correction_scale,
     * the coarsest level (<= mg_dense_max_nodes) inverted densely.  The operators are the Galerkin products of the current LM system,
     * rebuilt for every LM system that uses them.  Replaces the two-level preconditioner above on graphs of at least mg_min_keyframes
     * keyframes (0 disables).  No comparison runs and no per-handle history: what runs depends on the solve alone (mg_switch_iterations).
     * Like every preconditioner it changes the iteration count of the PCG, not the solution of a step beyond cg_rel_tolerance. */
    int32_t mg_min_keyframes;            /* 5000 (graphs without switchable edges; see mg_min_keyframes_switchable).  0 turns the multigrid off for every graph.  ROUND 6: with the smoothed keyframe
                                          *      transition (mg_smoothed_fine, by default wherever its levels stay small) the multigrid beats the two-level method from ~4 000 keyframes on, on plain and
                                          *      switchable loops alike — 10 LM iterations, two-level / multigrid time: 2 000 keyframes 0.60-0.81, 3 000 0.85-0.97, 4 000 0.89-1.08, 6 000 1.45-2.2,
                                          *      8 000 1.13-2.1, 12 000 1.4-2.4; BASELINE config 2 (10 000 keyframes, plain loops) 0.163 -> 0.111 s (profiles/r06_mg_crossover.txt) — rounds 2-5 had 24 000 here.
                                          *      Measured in round 2 (20 LM steps) on four graph types at 8k / 15k / 25k keyframes the two-level method wins 3:1 / 3:1 / 1:3
                                          *      against the multigrid (which costs ~3x per iteration); C3-structured 20k keyframes
                                          *      0.66 -> 0.50 s, 100k (C3) 0.70 -> 0.54 s, 200k (C4) 8.8 -> 4.6 s, 1M (C5, 10 steps) 15.8 -> 10.9 s; a 10k-keyframe
                                          *      chain with few loops (C2) is better off with the two-level method (0.32 vs 0.42 s) */
    double coarse_min_radius;            /* 1e7 (measured on the 100k-keyframe benchmark graph, 196 keyframes per aggregate: at radius 1e5..1e6 the coarse space saves 1.3x
                                          *      iterations at 2.7x the cost per iteration — and its comparison run doubled the cost of that LM step) */
    double mg_omega;                     /* 0.9 (block-Jacobi damping of the level smoothers; values in (0, 1] are accepted.  Measured on C3 / C4, 20 steps: 0.7 0.490 / 3.70 s,
                                          *      0.8 0.478 / 3.57, 0.9 0.469 / 3.48, 1.0 0.465 / 3.46; from 1.1 on the smoother diverges on some level — the cycle is no longer
                                          *      positive definite, the PCG breaks down and the LM iterates leave the exact-solve path — so 1.0 has no margin and 0.9 stays) */
    double mg_correction_scale;          /* 1.0 (1.6 saves 15-25 % of the iterations on the first linearisation at radius >= 1e6 and costs 5-10 % on later ones) */
    int32_t mg_first_passes;             /* 3 */
    int32_t mg_passes;                   /* 0 = by the prolongator: 2 where the transition above level 1 is smoothed (round 3: C3 / C4, 20 steps: 3: 0.453 / 3.56 s, 2: 0.438 / 2.37 s, 1: 0.586 s / -), 3 with plain aggregation; values 1..3 (measured with up to 5 on the final build, C3 / C4 20 steps: 3: 0.467 / 3.46 s, 4: 0.514 / 4.36 s, 5: 0.621 / 4.81 s) (one level less than with 2 at +6 % iterations: 3 % faster on C3 and C4, each level costs two ~10-us kernels) */
    int32_t mg_dense_max_nodes;          /* 512 (dense coarsest operator of <= 3072 unknowns) */
    int32_t mg_switch_iterations;        /* 400: every PCG starts with plain block-Jacobi (most LM systems — small trust regions, steps about to be
                                          *      rejected — need a few hundred cheap iterations); one that has not converged after this many iterations
                                          *      is restarted from its current iterate with the multigrid; a system predicted (from the previous LM step of the same
                                          *      solve, block-Jacobi iterations growing like sqrt(radius)) to need >= 1.75x this many gets the multigrid at once — or, while the
                                          *      step can still be rejected early (cg_early_tolerance), after a block-Jacobi prelude of at most 72 iterations: a step rejected at
                                          *      the first pause never pays for the operators (C3's step 4: 32 -> 2.6 ms); one predicted easier than that switches only after
                                          *      twice its prediction.  0: multigrid from the first iteration of every system. */
    double mg_loop_discount;             /* 3.0: in the matching of the levels ABOVE level 1 the switchable loop closures between two level-1 nodes count as (their number - this).  A single
                                          *      loop closure may be an outlier the solver switches off a few LM steps later, and an aggregate held together by nothing else then stops being a
                                          *      rigid piece — the hierarchy is built once per graph, before the switches are known; several loop closures between the same two pieces (revisited
                                          *      places) are not all outliers.  Measured, 20 LM steps: C3 (10 % outliers) 0.444 / 0.422 / 0.410 / 0.409 s and C4 2.38 / 1.72 / 1.47 / 1.60 s
                                          *      with 0 / 2 / 3 / 5; a 60k-keyframe graph WITHOUT outliers 0.71 / 0.87 / - / 1.02 s.  Relative-pose loop edges (no switch) are not discounted. */
    double mg_regroup_fraction;          /* 0.02: REGROUP — when, after an accepted step (not in the first three LM iterations), the switchable edges that have moved by > 0.5
                                          *      in s^2 since the hierarchy was built (outliers switched off) make up more than this fraction of ALL edges, the levels above level 1 are matched
                                          *      again along the couplings alive now (at most twice per solve; the keyframes' level-1 aggregates follow relative-pose edges only and are kept, as is
                                          *      level 1's structure).  On one GPU the host half of the rebuild (~25-30 ms for C3) runs on a worker thread while the solve goes on and is
                                          *      installed where multigrid operators are next built (both points depend on the solve's history only); with several ranks it happens there,
                                          *      synchronously.  0 disables.  Measured on C3: the late LM systems need 155 / 184 / 246 multigrid iterations after the
                                          *      regroup against 305 / 367 / 454 without */
    double mg_prolongation_damping;      /* 0.6: w_p of the smoothed prolongators Ps = (I - w_p D^-1 A) P (smoothed aggregation's 4 / (3 rho(D^-1 A)), rho ~ 2; from 0.9 on
                                          *      I - w_p D^-1 A is singular inside the spectrum and the Galerkin product degenerates: measured, scripts/research/r3_cycle_probe.py) */
    int32_t mg_smoothed_levels;          /* -1 = by size: 1 up to 500 000 keyframes, 0 beyond (C5, 1M keyframes: its coarse levels are bandwidth-bound and the denser operators cost more than the
                                          *      15 % of iterations they save: 8.5 -> 11.1 s).  n: the transitions level l -> l+1, l = 1 .. n, use the SMOOTHED prolongator (the level above is Ps^T A Ps: denser, and each
                                          *      such level costs two more row-product kernels per cycle); the keyframes -> level 1 transition stays the rigid one (its restriction and
                                          *      prolongation ride in the PCG's own kernels).  0: plain aggregation on every level (round 2's cycle).  Measured on a C3-structured
                                          *      20k-keyframe system with four levels, radius 1e6: 1237 PCG iterations without, 551 with the first transition smoothed, 487 with two */
    int32_t mg_min_keyframes_switchable; /* 5000 (round 6, see mg_min_keyframes; rounds 3-5: 8000): graphs WITH switchable loop closures (every graph the reference builds) take the multigrid already from this many keyframes (0: mg_min_keyframes for all).
                                          *      Round 3, after the cycle got cheaper (smoothed transition, 8-group rows, regroup): 10 LM iterations, two-level method vs multigrid, time ratio at
                                          *      4 000 / 6 000 / 8 000 / 12 000 / 18 000 keyframes — chain-like 0.93 / 0.97 / 2.05 / 1.30 / 1.03, no outliers 0.98 / 1.04 / 1.09 / 1.41 / 1.25,
                                          *      f = 1..5 + yaw weights 0.99 / 1.20 / 1.35 / 1.40 / 3.18, session-structured 1.41 / 1.20 / 1.43 / 1.13 / 1.70; PLAIN loops (no switches:
                                          *      level 1 then also follows the loops) 0.66 / 0.63 / 0.65 / 0.92 / 0.95 — those stay with mg_min_keyframes (profiles/r03_mg_crossover.txt) */
    /* device selection */
    int32_t device_id;                   /* -1: use the current HIP device */
    int32_t verbosity;                   /* 0 silent (minimizer_progress_to_stdout=false, :1271), 1 per-iteration line on stderr */
    /* round 5 (appended: the layout of everything above is unchanged) */
    int32_t cg_single_reduction;         /* 1: one GPU runs the PCG in its single-reduction (Chronopoulos-Gear) form — w = A u, both dot products r.u and u.w known right after the matvec,
                                          *    ONE partial-sum re-reduction per iteration (in the vector update) instead of two — whenever cg_rel_tolerance >= 1e-11 and the graph has at most 150 000
                                          *    keyframes (beyond that the four more vectors its update moves cost more than the head it saves: measured); the same iterates in exact
                                          *    arithmetic, its attainable accuracy is a little lower, so tighter tolerances (the 1e-12 / 1e-13 parity settings) keep the classic two-reduction
                                          *    form.  The two-level method: its FUSED three-kernel iteration (aggregates small enough for the update kernel's groups: every session-sized graph) runs the
     *    single-reduction form under the same two gates; its unfused form stays classic.  0: classic form everywhere.  Several ranks always run the single-reduction form. */
    int32_t mg_explicit_transfer;        /* 1: a multigrid level with a smoothed transition above it applies that transition through the EXPLICIT operator R^T = Ps - Dinv W (fp32 blocks on the
                                          *    pattern of W = A Ps, formed once per LM system): pre-smoothing step + smoothed restriction and smoothed prolongation + post-smoothing step
                                          *    become  v = x + Dinv (r - A x), r_next = R r  and  x = v + R^T x_next — two row products on that level per cycle, two launches fewer per
                                          *    PCG iteration; algebraically the same V(1,1) cycle.  0: the implicit form of rounds 3-4 (four row products with the level's own matrix) */
    int32_t cg_end_game;                 /* 1: once the polled r.z values predict fewer than two chunks of PCG iterations to go, the host stops running a chunk ahead and enqueues what the
                                          *    prediction asks for (one GPU); 0: always one full chunk in flight, as in rounds 1-4.  Changes how many early-exit kernels follow a stopped PCG,
                                          *    never its iterates. */
    int32_t cg_pause_always;             /* 0: both early-rejection pauses where a rejection is in the air (previous step rejected, or the last accepted step's relative decrease below 0.8), the
                                          *    first pause alone where the system is expensive (predicted block-Jacobi-equivalent iterations x keyframes >= 5.6e7: one wasted solve outweighs dozens
                                          *    of pauses), none elsewhere; 1: both at every LM system of graphs >= 20 000 keyframes / after the solve's first rejection (round 4's rule) */
    int32_t mg_smoothed_fine;            /* -1.  1: the transition keyframes -> level 1 is SMOOTHED as well (one GPU): Ps_0 = (I - w_p D^-1 A) P_0, level 1 = Ps_0^T A Ps_0 (one hop wider),
                                          *    inside the cycle z = D^-1 r + s Ps_0 V_1(Ps_0^T r) with Ps_0 as fp32 blocks in both orientations (two launches of their own per iteration
                                          *    instead of riding in the vector update and level 1's up-sweep).  It halves the multigrid iterations on every graph measured and costs denser
                                          *    levels: it pays while those are latency-sized.  -1 (default, round 6): BY THE DENSITY OF THE LEVELS IT MAKES — graphs of up to 80 000
                                          *    keyframes build the hierarchy with it, and it is kept when the sparse levels hold at most 280 000 blocks (round 5's measurements: +9 ... +52 %
                                          *    on graphs of 10 000 - 60 000 keyframes whose smoothed levels hold 43 000 - 375 000 blocks — config 2's graph with switchable loop closures
                                          *    0.107 -> 0.071 s — against -19 ... -43 % on C3, C4 and the f = 1..5 / plain-loop types at 714 000 - 3.9 M blocks,
                                          *    profiles/r05_smoothed_fine_measured.txt; round 6's soak of 36 random graphs found 300 000 - 375 000 blocks a mixed zone, -28 ... +18 %, and no loss below:
                                          *    profiles/r06_mid_soak.txt; C3, the benchmark graph: 734 000 blocks, not used).  0: never.  Several ranks: never. */
    /* round 6 (appended) */
    int32_t mg_dist_min_rows;            /* 8192.  Several ranks: a multigrid level with at least this many rows is DISTRIBUTED — every rank runs the cycle's kernels on the rows it owns and
                                          *    receives the rows of other ranks its kernels read by neighbour send/receive; a smaller level is run completely by every rank from gathered vectors
                                          *    (a level kernel stays at its 8-10 us latency floor up to ~10 000 rows, so distributing a smaller level buys no kernel time and costs two exchanges).  The hierarchy is the same on every rank, its aggregates never mix owners. */
    int32_t mg_fine_filter;              /* 0.  1: the smoothed keyframe transition (mg_smoothed_fine) forms its prolongator with a FILTERED matrix: Ps_0 = (I - w_p D_f^-1 A_f) P_0 where A_f keeps the
                                          *    odometry blocks and drops the switchable loop closures, the dropped blocks LUMPED into the diagonal so that A_f acts on the rigid-body modes as A
                                          *    does (A_f,ii = A_ii + sym(sum_dropped A_ij T_ji)); W_0 = A Ps_0 and level 1 = Ps_0^T A Ps_0 use the whole matrix.  Level 1 then does not take every
                                          *    loop closure of a neighbouring keyframe along.  Built and measured in round 6 (profiles/r06_fine_filter_measured.txt), OFF: on C3 the levels come out
                                          *    1.46x / 1.87x as dense as the plain hierarchy's instead of 2.4x / 2.9x, but the multigrid iterations only fall to 899 (unfiltered: 612, plain: 1 291;
                                          *    the 20 000-keyframe CPU probe had said 146 : 127 : 265) — 0.272 s against 0.247 s plain on C3, and slower than the unfiltered form on all four
                                          *    smaller graph types where the smoothed transition pays. */
    int32_t mg_dist_setup;               /* 1.  Several ranks: the multigrid's SET-UP (Galerkin products, block-Jacobi inverses, smoothed prolongators and transfer operators of every LM system) is
                                          *    distributed like its cycle — every rank forms the numbers of its OWN rows of every distributed level; the blocks two ranks share travel by neighbour
                                          *    send/receive (the parts of a block formed on several ranks are summed, in ascending rank order, where the block is needed), the first level every rank
                                          *    runs completely is gathered, the small levels above it and the dense inverse are formed by every rank.  0: rounds 3-5's set-up — level 1's blocks
                                          *    all-reduced (288 B per block), every level above formed by every rank. */
} pgo_options;

/* Per-iteration record; mirrors ceres::IterationSummary fields the BriefReport is built from. */
typedef struct pgo_iteration {
    int32_t iteration;
    int32_t step_is_valid;
    int32_t step_is_successful;
    int32_t cg_iterations;
    double cost;               /* cost after this iteration (Ceres convention, 0.5 sum r^2) */
    double cost_change;
    double model_cost_change;
    double relative_decrease;
    double gradient_max_norm;
    double step_norm;
    double trust_region_radius;
    double cg_residual;        /* final relative preconditioned residual of the PCG solve */
    double seconds;            /* wall seconds of this iteration (device-synchronised) */
    int32_t reason;            /* PGO_STEP_*: WHY the step ended the way step_is_valid / step_is_successful say */
    int32_t preconditioner;    /* PGO_PRECOND_* of the PCG that produced the step | PGO_PRECOND_RETRIED when a breakdown under the two-level method / the
                                * multigrid was answered by solving the same system again with plain block-Jacobi */
    /* round 5 (appended): where `seconds` went, from the library's host clock around its own (device-synchronised) phases */
    double seconds_system;     /* LM diagonal, Schur-reduced system, block-Jacobi factors, and the operators of the preconditioner built BEFORE the PCG starts */
    double seconds_pcg;        /* the PCG, its early-rejection pauses (candidate evaluations there) and operators built while it runs (in-flight switch to the multigrid) */
    double seconds_evaluate;   /* candidate point, its cost, model cost change (the full-accuracy evaluation; 0 for a step rejected at a pause) */
    double seconds_linearize;  /* K1 + K2 at the accepted point (0 for rejected steps) */
    int32_t cg_iterations_multigrid; /* of cg_iterations: those preconditioned by the aggregation multigrid */
    int32_t single_reduction;  /* 1: the PCG ran in its single-reduction form (pgo_options.cg_single_reduction) */
} pgo_iteration;

/* pgo_iteration.reason.  Ceres' IterationSummary only has step_is_valid / step_is_successful; an inexact linear solver adds ways for a step to fail that a log must
 * be able to tell apart (trust_region_minimizer.cc: HandleInvalidStep / HandleUnsuccessfulStep / convergence tests). */
enum {
    PGO_STEP_ACCEPTED = 0,               /* relative_decrease > min_relative_decrease (iteration 0 carries this value too) */
    PGO_STEP_REJECTED_RHO = 1,           /* full-accuracy step evaluated, relative_decrease <= min_relative_decrease (Ceres' unsuccessful step) */
    PGO_STEP_REJECTED_AT_PAUSE = 2,      /* rejected at an early-rejection pause (cg_early_tolerance / cg_mid_tolerance): the logged values are those at the pause */
    PGO_STEP_INVALID_FACTORIZATION = 3,  /* a damped 6x6 diagonal block was not positive definite (Ceres: linear solver failure) */
    PGO_STEP_INVALID_BREAKDOWN = 4,      /* the PCG broke down (p.Ap <= 0, r.z < 0 or NaN) — also after the block-Jacobi retry */
    PGO_STEP_INVALID_MODEL = 5,          /* model_cost_change <= 0 or not finite */
    PGO_STEP_CONVERGED = 6               /* parameter or function tolerance fired on this step: the minimiser stopped, the step is not applied (Ceres) */
};
enum { PGO_PRECOND_BLOCK_JACOBI = 0, PGO_PRECOND_TWO_LEVEL = 1, PGO_PRECOND_MULTIGRID = 2, PGO_PRECOND_RETRIED = 16 };

/* pgo_summary.iterations[] keeps the first PGO_MAX_ITERATION_LOG records (iteration 0 included); a stepping run that goes on longer
 * (pgo_lm_step with ignore_termination) still counts every iteration in num_iterations / cg_iterations — compare num_logged. */
#define PGO_MAX_ITERATION_LOG 256

/* Replaces ceres::Solver::Summary as read at src/PoseGraphSLAM.cpp:1905,1912,1921. */
typedef struct pgo_summary {
    int32_t termination_type;        /* PGO_CONVERGENCE / PGO_NO_CONVERGENCE / PGO_FAILURE */
    int32_t num_iterations;          /* LM iterations executed, NOT counting iteration 0 (Ceres' iterations.size()-1) */
    int32_t num_successful_steps;
    int32_t num_unsuccessful_steps;
    int64_t cg_iterations;           /* total PCG iterations */
    double initial_cost;
    double final_cost;
    double seconds_total;            /* whole pgo_solve, including H2D/D2H */
    double seconds_device;           /* LM loop only (state resident in HBM) */
    int32_t num_logged;              /* entries valid in iterations[] (entry 0 = iteration 0) */
    int32_t reserved_;
    pgo_iteration iterations[PGO_MAX_ITERATION_LOG];
    char message[256];
    int64_t cg_iterations_multigrid; /* of cg_iterations: those preconditioned by the aggregation multigrid (the rest: block-Jacobi / two-level) */
    int32_t pcg_retries;             /* LM systems whose PCG broke down under the two-level method / the multigrid and were solved again by plain block-Jacobi */
    int32_t reserved2_;
} pgo_summary;

/* ABI contract: pgo_options, pgo_iteration, pgo_summary and pgo_sharding_stats carry no size field — fields are only ever APPENDED, and the library copies the struct at ITS
 * size.  A caller must therefore be compiled against the header of the library it loads: check pgo_abi_version() == PGO_ABI_VERSION (bumped whenever a struct grows) or the
 * sizes below at start-up; never pass a struct compiled against an older header (the library would read past it).  INTEGRATION.md §2. */
#define PGO_ABI_VERSION 7
int32_t pgo_abi_version(void);
/* sizeof(pgo_options), sizeof(pgo_iteration), sizeof(pgo_summary) as the LIBRARY was compiled: a caller built against another header version finds out at start-up
 * instead of reading a shifted struct (which = 0, 1, 2; anything else: 0). */
int64_t pgo_abi_sizeof(int32_t which);

/* ------------------------------------------------------------------------------------------ */
/* lifecycle                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Fills `o` with the defaults documented above. */
void pgo_options_init(pgo_options* o);

/* Creates a persistent problem bound to one HIP device (replaces constructing `reint_problem`,
 * src/PoseGraphSLAM.h:197).  `opts` may be NULL (defaults).  Fails with PGO_ERR_NO_DEVICE when
 * no GPU is usable: there is NO CPU fallback in this library. */
int pgo_create(pgo_problem** out, const pgo_options* opts);
int pgo_destroy(pgo_problem* p);

/* Replaces the option assignments at src/PoseGraphSLAM.cpp:1268-1272. */
int pgo_set_options(pgo_problem* p, const pgo_options* opts);

/* Capacity hint (the reference pre-allocates 30000/30000, src/PoseGraphSLAM.cpp:17-25; here growable). */
int pgo_reserve(pgo_problem* p, int64_t n_nodes, int64_t n_edges);

/* ------------------------------------------------------------------------------------------ */
/* problem construction (library copies edge data at add time, as Ceres owns cost functions)   */
/* ------------------------------------------------------------------------------------------ */

/* n x `AddResidualBlock(SixDOFError::Create(c1_T_c2, weight), NULL, q[c1], t[c1], q[c2], t[c2])`
 * — reference src/PoseGraphSLAM.cpp:1629-1633 with functor src/CeresResidues.h:19-90.
 * c1_T_c2: n x 16 doubles (column-major Matrix4d).  weight: n doubles. */
int pgo_add_relpose_edges(pgo_problem* p, int64_t n, const int32_t* c1, const int32_t* c2,
                          const double* c1_T_c2, const double* weight);

/* n x `AddResidualBlock(SixDOFErrorWithSwitchingConstraints::Create(bTa, weight), NULL,
 *                       q[c1], t[c1], q[c2], t[c2], &switch[switch_idx])`
 * — reference src/PoseGraphSLAM.cpp:1550-1556 with functor src/CeresResidues.h:145-222.
 * `weight` is accepted and IGNORED exactly as the reference functor ignores it (CeresResidues.h:198);
 * it may be NULL.  Each switch index must be used by at most one edge (reference: switch e <-> loop edge e). */
int pgo_add_switchable_edges(pgo_problem* p, int64_t n, const int32_t* c1, const int32_t* c2,
                             const double* c1_T_c2, const double* weight, const int32_t* switch_idx);

/* REPLACES the current regulariser set: `RemoveResidualBlock` of all previous ones followed by
 * n x `AddResidualBlock(NodePoseRegularization::Create(target, weight), NULL, q[node], t[node])`
 * — reference src/PoseGraphSLAM.cpp:1803-1807,1844-1849 with functor src/CeresResidues.h:96-141. */
int pgo_set_node_regularizers(pgo_problem* p, int64_t n, const int32_t* node,
                              const double* target, const double* weight);

/* `SetParameterBlockConstant(q[node]); SetParameterBlockConstant(t[node])`
 * — reference src/PoseGraphSLAM.cpp:143-144 (load_state path).  Cumulative. */
int pgo_set_nodes_constant(pgo_problem* p, int64_t n, const int32_t* node);

/* Introspection. */
int pgo_num_relpose_edges(const pgo_problem* p, int64_t* n);
int pgo_num_switchable_edges(const pgo_problem* p, int64_t* n);
int pgo_num_regularizers(const pgo_problem* p, int64_t* n);

/* ------------------------------------------------------------------------------------------ */
/* solve                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* Replaces `ceres::Solve(reint_options, &reint_problem, &reint_summary)` — src/PoseGraphSLAM.cpp:1903.
 * In/out, in place: quat_xyzw[4*n_nodes], t[3*n_nodes], sw[n_switch].  Arrays are HOST pointers.
 * On PGO_FAILURE the arrays are left unmodified (Ceres IsSolutionUsable semantics).
 * Equivalent to pgo_solve_begin + pgo_lm_step until done + pgo_solve_end. */
int pgo_solve(pgo_problem* p, double* quat_xyzw, double* t, double* sw,
              int64_t n_nodes, int64_t n_switch, pgo_summary* summary);

/* Stepping form of the same solve (used by bench.py to time exactly K LM iterations with the state
 * resident in HBM).  begin: upload + (re)build device graph + iteration 0 (evaluate, Jacobi scaling,
 * gradient).  step: ONE trust-region iteration; *done = 1 when a Ceres termination test fired
 * (the step may still be called again when `ignore_termination` != 0).  end: final write-back. */
int pgo_solve_begin(pgo_problem* p, const double* quat_xyzw, const double* t, const double* sw,
                    int64_t n_nodes, int64_t n_switch);
int pgo_lm_step(pgo_problem* p, int32_t ignore_termination, int32_t* done);
int pgo_solve_end(pgo_problem* p, double* quat_xyzw, double* t, double* sw, pgo_summary* summary);

/* Parity hook = `ceres::Problem::Evaluate` at the given point (all outputs optional, HOST pointers):
 *   cost                      0.5 sum r^2 over all residual blocks
 *   residuals                 concatenated in block order: relpose edges (6 each, add order),
 *                             switchable edges (7 each, add order), regularisers (6 each)
 *   gradient                  J^T r in the tangent layout documented at the top (6*n_nodes + n_switch)
 * Runs K1 (+K2 when a gradient is asked for) on the device. */
int pgo_evaluate(pgo_problem* p, const double* quat_xyzw, const double* t, const double* sw,
                 int64_t n_nodes, int64_t n_switch,
                 double* cost, double* residuals, double* gradient);

/* Parity hook for the Jacobian blocks K1 produced at the last pgo_evaluate / pgo_solve_begin point.
 * For edge kind k (0 relpose, 1 switchable, 2 regulariser) copies, for edges [first, first+count):
 *   J1[count*36], J2[count*36]  row-major (rows = residual 0..5, cols = [dtheta(3), dt(3)] of c1 / c2;
 *                               regularisers: J1 only, J2 may be NULL)
 *   dr_ds[count*7]              switchable only: d r / d s   (NULL otherwise)
 * These are the Ceres tangent-space blocks (autodiff 6x4 times the 4x3 parameterization Jacobian). */
int pgo_get_jacobian_blocks(pgo_problem* p, int32_t kind, int64_t first, int64_t count,
                            double* J1, double* J2, double* dr_ds);

/* Parity hook for K2: the assembled (undamped, unscaled) normal matrix at the last linearisation.
 *   diag[n_nodes*36]     H_ii row-major (pose part BEFORE switch elimination: sum J^T J)
 *   grad[n_nodes*6]      g_i
 *   offdiag[E*36]        J1^T J2 per edge in internal edge order: relpose (add order) then switchable
 *   sw_c[E_s*12], sw_hss[E_s], sw_gs[E_s]   switch couplings [J1^T Js ; J2^T Js], Js^T Js, Js^T r
 * Any pointer may be NULL. */
int pgo_get_normal_blocks(pgo_problem* p, double* diag, double* grad, double* offdiag,
                          double* sw_c, double* sw_hss, double* sw_gs);

/* Parity hook for the manifold step (a4): `ceres::EigenQuaternionParameterization::Plus` (reference src/PoseGraphSLAM.cpp:1276,1352) of n
 * keyframes on the device, with the same device function the solver's candidate step uses:
 *   quat_out[i] = [sin|d_i| d_i/|d_i| ; cos|d_i|] (x) quat[i],   t_out[i] = t[i] + dt[i]        (delta: 6 doubles per keyframe, [dtheta, dt])
 * HOST arrays; t / t_out may be NULL. */
int pgo_manifold_plus(pgo_problem* p, int64_t n, const double* quat_xyzw, const double* t, const double* delta, double* quat_out, double* t_out);

/* Parity hook for K3: y = (H_reduced + damping) * x on the device, x,y HOST arrays of 6*n_nodes,
 * using the damping of the current trust-region radius.  Valid after pgo_solve_begin. */
int pgo_apply_normal_operator(pgo_problem* p, const double* x, double* y);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU (edge sharding).  One process per GPU.                             */
/* ------------------------------------------------------------------------------------------ */

#define PGO_COMM_ID_BYTES 128

/* Rank 0 produces an RCCL unique id (ncclGetUniqueId); the host (torch.distributed) broadcasts the
 * bytes; every rank then calls pgo_comm_init.  After that the edges added on this rank are this
 * rank's SHARD and the handle works on the keyframes those edges touch (a rank-local subgraph).
 * Keyframes touched by two or more ranks are shared: only THEIR rows travel — diagonal blocks +
 * gradient once per linearisation, and per CG iteration the matvec output (6 doubles per shared
 * keyframe) with both dot products of the iteration riding along, in a single
 * ncclAllReduce(sum, fp64) over xGMI (the PCG runs in Chronopoulos-Gear form on several ranks).  Deal the edges out with locality (sharding.py:
 * `spatial`) to keep the shared set small.  The contract on every rank: the same n_nodes / n_switch,
 * the same initial arrays, the same constant keyframes, every switch index on exactly one rank; a
 * rank without edges is fine.  pgo_solve returns the COMPLETE solution on every rank (each
 * keyframe from its owner, each switch from the rank holding its edge).  At most 52 ranks.  Calls that issue
 * collectives (solve*, evaluate, get_normal_blocks, apply_normal_operator, time_kernel) must be
 * made by all ranks in the same order. */
int pgo_comm_get_unique_id(uint8_t id[PGO_COMM_ID_BYTES]);
int pgo_comm_init(pgo_problem* p, int32_t rank, int32_t world_size, const uint8_t id[PGO_COMM_ID_BYTES]);
int pgo_comm_destroy(pgo_problem* p);

/* What travels (round 6).  Every data-path exchange is a NEIGHBOUR exchange: a rank sends another rank exactly the rows that one reads and does not own —
 *   per CG iteration   the partial rows of w = A u of the keyframes two ranks share, to the ranks sharing them (ncclSend / ncclRecv in one group), the parts summed in
 *                      ascending rank order on every rank (same bits everywhere), plus ONE all-reduce of the iteration's two dot products (2 doubles);
 *   per multigrid cycle  the halo rows of the level vectors: the hierarchy's aggregates never mix owners, every level is numbered owner-major, and a rank runs the level
 *                      kernels on its own rows only (pgo_options.mg_dist_min_rows; smaller levels are run completely by every rank from gathered vectors);
 *   per linearisation / LM system  diagonal blocks, gradient, reduced diagonal and right-hand side of the shared keyframes, the same way.
 * The multigrid's SET-UP is distributed the same way (pgo_options.mg_dist_setup): every rank forms the operators of its own rows of a distributed level, the blocks two ranks share
 * travel by neighbour send/receive, the first level every rank runs completely is gathered; the small levels above it and the dense inverse are formed by every rank. */

/* Bring-your-own collective (e.g. torch.distributed, MPI, or an in-process test harness): `fn` must all-reduce `count` doubles in
 * DEVICE memory in place across the `world_size` ranks (op 0 = sum, 2 = max; work enqueued before the call on `hip_stream` must be
 * honoured, and the result must be visible to work enqueued on it afterwards) and return 0 on success.  Replaces the RCCL
 * communicator of pgo_comm_init; same sharding contract. */
typedef int (*pgo_allreduce_fn)(void* ctx, double* device_buf, int64_t count, int32_t op, void* hip_stream);
int pgo_comm_init_custom(pgo_problem* p, int32_t rank, int32_t world_size, pgo_allreduce_fn fn, void* ctx);
/* ... and its neighbour exchange (optional; without it the library emulates the exchange through `fn`: an all-reduce of a zero-padded buffer holding every pair's segment —
 * correct, but it moves world x the bytes).  `fn` must deliver, for every peer q, the doubles send_buf[send_off[q] .. send_off[q+1]) of this rank to
 * recv_buf[recv_off[r] .. recv_off[r+1]) of rank q's call (r = this rank) — MPI_Alltoallv in doubles on DEVICE buffers, offsets as HOST arrays of world_size + 1 entries,
 * same stream contract as pgo_allreduce_fn.  The segment sizes agree on both sides by construction. */
typedef int (*pgo_exchange_fn)(void* ctx, const double* send_buf, const int64_t* send_off, double* recv_buf, const int64_t* recv_off, void* hip_stream);
int pgo_comm_set_exchange(pgo_problem* p, pgo_exchange_fn fn);

/* In-process communicator: the ranks are `world_size` handles of ONE process, each driven by its own thread (on one GPU, or on several GPUs of one node with peer access).
 * A collective is then a kernel that reads the peers' device buffers directly, ordered by events between the handles' streams — no host staging, no RCCL.  Create one group,
 * hand it to every rank's pgo_comm_init_local, destroy it after the last pgo_comm_destroy.  Same sharding contract; at most 16 ranks.  Every rank must make the same
 * collective-issuing calls in the same order, each from its own thread (a rank blocks at a host barrier until all have arrived). */
int pgo_local_group_create(int32_t world_size, void** group);
int pgo_local_group_abort(void* group);      /* a rank failed outside the library: releases the ranks waiting at a barrier (their calls return PGO_ERR_COMM) */
int pgo_local_group_destroy(void* group);
int pgo_comm_init_local(pgo_problem* p, int32_t rank, int32_t world_size, void* group);

/* What the rank-local handle looks like and what its exchanges moved (several ranks; zeros on one GPU).  Counters are reset by pgo_solve_begin. */
typedef struct pgo_sharding_stats {
    int32_t world, rank;
    int64_t keyframes_local, keyframes_owned, keyframes_shared;      /* keyframes the rank's edges touch; of them owned; of them touched by other ranks too */
    int64_t shared_global;                                           /* keyframes touched by >= 2 ranks, over all ranks (rows of round 5's union all-reduce) */
    int32_t mg_levels, mg_levels_distributed;                        /* sparse + dense levels of the hierarchy; sparse levels whose kernels run on the owner's rows only */
    int64_t mg_rows_total, mg_rows_own;                              /* rows of all sparse levels; rows this rank's level kernels work on (its own on distributed levels, all on the others) */
    int64_t mg_blocks_total, mg_blocks_own;                          /* the same in 6x6 blocks of the level matrices (+ transfer operators): what the level kernels stream */
    int64_t pcg_iterations;                                          /* PCG iterations of this solve so far */
    int64_t exchanges;                                               /* neighbour exchanges issued */
    int64_t allreduces;                                              /* all-reduces issued (any size) */
    double bytes_sent_neighbour;                                     /* payload this rank sent in neighbour exchanges */
    double bytes_allreduce;                                          /* payload of its all-reduces (buffer sizes) */
    double bytes_sent_per_mg_iteration;                              /* by the plans: rows this rank sends in ONE multigrid-preconditioned PCG iteration x 48 B (+ 16 B of scalars) */
    double bytes_sent_per_bj_iteration;                              /* ... in one block-Jacobi iteration */
    double bytes_round5_per_mg_iteration;                            /* what round 5's design all-reduced per multigrid iteration on the same graph: (6 shared_global + 2 + 6 n_1) x 8 B */
    double bytes_round5_per_bj_iteration;                            /* (6 shared_global + 2) x 8 B */
    int32_t exchanges_per_mg_iteration, exchanges_per_bj_iteration;  /* neighbour exchanges on the critical path of one iteration */
    /* the multigrid's set-up (pgo_options.mg_dist_setup; appended in ABI 7) */
    int32_t mg_setup_levels_own_rows;                                /* sparse levels whose operators this rank forms for its own rows only (0: the set-up is replicated) */
    int32_t mg_setup_exchanges;                                      /* collectives of one set-up: block exchanges + per such level the smoother's safety estimate (8 halo exchanges of its power method's iterate, one 3-double all-reduce); replicated: 1 all-reduce */
    int64_t mg_setup_blocks_total, mg_setup_blocks_own;              /* 6x6 blocks one set-up forms (level matrices, Ps, W, R^T of every sparse level): in all, and by THIS rank (replicated: all of them on every rank) */
    double bytes_sent_per_mg_setup;                                  /* by the plans: what this rank sends in the block exchanges of one set-up */
    double bytes_allreduce_replicated_setup;                         /* what the replicated set-up all-reduces per LM system on the same graph: level 1's blocks x 288 B */
} pgo_sharding_stats;
int pgo_get_sharding_stats(pgo_problem* p, pgo_sharding_stats* out);
/* Diagnostic for tests of the distributed set-up: sums of squares of what THIS rank's cycle kernels read of multigrid level `level` (1-based) after the last set-up —
 * out8 = {its rows' fp64 blocks, their fp32 copy, its rows' block-Jacobi inverses, its rows of R^T, its coarse rows of R (smoothed transition above; else 0), the dense
 * inverse (coarsest level only), 0, 0}.  The same numbers whichever way the set-up ran (pgo_options.mg_dist_setup), up to the order of the sums. */
int pgo_mg_level_norms(pgo_problem* p, int32_t level, double* out8);

/* ------------------------------------------------------------------------------------------ */
/* graph construction on the device from the raw VIO poses                   */
/* ------------------------------------------------------------------------------------------ */

/* Device-resident copy of the odometry poses `manager->getNodePose(i)` (w_M_i in its own world,
 * NodeDataManager.h:95; n x 16 doubles, column-major Matrix4d).  Append-only in the reference (poses are
 * never revised once published), so a trigger sends only the new ones: poses [first, first+n) with
 * first <= pgo_num_vio_poses (rewriting earlier entries is allowed). */
int pgo_set_vio_poses(pgo_problem* p, int64_t first, int64_t n, const double* w_M);
int pgo_num_vio_poses(const pgo_problem* p, int64_t* n);

/* The odometry-residue loop of the trigger, src/PoseGraphSLAM.cpp:1570-1639, for u in [u_begin, u_end) and
 * f = 1..f_max (reference: 5): skips u-f < 0 (:1588) and pairs with an endpoint whose set id is negative
 * (dead zone, :1583); measurement u_M_umf = w_M_u^-1 * w_M_umf (:1597-1599), weight 0.9^f * exp(-yaw^2/6) with
 * yaw = R2ypr(u_M_umf)(0) in degrees (:1603-1606; use_yaw_weight = 0 keeps 0.9^f only), all computed by one
 * kernel from the resident VIO poses, appended as `SixDOFError` blocks on (u, u-f) in the reference's order
 * (u outer, f inner).  node_set_id: one int per VIO pose (find_setID_of_world_i(which_world_is_this(u))), or
 * NULL = every keyframe usable.  n_added (optional) receives the number of edges appended. */
int pgo_add_odometry_edges_from_vio(pgo_problem* p, const int32_t* node_set_id, int64_t u_begin, int64_t u_end,
                                    int32_t f_max, int32_t use_yaw_weight, int64_t* n_added);

/* Initial guesses of the trigger, src/PoseGraphSLAM.cpp:1727-1786: for u in [u_begin, u_end),
 * pose_u = left[left_of_node[u - u_begin]] * w_M_u written as (xyzw, t) the way update_opt_variable_with
 * stores it (PoseManipUtils.cpp:87-98); `left` is a small table of Matrix4d (n_left x 16): w_T_last * w_M_last^-1
 * for keyframes chained from the last solved pose (:1770-1775) or wset_T_w for other worlds (:1777-1780);
 * a negative selector leaves quat/t of that keyframe untouched.  quat/t are the caller's FULL arrays
 * (entries u_begin.. are written). */
int pgo_initial_guess_from_vio(pgo_problem* p, int64_t n_left, const double* left, const int32_t* left_of_node,
                               int64_t u_begin, int64_t u_end, double* quat_xyzw, double* t);

/* Parity hook: the stored records of relative-pose edges [first, first+n) as the kernels consume them —
 * c1, c2 and (q_obs xyzw, t_obs, weight) = 8 doubles per edge, i.e. what SixDOFError's constructor keeps
 * (CeresResidues.h:22-28).  Any output may be NULL. */
int pgo_get_relpose_edge_records(const pgo_problem* p, int64_t first, int64_t n, int32_t* c1, int32_t* c2, double* record8);

/* Edge sharding policies behind the C-ABI (host only: no handle, no device): which rank gets which residual block.  A C++ caller deals its edges out with this
 * and then calls pgo_add_*_edges on every rank's handle with that rank's share.  What the policy decides is the number of keyframes shared between ranks, i.e. the
 * rows every exchange carries:
 *   PGO_PARTITION_CONTIGUOUS  a contiguous index range of every edge class per rank
 *   PGO_PARTITION_CHAIN       keyframes in `world` consecutive index ranges balanced by edge load; an edge follows its LATER endpoint
 *   PGO_PARTITION_SPATIAL     recursive coordinate bisection of the keyframe positions `t_xyz` into `world` cells of equal edge load; an edge follows its later endpoint
 * node_part [n_nodes] (may be NULL; contiguous: not written) receives the part of every keyframe — a regulariser goes to its keyframe's part (contiguous: rank 0) —,
 * rel_rank [n_rel] / sw_rank [n_sw] the rank of every edge.  Same results as solve_keyframe_pose_graph_amd/sharding.py (tests/test_partition_capi.py). */
enum { PGO_PARTITION_CONTIGUOUS = 0, PGO_PARTITION_CHAIN = 1, PGO_PARTITION_SPATIAL = 2 };
int pgo_partition_edges(int32_t policy, int32_t world, int64_t n_nodes, const double* t_xyz, int64_t n_rel, const int32_t* rel_c1, const int32_t* rel_c2,
                        int64_t n_sw, const int32_t* sw_c1, const int32_t* sw_c2, int32_t* node_part, int32_t* rel_rank, int32_t* sw_rank);

/* ------------------------------------------------------------------------------------------ */
/* measurement helpers (bench.py): HIP-event timing of the dominant kernel on the library's stream */
/* ------------------------------------------------------------------------------------------ */

/* Launches K1 (residual + Jacobian) `launches` times back-to-back on the library stream at the current
 * device state (valid after pgo_solve_begin) bracketed by hipEvents; returns the average
 * milliseconds per launch and the algorithmic bytes one launch moves (SURVEY.md §8d formula). */
int pgo_time_linearize_kernel(pgo_problem* p, int32_t launches, double* avg_ms, double* algorithmic_bytes);

/* Same for one PCG iteration (K3+K4) and the assembly (K2). which: 0 = K1, 1 = K2, 2 = one block-Jacobi PCG iteration (matvec + update),
 * 3 = K1 cost-only, 4 = the matvec of the iteration alone, 5 = its vector update alone, 6 = one MULTIGRID-preconditioned PCG iteration (matvec + update with
 * the restriction + every level kernel; graphs with a hierarchy only), 7 = its level kernels alone (several ranks: THIS rank's share of them, without the exchanges —
 * what the rank's GPU computes per cycle; call it rank by rank), 8 = the kernels of ONE multigrid set-up (the level operators of an LM system and the dense inverse;
 * several ranks: THIS rank's kernels without the block exchanges between them, every rank must call it; algorithmic_bytes 0).  algorithmic_bytes of 2/4/5/6/7: what THIS design moves
 * per iteration with every array counted once (matrix-free: compact edge-side records + index data + vectors + the fp32 block-Jacobi
 * factors; block-CSR: SURVEY.md 8d's assembled form). */
int pgo_time_kernel(pgo_problem* p, int32_t which, int32_t launches, double* avg_ms, double* algorithmic_bytes);

/* K0 (odometry records from the resident VIO poses, f = 1..f_max over ALL resident poses): HIP-event average per launch and the
 * algorithmic bytes 128 B x poses + (8 + 64) B x edges. */
int pgo_time_vio_odometry_kernel(pgo_problem* p, int32_t f_max, int32_t launches, double* avg_ms, double* algorithmic_bytes);

/* K6's dense inverse on its own (test and measurement hook): inverts the symmetric positive definite n x n matrix `a` (row-major, host)
 * with the blocked Gauss-Jordan kernels the two-level preconditioner uses for its coarse operator, `launches` times; `a_inv` (host) gets
 * the result, *avg_ms (may be NULL) the HIP-event average of one inversion.  PGO_ERR_NUMERIC when a pivot is not positive. */
int pgo_dense_spd_inverse(pgo_problem* p, int32_t n, const double* a, double* a_inv, int32_t launches, double* avg_ms);

/* Waits for everything the handle has in flight: its stream and — after a pgo_solve_begin that (re)built the device graph — the worker thread that prepares the multigrid
 * hierarchy's host half beside the build (otherwise installed where the solve first needs it).  bench.py calls it before its timed region starts. */
int pgo_device_synchronize(pgo_problem* p);

const char* pgo_strerror(int code);
/* "libpgo sources sha256:<64 hex digits>": the hash the build recipe (solve_keyframe_pose_graph_amd/_build.py: source_tree_hash) computed over the HIP sources, their headers,
 * this header and the compiler flags the library was built from — a caller (bench.py) recomputes it from its checkout to tell whether the binary it loaded was built from it. */
const char* pgo_build_info(void);
/* Text of the last error on this handle (HIP/RCCL error string); never NULL. */
const char* pgo_last_error(const pgo_problem* p);

#ifdef __cplusplus
}
#endif
#endif /* PGO_H_ */
