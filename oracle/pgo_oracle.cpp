// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and only as the
// checker / reported CPU baseline.  PARITY UNPINNED: the reference has no tests, fixtures or golden vectors
// for this path and its Ceres/Eigen/ROS dependencies are absent from this image, so the reference itself
// cannot be built or run here.  The oracle is pinned by tests/golden/ (50-digit mpmath
// restatement of the functor algebra, finite differences, and a second independent minimiser).
//
// pgo_oracle.cpp — CPU restatement of the reference's pose-graph solve:
//   * residual blocks and their order of parameters exactly as the reference adds them
//       SixDOFError                         src/PoseGraphSLAM.cpp:1629-1633   (q_u,t_u,q_{u-f},t_{u-f})
//       SixDOFErrorWithSwitchingConstraints src/PoseGraphSLAM.cpp:1550-1556   (q_b,t_b,q_a,t_a,s_e)
//       NodePoseRegularization              src/PoseGraphSLAM.cpp:1844-1849   (q,t)
//   * derivatives by forward-mode Jets as ceres::AutoDiffCostFunction<F,6|7,4,3,4,3[,1]> does
//     (src/CeresResidues.h:74,131,206), projected by EigenQuaternionParameterization::ComputeJacobian
//   * ceres::Solve with the options of src/PoseGraphSLAM.cpp:1268-1272 and Ceres defaults otherwise:
//     trust-region Levenberg-Marquardt, Jacobi scaling, SPARSE_NORMAL_CHOLESKY (exact factorisation).
//     Ceres is a third-party dependency, not vendored and unpinned (CMakeLists.txt:23; API usage implies
//     1.12 <= version <= 2.1); the loop below restates trust_region_minimizer.cc /
//     levenberg_marquardt_strategy.cc as published for 1.13-2.1.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "functors.hpp"
#include "sparse_chol.hpp"

namespace orc {

struct RelEdge { int c1, c2; Mat4d T; double w; };
struct SwEdge { int c1, c2; Mat4d T; double w; int sw; };
struct Prior { int node; Mat4d T; double w; };

struct Problem {
    std::vector<RelEdge> rel;
    std::vector<SwEdge> swe;
    std::vector<Prior> pri;
    std::vector<int> constant_nodes;
};

// Linearised residual block in the tangent space ([dtheta(3), dt(3)] per pose, then s).
struct Lin {
    double r[7];
    double J1[42];  // rows x 6, row-major (rows = 6 or 7)
    double J2[42];
    double Js[7];
};

// ---- AutoDiff of one block + projection by the 4x3 parameterization Jacobian ----
// ambient Jacobian layout returned to callers: rows x NPAR row-major with parameter order
// (q1[4], t1[3], q2[4], t2[3], [s]) as in the AutoDiffCostFunction template arguments.
static void project_pose_block(const double* Jamb, int rows, int npar, int qoff, int toff, const double* q, double* Jtan /*rows x 6*/) {
    double P[12];
    eigen_quaternion_plus_jacobian(q, P);
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += Jamb[r * npar + qoff + k] * P[k * 3 + c];
            Jtan[r * 6 + c] = s;
        }
        for (int c = 0; c < 3; ++c) Jtan[r * 6 + 3 + c] = Jamb[r * npar + toff + c];
    }
}

static void eval_relpose(const RelEdge& e, const double* q1, const double* t1, const double* q2, const double* t2,
                         double* r, double* Jamb /*6x14 or null*/, double* J1 /*6x6 or null*/, double* J2) {
    SixDOFError f(e.T, e.w);
    if (!Jamb && !J1) { f(q1, t1, q2, t2, r); return; }
    typedef Jet<14> J;
    J jq1[4], jt1[3], jq2[4], jt2[3], jr[6];
    for (int i = 0; i < 4; ++i) jq1[i] = J(q1[i], i);
    for (int i = 0; i < 3; ++i) jt1[i] = J(t1[i], 4 + i);
    for (int i = 0; i < 4; ++i) jq2[i] = J(q2[i], 7 + i);
    for (int i = 0; i < 3; ++i) jt2[i] = J(t2[i], 11 + i);
    f(jq1, jt1, jq2, jt2, jr);
    double amb[6 * 14];
    for (int i = 0; i < 6; ++i) { r[i] = jr[i].a; for (int k = 0; k < 14; ++k) amb[i * 14 + k] = jr[i].v[k]; }
    if (Jamb) std::memcpy(Jamb, amb, sizeof(amb));
    if (J1) { project_pose_block(amb, 6, 14, 0, 4, q1, J1); project_pose_block(amb, 6, 14, 7, 11, q2, J2); }
}

static void eval_switch(const SwEdge& e, const double* q1, const double* t1, const double* q2, const double* t2, const double* s,
                        double* r, double* Jamb /*7x15*/, double* J1 /*7x6*/, double* J2, double* Js /*7*/) {
    SixDOFErrorWithSwitchingConstraints f(e.T, e.w);
    if (!Jamb && !J1) { f(q1, t1, q2, t2, s, r); return; }
    typedef Jet<15> J;
    J jq1[4], jt1[3], jq2[4], jt2[3], js[1], jr[7];
    for (int i = 0; i < 4; ++i) jq1[i] = J(q1[i], i);
    for (int i = 0; i < 3; ++i) jt1[i] = J(t1[i], 4 + i);
    for (int i = 0; i < 4; ++i) jq2[i] = J(q2[i], 7 + i);
    for (int i = 0; i < 3; ++i) jt2[i] = J(t2[i], 11 + i);
    js[0] = J(s[0], 14);
    f(jq1, jt1, jq2, jt2, js, jr);
    double amb[7 * 15];
    for (int i = 0; i < 7; ++i) { r[i] = jr[i].a; for (int k = 0; k < 15; ++k) amb[i * 15 + k] = jr[i].v[k]; }
    if (Jamb) std::memcpy(Jamb, amb, sizeof(amb));
    if (J1) {
        project_pose_block(amb, 7, 15, 0, 4, q1, J1);
        project_pose_block(amb, 7, 15, 7, 11, q2, J2);
        for (int i = 0; i < 7; ++i) Js[i] = amb[i * 15 + 14];
    }
}

static void eval_prior(const Prior& e, const double* q1, const double* t1, double* r, double* Jamb /*6x7*/, double* J1 /*6x6*/) {
    NodePoseRegularization f(e.T, e.w);
    if (!Jamb && !J1) { f(q1, t1, r); return; }
    typedef Jet<7> J;
    J jq1[4], jt1[3], jr[6];
    for (int i = 0; i < 4; ++i) jq1[i] = J(q1[i], i);
    for (int i = 0; i < 3; ++i) jt1[i] = J(t1[i], 4 + i);
    f(jq1, jt1, jr);
    double amb[6 * 7];
    for (int i = 0; i < 6; ++i) { r[i] = jr[i].a; for (int k = 0; k < 7; ++k) amb[i * 7 + k] = jr[i].v[k]; }
    if (Jamb) std::memcpy(Jamb, amb, sizeof(amb));
    if (J1) project_pose_block(amb, 6, 7, 0, 4, q1, J1);
}

// ------------------------------------------------------------------------------------------------
// whole-problem evaluation (= ceres::Problem::Evaluate / the evaluator inside the minimizer)
// ------------------------------------------------------------------------------------------------
struct State {
    std::vector<double> q, t, s;  // 4N, 3N, S
};

static double evaluate_cost(const Problem& P, const State& x, double* residuals /*or null*/) {
    double cost = 0;
    double r[7];
    size_t off = 0;
    for (const RelEdge& e : P.rel) {
        eval_relpose(e, &x.q[4 * e.c1], &x.t[3 * e.c1], &x.q[4 * e.c2], &x.t[3 * e.c2], r, nullptr, nullptr, nullptr);
        for (int i = 0; i < 6; ++i) cost += r[i] * r[i];
        if (residuals) { std::memcpy(residuals + off, r, 6 * sizeof(double)); } off += 6;
    }
    for (const SwEdge& e : P.swe) {
        eval_switch(e, &x.q[4 * e.c1], &x.t[3 * e.c1], &x.q[4 * e.c2], &x.t[3 * e.c2], &x.s[e.sw], r, nullptr, nullptr, nullptr, nullptr);
        for (int i = 0; i < 7; ++i) cost += r[i] * r[i];
        if (residuals) { std::memcpy(residuals + off, r, 7 * sizeof(double)); } off += 7;
    }
    for (const Prior& e : P.pri) {
        eval_prior(e, &x.q[4 * e.node], &x.t[3 * e.node], r, nullptr, nullptr);
        for (int i = 0; i < 6; ++i) cost += r[i] * r[i];
        if (residuals) { std::memcpy(residuals + off, r, 6 * sizeof(double)); } off += 6;
    }
    return 0.5 * cost;
}

// num_threads > 1: the residual blocks are evaluated by that many OpenMP threads (what Ceres' `num_threads` does for the Jacobian evaluation;
// the reference leaves it at 1).  Used by bench.py's all-cores baseline only: the summation order of the cost then differs from the serial one.
static double linearize(const Problem& P, const State& x, std::vector<Lin>& rel, std::vector<Lin>& swe, std::vector<Lin>& pri, int num_threads = 1) {
    rel.resize(P.rel.size()); swe.resize(P.swe.size()); pri.resize(P.pri.size());
    double cost = 0;
#pragma omp parallel for num_threads(num_threads) reduction(+ : cost) schedule(static) if (num_threads > 1)
    for (size_t k = 0; k < P.rel.size(); ++k) {
        const RelEdge& e = P.rel[k]; Lin& L = rel[k];
        eval_relpose(e, &x.q[4 * e.c1], &x.t[3 * e.c1], &x.q[4 * e.c2], &x.t[3 * e.c2], L.r, nullptr, L.J1, L.J2);
        for (int i = 0; i < 6; ++i) cost += L.r[i] * L.r[i];
    }
#pragma omp parallel for num_threads(num_threads) reduction(+ : cost) schedule(static) if (num_threads > 1)
    for (size_t k = 0; k < P.swe.size(); ++k) {
        const SwEdge& e = P.swe[k]; Lin& L = swe[k];
        eval_switch(e, &x.q[4 * e.c1], &x.t[3 * e.c1], &x.q[4 * e.c2], &x.t[3 * e.c2], &x.s[e.sw], L.r, nullptr, L.J1, L.J2, L.Js);
        for (int i = 0; i < 7; ++i) cost += L.r[i] * L.r[i];
    }
    for (size_t k = 0; k < P.pri.size(); ++k) {
        const Prior& e = P.pri[k]; Lin& L = pri[k];
        eval_prior(e, &x.q[4 * e.node], &x.t[3 * e.node], L.r, nullptr, L.J1);
        for (int i = 0; i < 6; ++i) cost += L.r[i] * L.r[i];
    }
    return 0.5 * cost;
}

// J^T r in the tangent layout [6N | S] and the squared column norms of J in the same layout.
static void gradient_and_colnorms(const Problem& P, int N, int S, const std::vector<Lin>& rel, const std::vector<Lin>& swe,
                                  const std::vector<Lin>& pri, const std::vector<char>& node_free, std::vector<double>& g, std::vector<double>& cn) {
    g.assign((size_t)6 * N + S, 0.0);
    cn.assign((size_t)6 * N + S, 0.0);
    auto acc = [&](int node, const double* J, const double* r, int rows) {
        if (!node_free[node]) return;
        for (int c = 0; c < 6; ++c) {
            double gs = 0, ns = 0;
            for (int i = 0; i < rows; ++i) { gs += J[i * 6 + c] * r[i]; ns += J[i * 6 + c] * J[i * 6 + c]; }
            g[(size_t)6 * node + c] += gs; cn[(size_t)6 * node + c] += ns;
        }
    };
    for (size_t k = 0; k < P.rel.size(); ++k) { acc(P.rel[k].c1, rel[k].J1, rel[k].r, 6); acc(P.rel[k].c2, rel[k].J2, rel[k].r, 6); }
    for (size_t k = 0; k < P.swe.size(); ++k) {
        acc(P.swe[k].c1, swe[k].J1, swe[k].r, 7); acc(P.swe[k].c2, swe[k].J2, swe[k].r, 7);
        double gs = 0, ns = 0;
        for (int i = 0; i < 7; ++i) { gs += swe[k].Js[i] * swe[k].r[i]; ns += swe[k].Js[i] * swe[k].Js[i]; }
        g[(size_t)6 * N + P.swe[k].sw] += gs; cn[(size_t)6 * N + P.swe[k].sw] += ns;
    }
    for (size_t k = 0; k < P.pri.size(); ++k) acc(P.pri[k].node, pri[k].J1, pri[k].r, 6);
}

static void plus(const State& x, const std::vector<double>& delta, int N, int S, State& out) {
    out = x;
    for (int i = 0; i < N; ++i) {
        eigen_quaternion_plus(&x.q[4 * i], &delta[(size_t)6 * i], &out.q[4 * i]);
        for (int c = 0; c < 3; ++c) out.t[3 * i + c] = x.t[3 * i + c] + delta[(size_t)6 * i + 3 + c];
    }
    for (int k = 0; k < S; ++k) out.s[k] = x.s[k] + delta[(size_t)6 * N + k];
}

struct Options {
    int max_num_iterations = 10;
    int jacobi_scaling = 1;
    int max_num_consecutive_invalid_steps = 5;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    int verbosity = 0;
    int num_threads = 1;   // Ceres default, which the reference keeps
};

struct IterLog {
    int iteration, step_is_valid, step_is_successful, reserved;
    double cost, cost_change, model_cost_change, relative_decrease, gradient_max_norm, step_norm, trust_region_radius, seconds;
};

struct Summary {
    int termination_type;  // 0 CONVERGENCE 1 NO_CONVERGENCE 2 FAILURE
    int num_iterations, num_successful_steps, num_unsuccessful_steps;
    double initial_cost, final_cost, seconds_total, seconds_linear_solver, seconds_jacobian;
    long long chol_nnz_blocks;
    int num_logged, reserved;
    IterLog iterations[256];
    char message[256];
};

static double g_last_chol_flops = 0.0;      // floating-point operations of the last Cholesky factorisation (orc_last_cholesky_flops: bench.py reports the port's rate)
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The LM normal equations in the Jacobi-scaled space:  (S H S + D^2) y = S J^T r,  step = -y.
// Switch variables are eliminated exactly first (each appears in exactly one residual block,
// src/PoseGraphSLAM.cpp:1555, so their diagonal block is scalar) — identical to a Cholesky
// factorisation that orders those columns first.
struct NormalSolver {
    BlockCholesky chol;
    std::vector<int> cached_perm;
    bool have_perm = false;
    size_t cached_sig = 0;

    bool solve(const Problem& P, int N, int S, const std::vector<Lin>& rel, const std::vector<Lin>& swe, const std::vector<Lin>& pri,
               const std::vector<char>& node_free, const std::vector<char>& sw_free, const std::vector<double>& scale, const std::vector<double>& D2,
               const std::vector<double>& g, std::vector<double>& step /*scaled space, size 6N+S*/) {
        // active node compaction
        std::vector<int> cid(N, -1);
        int na = 0;
        for (int i = 0; i < N; ++i) if (node_free[i]) cid[i] = na++;
        BlockSPD A;
        A.n = na;
        A.diag.assign((size_t)na * 36, 0.0);
        std::vector<double> b((size_t)na * 6, 0.0);
        for (int i = 0; i < N; ++i) if (cid[i] >= 0) {
            for (int c = 0; c < 6; ++c) {
                A.diag[(size_t)cid[i] * 36 + c * 6 + c] = D2[(size_t)6 * i + c];
                b[(size_t)cid[i] * 6 + c] = scale[(size_t)6 * i + c] * g[(size_t)6 * i + c];
            }
        }
        auto add_diag = [&](int node, const double* Ja, const double* Jb, int rows, double f) {  // diag[node] += f * S Ja^T Jb S
            double* Dg = &A.diag[(size_t)cid[node] * 36];
            const double* sc = &scale[(size_t)6 * node];
            for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int i = 0; i < rows; ++i) s += Ja[i * 6 + a] * Jb[i * 6 + c];
                Dg[a * 6 + c] += f * sc[a] * sc[c] * s;
            }
        };
        auto add_off = [&](int n1, int n2, const double* J1, const double* J2, int rows) {
            A.oi.push_back(cid[n1]); A.oj.push_back(cid[n2]);
            A.oval.resize(A.oval.size() + 36);
            double* O = &A.oval[A.oval.size() - 36];
            const double* s1 = &scale[(size_t)6 * n1]; const double* s2 = &scale[(size_t)6 * n2];
            for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int i = 0; i < rows; ++i) s += J1[i * 6 + a] * J2[i * 6 + c];
                O[a * 6 + c] = s1[a] * s2[c] * s;
            }
            return O;
        };
        for (size_t k = 0; k < P.rel.size(); ++k) {
            const int n1 = P.rel[k].c1, n2 = P.rel[k].c2;
            if (n1 == n2) continue;
            if (cid[n1] >= 0) add_diag(n1, rel[k].J1, rel[k].J1, 6, 1.0);
            if (cid[n2] >= 0) add_diag(n2, rel[k].J2, rel[k].J2, 6, 1.0);
            if (cid[n1] >= 0 && cid[n2] >= 0) add_off(n1, n2, rel[k].J1, rel[k].J2, 6);
        }
        std::vector<double> sw_a(P.swe.size(), 0.0), sw_c1((size_t)P.swe.size() * 6, 0.0), sw_c2((size_t)P.swe.size() * 6, 0.0), sw_gs(P.swe.size(), 0.0);
        for (size_t k = 0; k < P.swe.size(); ++k) {
            const int n1 = P.swe[k].c1, n2 = P.swe[k].c2, si = P.swe[k].sw;
            const bool f1 = cid[n1] >= 0, f2 = cid[n2] >= 0;
            if (f1) add_diag(n1, swe[k].J1, swe[k].J1, 7, 1.0);
            if (f2) add_diag(n2, swe[k].J2, swe[k].J2, 7, 1.0);
            double* O = nullptr;
            if (f1 && f2) O = add_off(n1, n2, swe[k].J1, swe[k].J2, 7);
            if (!sw_free[si]) continue;
            // Schur-eliminate the switch column
            const double ss = scale[(size_t)6 * N + si];
            double hss = 0, gs = 0;
            for (int i = 0; i < 7; ++i) { hss += swe[k].Js[i] * swe[k].Js[i]; gs += swe[k].Js[i] * swe[k].r[i]; }
            const double a = ss * ss * hss + D2[(size_t)6 * N + si];
            sw_a[k] = a; sw_gs[k] = ss * gs;
            double* c1 = &sw_c1[k * 6]; double* c2 = &sw_c2[k * 6];
            for (int c = 0; c < 6; ++c) {
                double s1 = 0, s2 = 0;
                for (int i = 0; i < 7; ++i) { s1 += swe[k].J1[i * 6 + c] * swe[k].Js[i]; s2 += swe[k].J2[i * 6 + c] * swe[k].Js[i]; }
                c1[c] = f1 ? scale[(size_t)6 * n1 + c] * s1 * ss : 0.0;
                c2[c] = f2 ? scale[(size_t)6 * n2 + c] * s2 * ss : 0.0;
            }
            if (f1) { double* Dg = &A.diag[(size_t)cid[n1] * 36]; for (int a_ = 0; a_ < 6; ++a_) for (int c = 0; c < 6; ++c) Dg[a_ * 6 + c] -= c1[a_] * c1[c] / a; for (int c = 0; c < 6; ++c) b[(size_t)cid[n1] * 6 + c] -= c1[c] * sw_gs[k] / a; }
            if (f2) { double* Dg = &A.diag[(size_t)cid[n2] * 36]; for (int a_ = 0; a_ < 6; ++a_) for (int c = 0; c < 6; ++c) Dg[a_ * 6 + c] -= c2[a_] * c2[c] / a; for (int c = 0; c < 6; ++c) b[(size_t)cid[n2] * 6 + c] -= c2[c] * sw_gs[k] / a; }
            if (O) for (int a_ = 0; a_ < 6; ++a_) for (int c = 0; c < 6; ++c) O[a_ * 6 + c] -= c1[a_] * c2[c] / a;
        }
        for (size_t k = 0; k < P.pri.size(); ++k) if (cid[P.pri[k].node] >= 0) add_diag(P.pri[k].node, pri[k].J1, pri[k].J1, 6, 1.0);

        // reuse the ordering while the sparsity pattern is unchanged
        size_t sig = (size_t)na * 1315423911u + A.oi.size();
        const std::vector<int>* fp = (have_perm && sig == cached_sig) ? &cached_perm : nullptr;
        if (!chol.analyze_and_factor(A, fp)) return false;
        if (!fp) { cached_perm = chol.perm; cached_sig = sig; have_perm = true; }
        chol.solve(b.data());
        step.assign((size_t)6 * N + S, 0.0);
        for (int i = 0; i < N; ++i) if (cid[i] >= 0) for (int c = 0; c < 6; ++c) step[(size_t)6 * i + c] = -b[(size_t)cid[i] * 6 + c];
        for (size_t k = 0; k < P.swe.size(); ++k) {
            const int n1 = P.swe[k].c1, n2 = P.swe[k].c2, si = P.swe[k].sw;
            if (!sw_free[si]) continue;
            double acc = sw_gs[k];
            // y_s = (gs_scaled - c1.y1 - c2.y2)/a ; step = -y
            for (int c = 0; c < 6; ++c) {
                if (cid[n1] >= 0) acc -= sw_c1[k * 6 + c] * b[(size_t)cid[n1] * 6 + c];
                if (cid[n2] >= 0) acc -= sw_c2[k * 6 + c] * b[(size_t)cid[n2] * 6 + c];
            }
            step[(size_t)6 * N + si] = -acc / sw_a[k];
        }
        return true;
    }
};

static int solve(const Problem& P, const Options& opt, State& x, int N, int S, Summary& sum) {
    const double t_start = now_s();
    std::memset(&sum, 0, sizeof(sum));
    // free / fixed bookkeeping (Ceres removes constant and unreferenced parameter blocks from the program)
    std::vector<char> node_used(N, 0), sw_free(S, 0), node_free(N, 0);
    for (const RelEdge& e : P.rel) { node_used[e.c1] = 1; node_used[e.c2] = 1; }
    for (const SwEdge& e : P.swe) { node_used[e.c1] = 1; node_used[e.c2] = 1; sw_free[e.sw] = 1; }
    for (const Prior& e : P.pri) node_used[e.node] = 1;
    for (int i = 0; i < N; ++i) node_free[i] = node_used[i];
    for (int c : P.constant_nodes) if (c >= 0 && c < N) node_free[c] = 0;

    std::vector<Lin> rel, swe, pri;
    std::vector<double> g, cn, scale((size_t)6 * N + S, 1.0), diagonal, D2((size_t)6 * N + S, 0.0), step, delta((size_t)6 * N + S, 0.0);
    State cand;
    NormalSolver ns;

    auto x_norm_of = [&](const State& st) {
        double s2 = 0;
        for (int i = 0; i < N; ++i) if (node_free[i]) { for (int c = 0; c < 4; ++c) s2 += st.q[4 * i + c] * st.q[4 * i + c]; for (int c = 0; c < 3; ++c) s2 += st.t[3 * i + c] * st.t[3 * i + c]; }
        for (int k = 0; k < S; ++k) if (sw_free[k]) s2 += st.s[k] * st.s[k];
        return std::sqrt(s2);
    };
    auto gradient_max_norm_of = [&](const State& st) {
        // || Plus(x, -g) - x ||_inf   (trust_region_minimizer.cc, projected gradient)
        std::vector<double> ng(g.size());
        for (size_t i = 0; i < g.size(); ++i) ng[i] = -g[i];
        State xp; plus(st, ng, N, S, xp);
        double m = 0;
        for (int i = 0; i < N; ++i) if (node_free[i]) { for (int c = 0; c < 4; ++c) m = std::max(m, std::fabs(xp.q[4 * i + c] - st.q[4 * i + c])); for (int c = 0; c < 3; ++c) m = std::max(m, std::fabs(xp.t[3 * i + c] - st.t[3 * i + c])); }
        for (int k = 0; k < S; ++k) if (sw_free[k]) m = std::max(m, std::fabs(xp.s[k] - st.s[k]));
        return m;
    };

    // ---- iteration 0
    double tj = now_s();
    double x_cost = linearize(P, x, rel, swe, pri, opt.num_threads);
    gradient_and_colnorms(P, N, S, rel, swe, pri, node_free, g, cn);
    sum.seconds_jacobian += now_s() - tj;
    if (!std::isfinite(x_cost)) { sum.termination_type = 2; std::snprintf(sum.message, sizeof(sum.message), "initial cost not finite"); return 0; }
    if (opt.jacobi_scaling) for (size_t i = 0; i < scale.size(); ++i) scale[i] = 1.0 / (1.0 + std::sqrt(cn[i]));
    double x_norm = x_norm_of(x);
    double gmax = gradient_max_norm_of(x);
    double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int invalid = 0;
    sum.initial_cost = x_cost;
    auto log_iter = [&](const IterLog& it) { if (sum.num_logged < 256) sum.iterations[sum.num_logged++] = it; };
    { IterLog it{}; it.iteration = 0; it.step_is_valid = 1; it.step_is_successful = 1; it.cost = x_cost; it.gradient_max_norm = gmax; it.trust_region_radius = radius; it.seconds = now_s() - t_start; log_iter(it); }
    int iteration = 0;
    sum.termination_type = 1;
    while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (iteration >= opt.max_num_iterations) { sum.termination_type = 1; std::snprintf(sum.message, sizeof(sum.message), "Maximum number of iterations reached."); break; }
        if (gmax <= opt.gradient_tolerance) { sum.termination_type = 0; std::snprintf(sum.message, sizeof(sum.message), "Gradient tolerance reached."); break; }
        if (radius < opt.min_trust_region_radius) { sum.termination_type = 0; std::snprintf(sum.message, sizeof(sum.message), "Minimum trust region radius reached."); break; }
        ++iteration;
        const double t_it = now_s();
        IterLog it{}; it.iteration = iteration; it.trust_region_radius = radius;
        // LevenbergMarquardtStrategy::ComputeStep
        if (!reuse_diagonal) {
            diagonal.resize(cn.size());
            for (size_t i = 0; i < cn.size(); ++i) {
                const double d = scale[i] * scale[i] * cn[i];  // squared column norm of the SCALED Jacobian
                diagonal[i] = std::min(std::max(d, opt.min_lm_diagonal), opt.max_lm_diagonal);
            }
        }
        for (size_t i = 0; i < D2.size(); ++i) D2[i] = diagonal[i] / radius;
        double tl = now_s();
        bool ok = ns.solve(P, N, S, rel, swe, pri, node_free, sw_free, scale, D2, g, step);
        sum.seconds_linear_solver += now_s() - tl;
        sum.chol_nnz_blocks = ns.chol.nnz_blocks;
        g_last_chol_flops = ns.chol.flops;
        double model_cost_change = 0;
        if (ok) {
            for (double v : step) if (!std::isfinite(v)) { ok = false; break; }
        }
        if (ok) {
            // model_cost_change = -(J s)^T (r + J s / 2), J the scaled Jacobian: J_scaled * step = J * (scale .* step)
            for (size_t i = 0; i < delta.size(); ++i) delta[i] = step[i] * scale[i];
            double mc = 0;
            auto blk = [&](const double* J, int node, int rows, double* u) { if (!node_free[node]) return; for (int i = 0; i < rows; ++i) { double s = 0; for (int c = 0; c < 6; ++c) s += J[i * 6 + c] * delta[(size_t)6 * node + c]; u[i] += s; } };
            for (size_t k = 0; k < P.rel.size(); ++k) { double u[7] = {0}; blk(rel[k].J1, P.rel[k].c1, 6, u); blk(rel[k].J2, P.rel[k].c2, 6, u); for (int i = 0; i < 6; ++i) mc += u[i] * (rel[k].r[i] + 0.5 * u[i]); }
            for (size_t k = 0; k < P.swe.size(); ++k) { double u[7] = {0}; blk(swe[k].J1, P.swe[k].c1, 7, u); blk(swe[k].J2, P.swe[k].c2, 7, u); const double ds = delta[(size_t)6 * N + P.swe[k].sw]; for (int i = 0; i < 7; ++i) { u[i] += swe[k].Js[i] * ds; mc += u[i] * (swe[k].r[i] + 0.5 * u[i]); } }
            for (size_t k = 0; k < P.pri.size(); ++k) { double u[7] = {0}; blk(pri[k].J1, P.pri[k].node, 6, u); for (int i = 0; i < 6; ++i) mc += u[i] * (pri[k].r[i] + 0.5 * u[i]); }
            model_cost_change = -mc;
            if (!(model_cost_change > 0.0)) ok = false;
        }
        it.model_cost_change = model_cost_change;
        if (!ok) {
            // HandleInvalidStep
            it.step_is_valid = 0;
            ++invalid;
            if (invalid >= opt.max_num_consecutive_invalid_steps) { sum.termination_type = 2; std::snprintf(sum.message, sizeof(sum.message), "Number of consecutive invalid steps more than max_num_consecutive_invalid_steps"); it.cost = x_cost; it.seconds = now_s() - t_it; log_iter(it); break; }
            radius *= 0.5; reuse_diagonal = true;  // StepIsInvalid
            it.cost = x_cost; it.gradient_max_norm = gmax; it.seconds = now_s() - t_it; log_iter(it);
            ++sum.num_unsuccessful_steps;
            continue;
        }
        invalid = 0;
        it.step_is_valid = 1;
        plus(x, delta, N, S, cand);
        const double cand_cost = evaluate_cost(P, cand, nullptr);
        // ParameterToleranceReached
        double sn2 = 0;
        for (int i = 0; i < N; ++i) if (node_free[i]) { for (int c = 0; c < 4; ++c) { const double d = x.q[4 * i + c] - cand.q[4 * i + c]; sn2 += d * d; } for (int c = 0; c < 3; ++c) { const double d = x.t[3 * i + c] - cand.t[3 * i + c]; sn2 += d * d; } }
        for (int k = 0; k < S; ++k) if (sw_free[k]) { const double d = x.s[k] - cand.s[k]; sn2 += d * d; }
        const double step_norm = std::sqrt(sn2);
        it.step_norm = step_norm;
        it.cost_change = x_cost - cand_cost;
        it.relative_decrease = it.cost_change / model_cost_change;
        if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
            sum.termination_type = 0; std::snprintf(sum.message, sizeof(sum.message), "Parameter tolerance reached.");
            it.cost = x_cost; it.gradient_max_norm = gmax; it.seconds = now_s() - t_it; log_iter(it); break;
        }
        // FunctionToleranceReached
        if (std::fabs(it.cost_change) <= opt.function_tolerance * x_cost) {
            sum.termination_type = 0; std::snprintf(sum.message, sizeof(sum.message), "Function tolerance reached.");
            it.cost = x_cost; it.gradient_max_norm = gmax; it.seconds = now_s() - t_it; log_iter(it); break;
        }
        if (std::isfinite(cand_cost) && it.relative_decrease > opt.min_relative_decrease) {
            // HandleSuccessfulStep
            x = cand; x_cost = cand_cost; x_norm = x_norm_of(x);
            tj = now_s();
            linearize(P, x, rel, swe, pri, opt.num_threads);
            gradient_and_colnorms(P, N, S, rel, swe, pri, node_free, g, cn);
            sum.seconds_jacobian += now_s() - tj;
            gmax = gradient_max_norm_of(x);
            it.step_is_successful = 1;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));  // StepAccepted
            radius = std::min(opt.max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false;
            ++sum.num_successful_steps;
        } else {
            // HandleUnsuccessfulStep -> StepRejected
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            ++sum.num_unsuccessful_steps;
        }
        it.cost = x_cost; it.gradient_max_norm = gmax; it.seconds = now_s() - t_it;
        log_iter(it);
        if (opt.verbosity) std::fprintf(stderr, "[oracle] it %3d cost %.12e dcost %.3e rho %.3e |step| %.3e radius %.3e %s\n", iteration, x_cost, it.cost_change, it.relative_decrease, step_norm, radius, it.step_is_successful ? "ok" : "REJ");
    }
    sum.num_iterations = iteration;
    sum.final_cost = x_cost;
    sum.seconds_total = now_s() - t_start;
    return 0;
}

}  // namespace orc

// ================================================================================================
// C interface for ctypes (tests/, smoke(), bench.py cpu_baseline)
// ================================================================================================
using namespace orc;

extern "C" {

// flops of the most recent block Cholesky factorisation of this process (2 * 216 per 6x6 block update, oracle/sparse_chol.hpp)
double orc_last_cholesky_flops(void) { return g_last_chol_flops; }

// --- single-block evaluation (golden-vector checks) ---
// ambient Jacobians are what AutoDiffCostFunction::Evaluate returns (row-major rows x {14,15,7});
// tangent blocks are after EigenQuaternionParameterization::ComputeJacobian.
int orc_eval_relpose(const double* q1, const double* t1, const double* q2, const double* t2, const double* T16, double w,
                     double* r6, double* Jamb_6x14, double* J1_6x6, double* J2_6x6) {
    RelEdge e; std::memcpy(e.T.d, T16, sizeof(e.T.d)); e.w = w; e.c1 = 0; e.c2 = 1;
    double j1[42], j2[42], amb[6 * 14];
    eval_relpose(e, q1, t1, q2, t2, r6, amb, j1, j2);
    if (Jamb_6x14) std::memcpy(Jamb_6x14, amb, sizeof(amb));
    if (J1_6x6) std::memcpy(J1_6x6, j1, 36 * sizeof(double));
    if (J2_6x6) std::memcpy(J2_6x6, j2, 36 * sizeof(double));
    return 0;
}
int orc_eval_switch(const double* q1, const double* t1, const double* q2, const double* t2, const double* s, const double* T16, double w,
                    double* r7, double* Jamb_7x15, double* J1_7x6, double* J2_7x6, double* Js7) {
    SwEdge e; std::memcpy(e.T.d, T16, sizeof(e.T.d)); e.w = w; e.c1 = 0; e.c2 = 1; e.sw = 0;
    double j1[42], j2[42], js[7], amb[7 * 15];
    eval_switch(e, q1, t1, q2, t2, s, r7, amb, j1, j2, js);
    if (Jamb_7x15) std::memcpy(Jamb_7x15, amb, sizeof(amb));
    if (J1_7x6) std::memcpy(J1_7x6, j1, sizeof(j1));
    if (J2_7x6) std::memcpy(J2_7x6, j2, sizeof(j2));
    if (Js7) std::memcpy(Js7, js, sizeof(js));
    return 0;
}
int orc_eval_prior(const double* q1, const double* t1, const double* T16, double w, double* r6, double* Jamb_6x7, double* J1_6x6) {
    Prior e; std::memcpy(e.T.d, T16, sizeof(e.T.d)); e.w = w; e.node = 0;
    double j1[42], amb[42];
    eval_prior(e, q1, t1, r6, amb, j1);
    if (Jamb_6x7) std::memcpy(Jamb_6x7, amb, sizeof(amb));
    if (J1_6x6) std::memcpy(J1_6x6, j1, 36 * sizeof(double));
    return 0;
}
// residual-only (double arithmetic, no Jets)
int orc_residual_relpose(const double* q1, const double* t1, const double* q2, const double* t2, const double* T16, double w, double* r6) {
    RelEdge e; std::memcpy(e.T.d, T16, sizeof(e.T.d)); e.w = w; eval_relpose(e, q1, t1, q2, t2, r6, nullptr, nullptr, nullptr); return 0;
}
int orc_mat_to_quat(const double* T16, double* q_xyzw) {
    Mat4d T; std::memcpy(T.d, T16, sizeof(T.d));
    Mat3<double> R; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R.m[r][c] = T(r, c);
    const Quat<double> q = fromRotationMatrix(R);
    q_xyzw[0] = q.x; q_xyzw[1] = q.y; q_xyzw[2] = q.z; q_xyzw[3] = q.w; return 0;
}
int orc_quat_to_mat(const double* q_xyzw, double* R9_rowmajor) {
    const Quat<double> q{q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]};
    const Mat3<double> R = toRotationMatrix(q);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R9_rowmajor[r * 3 + c] = R.m[r][c]; return 0;
}
int orc_quat_plus(const double* q, const double* delta3, double* q_out) { eigen_quaternion_plus(q, delta3, q_out); return 0; }
int orc_quat_plus_jacobian(const double* q, double* jac_4x3) { eigen_quaternion_plus_jacobian(q, jac_4x3); return 0; }

// --- graph construction from raw VIO poses (restates src/PoseGraphSLAM.cpp:1570-1639 and :1770-1786; test infrastructure) ---
static Mat4d mul4(const Mat4d& A, const Mat4d& B) {
    Mat4d C;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { double a = 0; for (int k = 0; k < 4; ++k) a += A(r, k) * B(k, c); C(r, c) = a; }
    return C;
}
// PoseManipUtils::R2ypr (src/utils/PoseManipUtils.cpp:143-158), first component only, degrees
static double r2ypr_yaw_deg(const Mat4d& T) { return std::atan2(T(1, 0), T(0, 0)) / M_PI * 180.0; }

// The odometry-residue loop.  Outputs sized for (u_end-u_begin)*f_max entries; returns the number of edges written.
// T_out holds the Matrix4d handed to SixDOFError::Create (:1620), w_out the weight (:1603-1606).
int64_t orc_odometry_edges_from_vio(int64_t n_vio, const double* w_M, const int32_t* set_id, int64_t u_begin, int64_t u_end, int f_max, int use_yaw,
                                    int32_t* c1, int32_t* c2, double* T_out, double* w_out) {
    int64_t n = 0;
    for (int64_t u = u_begin; u < u_end && u < n_vio; ++u) {
        for (int f = 1; f <= f_max; ++f) {
            if (u - f < 0) continue;                                               // :1588-1591
            if (set_id && (set_id[u] < 0 || set_id[u - f] < 0)) continue;          // :1583-1586
            Mat4d w_M_u, w_M_umf;
            std::memcpy(w_M_u.d, w_M + 16 * u, sizeof(w_M_u.d));
            std::memcpy(w_M_umf.d, w_M + 16 * (u - f), sizeof(w_M_umf.d));
            const Mat4d u_M_umf = mul4(inverse4(w_M_u), w_M_umf);                 // :1597-1599
            double odom_edge_weight = 1.0;
            odom_edge_weight *= std::pow(0.9, f);                                  // :1604
            if (use_yaw) { const double yaw = r2ypr_yaw_deg(u_M_umf); odom_edge_weight *= std::exp(-yaw * yaw / 6.); }   // :1605-1606
            c1[n] = (int32_t)u; c2[n] = (int32_t)(u - f);
            std::memcpy(T_out + 16 * n, u_M_umf.d, sizeof(u_M_umf.d));
            w_out[n] = odom_edge_weight;
            ++n;
        }
    }
    return n;
}

// Initial guess of keyframe u: left * w_M_u stored as (xyzw, t) (update_opt_variable_with -> eigenmat_to_raw_xyzw, PoseManipUtils.cpp:87-98)
int orc_initial_guess_from_vio(int64_t n_left, const double* left, const int32_t* left_of_node, const double* w_M, int64_t u_begin, int64_t u_end,
                               double* quat, double* t) {
    for (int64_t u = u_begin; u < u_end; ++u) {
        const int sel = left_of_node[u - u_begin];
        if (sel < 0) continue;
        if (sel >= n_left) return -1;
        Mat4d L, M;
        std::memcpy(L.d, left + 16 * sel, sizeof(L.d)); std::memcpy(M.d, w_M + 16 * u, sizeof(M.d));
        const Mat4d P = mul4(L, M);
        orc_mat_to_quat(P.d, quat + 4 * u);
        t[3 * u] = P(0, 3); t[3 * u + 1] = P(1, 3); t[3 * u + 2] = P(2, 3);
    }
    return 0;
}

// --- problem ---
void* orc_create() { return new Problem(); }
void orc_destroy(void* h) { delete (Problem*)h; }
int orc_add_relpose_edges(void* h, int64_t n, const int32_t* c1, const int32_t* c2, const double* T, const double* w) {
    Problem* P = (Problem*)h;
    for (int64_t k = 0; k < n; ++k) { RelEdge e; e.c1 = c1[k]; e.c2 = c2[k]; std::memcpy(e.T.d, T + 16 * k, sizeof(e.T.d)); e.w = w[k]; P->rel.push_back(e); }
    return 0;
}
int orc_add_switchable_edges(void* h, int64_t n, const int32_t* c1, const int32_t* c2, const double* T, const double* w, const int32_t* sw) {
    Problem* P = (Problem*)h;
    for (int64_t k = 0; k < n; ++k) { SwEdge e; e.c1 = c1[k]; e.c2 = c2[k]; std::memcpy(e.T.d, T + 16 * k, sizeof(e.T.d)); e.w = w ? w[k] : 1.0; e.sw = sw[k]; P->swe.push_back(e); }
    return 0;
}
int orc_set_node_regularizers(void* h, int64_t n, const int32_t* node, const double* T, const double* w) {
    Problem* P = (Problem*)h; P->pri.clear();
    for (int64_t k = 0; k < n; ++k) { Prior e; e.node = node[k]; std::memcpy(e.T.d, T + 16 * k, sizeof(e.T.d)); e.w = w[k]; P->pri.push_back(e); }
    return 0;
}
int orc_set_nodes_constant(void* h, int64_t n, const int32_t* node) { Problem* P = (Problem*)h; for (int64_t k = 0; k < n; ++k) P->constant_nodes.push_back(node[k]); return 0; }

static State make_state(const double* q, const double* t, const double* s, int64_t N, int64_t S) {
    State x; x.q.assign(q, q + 4 * N); x.t.assign(t, t + 3 * N); x.s.assign(s, s + S); return x;
}

// = ceres::Problem::Evaluate.  residual order: relpose, switchable, regularisers.  gradient: tangent layout [6N | S].
int orc_evaluate(void* h, const double* q, const double* t, const double* s, int64_t N, int64_t S, double* cost, double* residuals, double* gradient) {
    Problem* P = (Problem*)h;
    State x = make_state(q, t, s, N, S);
    if (gradient) {
        std::vector<Lin> rel, swe, pri; std::vector<double> g, cn; std::vector<char> nf(N, 1);
        for (int c : P->constant_nodes) if (c >= 0 && c < N) nf[c] = 0;
        const double c = linearize(*P, x, rel, swe, pri);
        gradient_and_colnorms(*P, (int)N, (int)S, rel, swe, pri, nf, g, cn);
        std::memcpy(gradient, g.data(), g.size() * sizeof(double));
        if (cost) *cost = c;
        if (residuals) evaluate_cost(*P, x, residuals);
        return 0;
    }
    const double c = evaluate_cost(*P, x, residuals);
    if (cost) *cost = c;
    return 0;
}

// Tangent-space Jacobian blocks of every residual block at x (same layout as pgo_get_jacobian_blocks).
// kind 0: J1,J2 6x6 per relpose edge; kind 1: first 6 rows of J1,J2 (7th row is identically zero) + dr_ds[7]; kind 2: J1.
int orc_jacobian_blocks(void* h, const double* q, const double* t, const double* s, int64_t N, int64_t S, int kind, double* J1, double* J2, double* drds) {
    Problem* P = (Problem*)h;
    State x = make_state(q, t, s, N, S);
    std::vector<Lin> rel, swe, pri;
    linearize(*P, x, rel, swe, pri);
    const std::vector<Lin>& L = kind == 0 ? rel : (kind == 1 ? swe : pri);
    for (size_t k = 0; k < L.size(); ++k) {
        if (J1) std::memcpy(J1 + 36 * k, L[k].J1, 36 * sizeof(double));
        if (J2 && kind != 2) std::memcpy(J2 + 36 * k, L[k].J2, 36 * sizeof(double));
        if (drds && kind == 1) std::memcpy(drds + 7 * k, L[k].Js, 7 * sizeof(double));
    }
    return 0;
}

// Dense J^T J (tangent layout, (6N+S)^2 row-major) for small graphs — golden (4) in SURVEY.md §8c.
int orc_dense_normal_matrix(void* h, const double* q, const double* t, const double* s, int64_t N, int64_t S, double* H) {
    Problem* P = (Problem*)h;
    State x = make_state(q, t, s, N, S);
    std::vector<Lin> rel, swe, pri;
    linearize(*P, x, rel, swe, pri);
    const int64_t n = 6 * N + S;
    std::fill(H, H + n * n, 0.0);
    auto add = [&](const double* Ja, int64_t oa, int wa, const double* Jb, int64_t ob, int wb, int rows) {
        for (int a = 0; a < wa; ++a) for (int b = 0; b < wb; ++b) { double sacc = 0; for (int i = 0; i < rows; ++i) sacc += Ja[i * wa + a] * Jb[i * wb + b]; H[(oa + a) * n + ob + b] += sacc; }
    };
    for (size_t k = 0; k < P->rel.size(); ++k) {
        const int64_t o1 = 6 * P->rel[k].c1, o2 = 6 * P->rel[k].c2;
        add(rel[k].J1, o1, 6, rel[k].J1, o1, 6, 6); add(rel[k].J2, o2, 6, rel[k].J2, o2, 6, 6);
        add(rel[k].J1, o1, 6, rel[k].J2, o2, 6, 6); add(rel[k].J2, o2, 6, rel[k].J1, o1, 6, 6);
    }
    for (size_t k = 0; k < P->swe.size(); ++k) {
        const int64_t o1 = 6 * P->swe[k].c1, o2 = 6 * P->swe[k].c2, os = 6 * N + P->swe[k].sw;
        add(swe[k].J1, o1, 6, swe[k].J1, o1, 6, 7); add(swe[k].J2, o2, 6, swe[k].J2, o2, 6, 7);
        add(swe[k].J1, o1, 6, swe[k].J2, o2, 6, 7); add(swe[k].J2, o2, 6, swe[k].J1, o1, 6, 7);
        add(swe[k].J1, o1, 6, swe[k].Js, os, 1, 7); add(swe[k].Js, os, 1, swe[k].J1, o1, 6, 7);
        add(swe[k].J2, o2, 6, swe[k].Js, os, 1, 7); add(swe[k].Js, os, 1, swe[k].J2, o2, 6, 7);
        add(swe[k].Js, os, 1, swe[k].Js, os, 1, 7);
    }
    for (size_t k = 0; k < P->pri.size(); ++k) { const int64_t o1 = 6 * P->pri[k].node; add(pri[k].J1, o1, 6, pri[k].J1, o1, 6, 6); }
    return 0;
}

void orc_options_init(Options* o) { *o = Options(); }

// = ceres::Solve (src/PoseGraphSLAM.cpp:1903).  In/out arrays like pgo_solve.
// Symbolic Cholesky of the problem's Schur-reduced normal matrix (pattern only: one block row per free keyframe touched by a residual block, one off-diagonal block per edge between two
// of them), with the port's own AMD ordering: fill in 6x6 blocks and the floating-point operations its numeric factorisation takes — without factorising.  bench.py reports it
// for the full C3 graph beside the measured time: what any direct solver with this ordering has to do per LM iteration.
int orc_cholesky_symbolic(void* h, int64_t N, long long* nnz_blocks, double* flops) {
    const Problem& P = *(Problem*)h;
    std::vector<char> used((size_t)N, 0);
    for (const RelEdge& e : P.rel) { used[e.c1] = 1; used[e.c2] = 1; }
    for (const SwEdge& e : P.swe) { used[e.c1] = 1; used[e.c2] = 1; }
    for (const Prior& pr : P.pri) used[pr.node] = 1;
    for (int c : P.constant_nodes) if (c >= 0 && c < N) used[c] = 0;
    std::vector<int> cid((size_t)N, -1);
    int na = 0;
    for (int64_t i = 0; i < N; ++i) if (used[i]) cid[i] = na++;
    BlockSPD A;
    A.n = na;
    A.diag.assign((size_t)na * 36, 0.0);
    auto off = [&](int a, int b) { if (a != b && cid[a] >= 0 && cid[b] >= 0) { A.oi.push_back(cid[a]); A.oj.push_back(cid[b]); } };
    for (const RelEdge& e : P.rel) off(e.c1, e.c2);
    for (const SwEdge& e : P.swe) off(e.c1, e.c2);
    A.oval.assign(A.oi.size() * 36, 0.0);
    BlockCholesky chol;
    chol.symbolic_only = true;
    if (!chol.analyze_and_factor(A, nullptr)) return 1;
    if (nnz_blocks) *nnz_blocks = chol.nnz_blocks;
    if (flops) *flops = chol.flops;
    return 0;
}

int orc_solve(void* h, const Options* opt, double* q, double* t, double* s, int64_t N, int64_t S, Summary* sum) {
    Problem* P = (Problem*)h;
    State x = make_state(q, t, s, N, S);
    Options o = opt ? *opt : Options();
    Summary local;
    Summary& S_ = sum ? *sum : local;
    solve(*P, o, x, (int)N, (int)S, S_);
    if (S_.termination_type != 2) {
        std::memcpy(q, x.q.data(), x.q.size() * sizeof(double));
        std::memcpy(t, x.t.data(), x.t.size() * sizeof(double));
        std::memcpy(s, x.s.data(), x.s.size() * sizeof(double));
    }
    return 0;
}

size_t orc_sizeof_summary() { return sizeof(Summary); }
size_t orc_sizeof_options() { return sizeof(Options); }

}  // extern "C"
