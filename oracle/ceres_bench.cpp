// ORACLE — TEST / MEASUREMENT INFRASTRUCTURE ONLY (never linked into the product).
//
// ceres_bench — the reference's CPU path with the REAL Ceres Solver: this repo's restatement of the reference's three cost functors
// (oracle/functors.hpp, statement by statement after reference src/CeresResidues.h:19-222) driven by real `ceres::AutoDiffCostFunction`,
// `ceres::EigenQuaternionParameterization` / `EigenQuaternionManifold` and `ceres::Solve` with the reference's options
// (src/PoseGraphSLAM.cpp:1268-1272: LEVENBERG_MARQUARDT, SPARSE_NORMAL_CHOLESKY, max_num_iterations = 10, everything else Ceres' defaults),
// on the same synthetic graph bench.py solves on the GPU (csrc/pgo_graphgen.cpp through libpgo_graphgen.so).  It is the only way to time genuine Ceres
// here: the reference's own sources need ROS / OpenCV on top and cannot be compiled.
//
// Built ONLY where <ceres/ceres.h> and Eigen3 exist (oracle/Makefile target `ceres_bench`; bench.py probes for the headers); this image and the GPU boxes
// have neither, so the file has never been compiled here — it is kept small and plain for that reason.
//
//   ./ceres_bench <n_poses> <n_loops> <odom_f_max> <seed> <max_iterations> <num_threads>     -> one JSON line on stdout
#include <ceres/ceres.h>
#include <ceres/version.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "functors.hpp"
#include "pgo_graphgen.h"

#if CERES_VERSION_MAJOR > 2 || (CERES_VERSION_MAJOR == 2 && CERES_VERSION_MINOR >= 2)
#define ORC_CERES_MANIFOLD 1      // LocalParameterization was removed in Ceres 2.2
#else
#define ORC_CERES_MANIFOLD 0
#endif

int main(int argc, char** argv) {
    if (argc < 7) { std::fprintf(stderr, "usage: %s n_poses n_loops odom_f_max seed max_iterations num_threads\n", argv[0]); return 2; }
    pgo_gen_config c;
    pgo_gen_config_init(&c);
    c.n_poses = std::atoll(argv[1]); c.n_loops = std::atoll(argv[2]); c.odom_f_max = std::atoi(argv[3]); c.seed = (uint64_t)std::atoll(argv[4]);
    const int max_it = std::atoi(argv[5]), threads = std::atoi(argv[6]);
    pgo_gen_graph* g = nullptr;
    if (pgo_gen_create(&c, &g) != 0) { std::fprintf(stderr, "graph generation failed\n"); return 1; }
    const int64_t N = pgo_gen_num_poses(g), Eo = pgo_gen_num_odom(g), El = pgo_gen_num_loops(g), Eg = pgo_gen_num_regularizers(g);
    std::vector<double> q((size_t)N * 4), t((size_t)N * 3), sw((size_t)El, 0.99);                 // switch init: reference src/PoseGraphSLAM.cpp:353
    pgo_gen_get_poses(g, nullptr, nullptr, q.data(), t.data(), nullptr);
    std::vector<int32_t> oc1(Eo), oc2(Eo), lc1(El), lc2(El), rn(Eg);
    std::vector<double> oT((size_t)Eo * 16), ow(Eo), lT((size_t)El * 16), lw(El), rT((size_t)Eg * 16), rw(Eg);
    pgo_gen_get_odom(g, oc1.data(), oc2.data(), oT.data(), ow.data());
    pgo_gen_get_loops(g, lc1.data(), lc2.data(), lT.data(), lw.data(), nullptr);
    pgo_gen_get_regularizers(g, rn.data(), rT.data(), rw.data());

    ceres::Problem problem;
#if ORC_CERES_MANIFOLD
    ceres::Manifold* quat_param = new ceres::EigenQuaternionManifold;
#else
    ceres::LocalParameterization* quat_param = new ceres::EigenQuaternionParameterization;       // reference src/PoseGraphSLAM.cpp:1276
#endif
    for (int64_t i = 0; i < N; ++i) {                                                             // (:1351-1353)
        problem.AddParameterBlock(&q[4 * i], 4);
#if ORC_CERES_MANIFOLD
        problem.SetManifold(&q[4 * i], quat_param);
#else
        problem.SetParameterization(&q[4 * i], quat_param);
#endif
        problem.AddParameterBlock(&t[3 * i], 3);
    }
    for (int64_t e = 0; e < El; ++e) problem.AddParameterBlock(&sw[e], 1);                         // (:1365)
    auto mat = [](const double* T16) { orc::Mat4d M; std::memcpy(M.d, T16, sizeof(M.d)); return M; };
    for (int64_t e = 0; e < Eo; ++e) {                                                            // odometry (:1629-1633), CeresResidues.h:72-79
        ceres::CostFunction* f = new ceres::AutoDiffCostFunction<orc::SixDOFError, 6, 4, 3, 4, 3>(new orc::SixDOFError(mat(&oT[16 * e]), ow[e]));
        problem.AddResidualBlock(f, nullptr, &q[4 * oc1[e]], &t[3 * oc1[e]], &q[4 * oc2[e]], &t[3 * oc2[e]]);
    }
    for (int64_t e = 0; e < El; ++e) {                                                            // loop closures (:1550-1556), CeresResidues.h:204-211
        ceres::CostFunction* f = new ceres::AutoDiffCostFunction<orc::SixDOFErrorWithSwitchingConstraints, 7, 4, 3, 4, 3, 1>(new orc::SixDOFErrorWithSwitchingConstraints(mat(&lT[16 * e]), lw[e]));
        problem.AddResidualBlock(f, nullptr, &q[4 * lc1[e]], &t[3 * lc1[e]], &q[4 * lc2[e]], &t[3 * lc2[e]], &sw[e]);
    }
    for (int64_t k = 0; k < Eg; ++k) {                                                            // regularisers (:1844-1848), CeresResidues.h:129-136
        ceres::CostFunction* f = new ceres::AutoDiffCostFunction<orc::NodePoseRegularization, 6, 4, 3>(new orc::NodePoseRegularization(mat(&rT[16 * k]), rw[k]));
        problem.AddResidualBlock(f, nullptr, &q[4 * rn[k]], &t[3 * rn[k]]);
    }
    ceres::Solver::Options opt;                                                                   // (:1268-1272)
    opt.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
    opt.minimizer_progress_to_stdout = false;
    opt.max_num_iterations = max_it;
    opt.num_threads = threads > 0 ? threads : 1;                                                  // the reference never sets it: Ceres' default 1
    ceres::Solver::Summary summary;
    const auto t0 = std::chrono::steady_clock::now();
    ceres::Solve(opt, &problem, &summary);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int iters = (int)summary.iterations.size() - 1;
    std::printf("{\"ceres_version\": \"%s\", \"poses\": %lld, \"edges\": %lld, \"threads\": %d, \"lm_iterations\": %d, \"successful_steps\": %d, \"seconds\": %.6f, "
                "\"seconds_linear_solver\": %.6f, \"seconds_jacobian\": %.6f, \"lm_iters_per_s\": %.6f, \"initial_cost\": %.12e, \"final_cost\": %.12e, \"termination\": %d}\n",
                CERES_VERSION_STRING, (long long)N, (long long)(Eo + El), opt.num_threads, iters, summary.num_successful_steps, wall, summary.linear_solver_time_in_seconds,
                summary.jacobian_evaluation_time_in_seconds, iters / summary.total_time_in_seconds, summary.initial_cost, summary.final_cost, (int)summary.termination_type);
    pgo_gen_destroy(g);
    return 0;
}
