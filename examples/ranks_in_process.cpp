// ranks_in_process.cpp — the sharded solver driven from ONE C++ process through the C-ABI alone (include/pgo.h): a synthetic Manhattan graph is dealt out to N ranks with
// pgo_partition_edges (by place), every rank is a libpgo handle on its own thread joined to an in-process communicator (pgo_local_group_create / pgo_comm_init_local — on one
// GPU, or one handle per GPU of a node with peer access: set device_id per rank), and the N-rank solve is compared with the single-handle solve of the same graph.
// What it replaces in the reference: nothing one-to-one — ceres::Solve (src/PoseGraphSLAM.cpp:1903) runs on one CPU; this is how a maintainer would spread ONE large solve over
// the GPUs of a node without MPI / torch.distributed.  (One process per GPU over RCCL: pgo_comm_get_unique_id + pgo_comm_init, same calls otherwise.)
// Build: solve_keyframe_pose_graph_amd/_build.py::build_examples (g++, links libpgo.so + libpgo_graphgen.so).
//
//   ranks_in_process [n_poses = 20000] [ranks = 4] [n_gpus = 1] [mg_dist_min_rows = 1000]
// (mg_dist_min_rows: the library's default, 8 192, distributes multigrid levels of graphs from ~70 000 keyframes on; the smaller value lets this 20 000-keyframe demonstration run the
// distributed cycle and set-up — every rank its own rows of level 1 — instead of every level completely on every rank)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pgo.h"
#include "pgo_graphgen.h"

struct Graph {
    int64_t n = 0, n_odom = 0, n_loops = 0, n_reg = 0;
    std::vector<double> q, t, odom_T, odom_w, loop_T, loop_w, reg_T, reg_w;
    std::vector<int32_t> oc1, oc2, lc1, lc2, reg_node;
};

static int solve_ranks(const Graph& g, int world, int n_gpus, int dist_min_rows, std::vector<double>& q, std::vector<double>& t, std::vector<double>& s, pgo_summary* summary, pgo_sharding_stats* stats) {
    // which rank gets which residual block: recursive coordinate bisection of the keyframe positions, an edge follows its later endpoint
    std::vector<int32_t> part((size_t)g.n), orank((size_t)g.n_odom), lrank((size_t)g.n_loops);
    if (pgo_partition_edges(PGO_PARTITION_SPATIAL, world, g.n, g.t.data(), g.n_odom, g.oc1.data(), g.oc2.data(), g.n_loops, g.lc1.data(), g.lc2.data(), part.data(), orank.data(), lrank.data()) != PGO_OK) return 1;
    void* group = nullptr;
    if (pgo_local_group_create(world, &group) != PGO_OK) return 1;
    std::vector<std::vector<double>> Q((size_t)world, g.q), T((size_t)world, g.t), S((size_t)world, std::vector<double>((size_t)g.n_loops, 0.99));
    std::vector<pgo_summary> sums((size_t)world);
    std::vector<pgo_sharding_stats> sts((size_t)world);
    std::vector<int> rc((size_t)world, PGO_OK);
    auto rank_main = [&](int r) {
        pgo_options o;
        pgo_options_init(&o);
        o.device_id = n_gpus > 1 ? r % n_gpus : -1;
        o.mg_dist_min_rows = dist_min_rows;
        pgo_problem* p = nullptr;
        if ((rc[(size_t)r] = pgo_create(&p, &o)) != PGO_OK) { pgo_local_group_abort(group); return; }
        auto fail = [&](int code) { rc[(size_t)r] = code; std::fprintf(stderr, "rank %d: %s — %s\n", r, pgo_strerror(code), pgo_last_error(p)); pgo_local_group_abort(group); };
        // this rank's share of the edges (the reference's AddResidualBlock calls, src/PoseGraphSLAM.cpp:1550-1556,1629-1633)
        std::vector<int32_t> c1, c2, sw; std::vector<double> Tm, w;
        for (int64_t e = 0; e < g.n_odom; ++e) if (orank[(size_t)e] == r) { c1.push_back(g.oc1[(size_t)e]); c2.push_back(g.oc2[(size_t)e]); w.push_back(g.odom_w[(size_t)e]); Tm.insert(Tm.end(), g.odom_T.begin() + 16 * e, g.odom_T.begin() + 16 * (e + 1)); }
        int code = pgo_add_relpose_edges(p, (int64_t)c1.size(), c1.data(), c2.data(), Tm.data(), w.data());
        c1.clear(); c2.clear(); Tm.clear(); w.clear();
        for (int64_t e = 0; e < g.n_loops && code == PGO_OK; ++e) if (lrank[(size_t)e] == r) { c1.push_back(g.lc1[(size_t)e]); c2.push_back(g.lc2[(size_t)e]); sw.push_back((int32_t)e); w.push_back(g.loop_w[(size_t)e]); Tm.insert(Tm.end(), g.loop_T.begin() + 16 * e, g.loop_T.begin() + 16 * (e + 1)); }
        if (code == PGO_OK) code = pgo_add_switchable_edges(p, (int64_t)c1.size(), c1.data(), c2.data(), Tm.data(), w.data(), sw.data());
        std::vector<int32_t> rn; std::vector<double> rT, rw;
        for (int64_t k = 0; k < g.n_reg; ++k) if (part[(size_t)g.reg_node[(size_t)k]] == r) { rn.push_back(g.reg_node[(size_t)k]); rw.push_back(g.reg_w[(size_t)k]); rT.insert(rT.end(), g.reg_T.begin() + 16 * k, g.reg_T.begin() + 16 * (k + 1)); }
        if (code == PGO_OK) code = pgo_set_node_regularizers(p, (int64_t)rn.size(), rn.data(), rT.data(), rw.data());
        if (code == PGO_OK) code = pgo_comm_init_local(p, r, world, group);
        if (code == PGO_OK) code = pgo_solve(p, Q[(size_t)r].data(), T[(size_t)r].data(), S[(size_t)r].data(), g.n, g.n_loops, &sums[(size_t)r]);     // every rank gets the COMPLETE solution back
        if (code == PGO_OK) code = pgo_get_sharding_stats(p, &sts[(size_t)r]);
        if (code != PGO_OK) fail(code);
        pgo_comm_destroy(p);
        pgo_destroy(p);
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r);
    for (std::thread& x : th) x.join();
    pgo_local_group_destroy(group);
    for (int r = 0; r < world; ++r) if (rc[(size_t)r] != PGO_OK) return 1;
    for (int r = 1; r < world; ++r) if (T[(size_t)r] != T[0] || S[(size_t)r] != S[0]) { std::fprintf(stderr, "rank %d returned other bits than rank 0\n", r); return 1; }
    q = Q[0]; t = T[0]; s = S[0]; *summary = sums[0]; *stats = sts[0];
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n_poses = argc > 1 ? std::atoll(argv[1]) : 20000;
    const int world = argc > 2 ? std::atoi(argv[2]) : 4;
    const int n_gpus = argc > 3 ? std::atoi(argv[3]) : 1;
    const int dist_min_rows = argc > 4 ? std::atoi(argv[4]) : 1000;
    if (pgo_abi_version() != PGO_ABI_VERSION) { std::fprintf(stderr, "libpgo.so speaks ABI %d, this program was compiled against %d\n", pgo_abi_version(), PGO_ABI_VERSION); return 1; }
    pgo_gen_config c;
    pgo_gen_config_init(&c);
    c.n_poses = n_poses; c.n_loops = n_poses; c.odom_f_max = 2; c.seed = 3;
    pgo_gen_graph* gg = nullptr;
    if (pgo_gen_create(&c, &gg) != 0) { std::fprintf(stderr, "graph generator failed\n"); return 1; }
    Graph g;
    g.n = pgo_gen_num_poses(gg); g.n_odom = pgo_gen_num_odom(gg); g.n_loops = pgo_gen_num_loops(gg); g.n_reg = pgo_gen_num_regularizers(gg);
    g.q.resize((size_t)4 * g.n); g.t.resize((size_t)3 * g.n);
    g.oc1.resize((size_t)g.n_odom); g.oc2.resize((size_t)g.n_odom); g.odom_T.resize((size_t)16 * g.n_odom); g.odom_w.resize((size_t)g.n_odom);
    g.lc1.resize((size_t)g.n_loops); g.lc2.resize((size_t)g.n_loops); g.loop_T.resize((size_t)16 * g.n_loops); g.loop_w.resize((size_t)g.n_loops);
    g.reg_node.resize((size_t)g.n_reg); g.reg_T.resize((size_t)16 * g.n_reg); g.reg_w.resize((size_t)g.n_reg);
    pgo_gen_get_poses(gg, nullptr, nullptr, g.q.data(), g.t.data(), nullptr);
    pgo_gen_get_odom(gg, g.oc1.data(), g.oc2.data(), g.odom_T.data(), g.odom_w.data());
    pgo_gen_get_loops(gg, g.lc1.data(), g.lc2.data(), g.loop_T.data(), g.loop_w.data(), nullptr);
    pgo_gen_get_regularizers(gg, g.reg_node.data(), g.reg_T.data(), g.reg_w.data());
    pgo_gen_destroy(gg);

    // the single handle
    pgo_problem* p = nullptr;
    if (pgo_create(&p, nullptr) != PGO_OK) { std::fprintf(stderr, "no usable GPU: libpgo has no CPU path\n"); return 1; }
    std::vector<int32_t> sw((size_t)g.n_loops);
    for (int64_t e = 0; e < g.n_loops; ++e) sw[(size_t)e] = (int32_t)e;
    pgo_add_relpose_edges(p, g.n_odom, g.oc1.data(), g.oc2.data(), g.odom_T.data(), g.odom_w.data());
    pgo_add_switchable_edges(p, g.n_loops, g.lc1.data(), g.lc2.data(), g.loop_T.data(), g.loop_w.data(), sw.data());
    pgo_set_node_regularizers(p, g.n_reg, g.reg_node.data(), g.reg_T.data(), g.reg_w.data());
    std::vector<double> q1 = g.q, t1 = g.t, s1((size_t)g.n_loops, 0.99);
    pgo_summary one;
    if (pgo_solve(p, q1.data(), t1.data(), s1.data(), g.n, g.n_loops, &one) != PGO_OK) { std::fprintf(stderr, "single handle: %s\n", pgo_last_error(p)); return 1; }
    pgo_destroy(p);

    std::vector<double> qr, tr, sr;
    pgo_summary many; pgo_sharding_stats st;
    if (solve_ranks(g, world, n_gpus, dist_min_rows, qr, tr, sr, &many, &st) != 0) return 1;
    double dt = 0.0;
    for (size_t i = 0; i < tr.size(); ++i) dt = std::fmax(dt, std::fabs(tr[i] - t1[i]));
    bool same = one.num_iterations == many.num_iterations;
    for (int k = 0; k < one.num_logged && same; ++k) same = one.iterations[k].step_is_successful == many.iterations[k].step_is_successful;
    std::printf("%lld keyframes / %lld edges: single handle %d LM iterations, cost %.9e, %lld PCG iterations | %d ranks: %d LM iterations, cost %.9e, %lld PCG iterations, decisions %s, "
                "max position difference %.2e m | rank 0: %lld of %lld keyframes local, %lld shared; %d of %d multigrid levels distributed; %.0f B sent per multigrid iteration "
                "(the union all-reduce of earlier rounds: %.0f B), %d exchanges per iteration; multigrid set-up: %d level(s) formed by their rows' owners, this rank forms %lld of %lld blocks "
                "and sends %.0f B per set-up (the replicated set-up all-reduces %.0f B)\n",
                (long long)g.n, (long long)(g.n_odom + g.n_loops), one.num_iterations, one.final_cost, (long long)one.cg_iterations, world, many.num_iterations, many.final_cost,
                (long long)many.cg_iterations, same ? "equal" : "DIFFERENT", dt, (long long)st.keyframes_local, (long long)g.n, (long long)st.keyframes_shared, st.mg_levels_distributed, st.mg_levels,
                st.bytes_sent_per_mg_iteration, st.bytes_round5_per_mg_iteration, st.exchanges_per_mg_iteration, st.mg_setup_levels_own_rows, (long long)st.mg_setup_blocks_own,
                (long long)st.mg_setup_blocks_total, st.bytes_sent_per_mg_setup, st.bytes_allreduce_replicated_setup);
    return same && std::fabs(many.final_cost - one.final_cost) <= 1e-6 * one.final_cost ? 0 : 1;
}
