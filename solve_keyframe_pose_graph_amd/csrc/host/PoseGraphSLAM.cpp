// PoseGraphSLAM.cpp — see PoseGraphSLAM.hpp.  Host code only; every numerical step of the solve happens inside libpgo (HIP).
#include "PoseGraphSLAM.hpp"
#include "GraphFormats.hpp"

#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../pgo_device_math.hpp"

namespace pgo_host {

// ---------------------------------------------------------------- Matrix4d
Matrix4d Matrix4d::Identity() { Matrix4d m; m.d.fill(0.0); m.d[0] = m.d[5] = m.d[10] = m.d[15] = 1.0; return m; }
Matrix4d Matrix4d::operator*(const Matrix4d& o) const {
    Matrix4d r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += (*this)(i, k) * o(k, j); r(i, j) = s; }
    return r;
}
// The reference calls Eigen's general Matrix4d::inverse() (src/PoseGraphSLAM.cpp:1463,1599,1770), which for a pose [A t; 0 1] is
// [A^-1, -A^-1 t] with A^-1 by cofactors — also for a not-quite-orthonormal A (VIO poses are only orthonormal to ~1e-7).  Same algebra
// as the K0 device path (pgo_device_math.hpp: vio_relative_pose), so the host and device paths of the trigger agree.
Matrix4d Matrix4d::inverse() const {
    const Matrix4d& M = *this;
    const double a00 = M(0, 0), a01 = M(0, 1), a02 = M(0, 2), a10 = M(1, 0), a11 = M(1, 1), a12 = M(1, 2), a20 = M(2, 0), a21 = M(2, 1), a22 = M(2, 2);
    const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const double idet = 1.0 / (a00 * c00 + a01 * c10 + a02 * c20);
    Matrix4d r = Identity();
    r(0, 0) = c00 * idet; r(0, 1) = c01 * idet; r(0, 2) = c02 * idet;
    r(1, 0) = c10 * idet; r(1, 1) = c11 * idet; r(1, 2) = c12 * idet;
    r(2, 0) = c20 * idet; r(2, 1) = c21 * idet; r(2, 2) = c22 * idet;
    for (int i = 0; i < 3; ++i) r(i, 3) = -(r(i, 0) * M(0, 3) + r(i, 1) * M(1, 3) + r(i, 2) * M(2, 3));
    return r;
}
void raw_xyzw_to_eigenmat(const double* quat, const double* t, Matrix4d& dst) {
    double R[9];
    pgo::quat_to_rot(quat[0], quat[1], quat[2], quat[3], R);     // Quaterniond(w,x,y,z).toRotationMatrix()
    dst = Matrix4d::Identity();
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dst(r, c) = R[r * 3 + c];
    dst(0, 3) = t[0]; dst(1, 3) = t[1]; dst(2, 3) = t[2];
}
void eigenmat_to_raw_xyzw(const Matrix4d& T, double* quat, double* t) {
    double R[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[r * 3 + c] = T(r, c);
    pgo::eigen_matrix_to_quat(R, quat);                          // Quaterniond q(T.topLeftCorner<3,3>())
    t[0] = T(0, 3); t[1] = T(1, 3); t[2] = T(2, 3);
}
double yaw_degrees(const Matrix4d& T) { return std::atan2(T(1, 0), T(0, 0)) / M_PI * 180.0; }   // R2ypr(...)(0)

// ---------------------------------------------------------------- PoseGraphSLAM
PoseGraphSLAM::PoseGraphSLAM(GraphSource* _manager, const pgo_options* options) : manager(_manager) {
    std::memset(&summary_, 0, sizeof(summary_));
    last_rc_ = pgo_create(&problem_, options);     // replaces the persistent ceres::Problem; fails without a GPU
    if (last_rc_ != PGO_OK) problem_ = nullptr;
}
PoseGraphSLAM::~PoseGraphSLAM() { if (problem_) pgo_destroy(problem_); }

int PoseGraphSLAM::nNodes() const { std::lock_guard<std::mutex> lk(mutex_opt_vars); return (int)(_opt_t_.size() / 3); }
int PoseGraphSLAM::n_opt_variables() const { return nNodes(); }
int PoseGraphSLAM::n_opt_switch() const { std::lock_guard<std::mutex> lk(mutex_opt_vars); return (int)_opt_switch_.size(); }
bool PoseGraphSLAM::nodePoseExists(int i) const { std::lock_guard<std::mutex> lk(mutex_opt_vars); return i >= 0 && i < (int)(_opt_t_.size() / 3); }
int PoseGraphSLAM::solvedUntil() const { std::lock_guard<std::mutex> lk(mutex_opt_vars); return solved_until; }
const Matrix4d PoseGraphSLAM::getNodePose(int i) const {
    std::lock_guard<std::mutex> lk(mutex_opt_vars);
    Matrix4d T = Matrix4d::Identity();
    if (i >= 0 && i < (int)(_opt_t_.size() / 3)) raw_xyzw_to_eigenmat(&_opt_quat_[4 * i], &_opt_t_[3 * i], T);
    return T;
}
void PoseGraphSLAM::getAllNodePose(std::vector<Matrix4d>& v) const {
    v.clear();
    const int n = nNodes();
    for (int i = 0; i < n; ++i) v.push_back(getNodePose(i));
}
double PoseGraphSLAM::get_loopedge_switching_variable_val(int i) const {
    std::lock_guard<std::mutex> lk(mutex_opt_vars);
    return (i >= 0 && i < (int)_opt_switch_.size()) ? _opt_switch_[i] : 0.0;
}
void PoseGraphSLAM::allocate_and_append_new_opt_variable_withpose(const Matrix4d& pose) {
    double q[4], t[3];
    eigenmat_to_raw_xyzw(pose, q, t);
    std::lock_guard<std::mutex> lk(mutex_opt_vars);
    _opt_quat_.insert(_opt_quat_.end(), q, q + 4);
    _opt_t_.insert(_opt_t_.end(), t, t + 3);
}
bool PoseGraphSLAM::update_opt_variable_with(int i, const Matrix4d& pose) {
    double q[4], t[3];
    eigenmat_to_raw_xyzw(pose, q, t);
    std::lock_guard<std::mutex> lk(mutex_opt_vars);
    if (i < 0 || i >= (int)(_opt_t_.size() / 3)) return false;
    std::copy(q, q + 4, &_opt_quat_[4 * i]);
    std::copy(t, t + 3, &_opt_t_[3 * i]);
    return true;
}
void PoseGraphSLAM::allocate_and_append_new_edge_switch_var() {
    std::lock_guard<std::mutex> lk(mutex_opt_vars);
    _opt_switch_.push_back(0.99);   // reference src/PoseGraphSLAM.cpp:353
}

bool PoseGraphSLAM::load_state(bool optimization_variable_as_constants) {
    if (!problem_) return false;
    if (!optimization_variable_as_constants) return true;          // the solver thread picks such keyframes up by itself (:60-63)
    const int node_len = manager->getNodeLen();
    std::vector<int32_t> constant;
    for (int yp = n_opt_variables(); yp < node_len; ++yp) {
        const int world = manager->which_world_is_this_node(yp);
        const int set_id = manager->find_setID_of_world_i(world);
        Matrix4d ws_T_w = Matrix4d::Identity();
        if (world >= 0 && world != set_id) {
            if (!manager->is_exist(set_id, world)) { last_rc_ = PGO_ERR_STATE; return false; }   // the reference exit(1)s here (:96-101)
            ws_T_w = manager->getPoseBetweenWorlds(set_id, world);
        }
        allocate_and_append_new_opt_variable_withpose(ws_T_w * manager->getNodePose(yp));            // (:103-114)
        constant.push_back(yp);
    }
    if (!constant.empty()) last_rc_ = pgo_set_nodes_constant(problem_, (int64_t)constant.size(), constant.data());   // (:143-144)
    {
        std::lock_guard<std::mutex> lk(mutex_opt_vars);
        solved_until = node_len - 1;                                                                  // (:158)
    }
    prev_node_len = node_len;
    odom_until_ = std::max(odom_until_, node_len);
    return last_rc_ == PGO_OK;
}

bool PoseGraphSLAM::reinit_ceres_problem_onnewloopedge_optimize6DOF_once() {
    if (!problem_) return false;
    const int node_len = manager->getNodeLen();
    const int loopedge_len = manager->getEdgeLen();
    if (prev_loopedge_len == loopedge_len) { status_ = 0; return false; }       // no new loop edge: sleep again (:1306-1312)
    if (manager->curr_kidnap_status()) { status_ = 0; return false; }           // kidnapped: sleep (:1314-1319)
    status_ = 1;
    last_rc_ = PGO_OK;
    // A failed pgo_* call ends the wake-up at once: nothing after it runs, and the progress markers only move past what libpgo has
    // accepted (loop edges: prev_loopedge_len; odometry residues: odom_until_), so the next wake-up neither repeats nor loses blocks.
    auto failed = [this]() { status_ = 0; return false; };
    // PGO_HOST_TIMING=1: wall time of the wake-up's steps on stderr (where a trigger's time goes besides pgo_solve)
    static const bool timing = []() { const char* e = std::getenv("PGO_HOST_TIMING"); return e && e[0] == '1'; }();
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto mark = [&](const char* what) { if (timing) { const double t = now(); std::fprintf(stderr, "[pgo host] %-28s %8.3f ms\n", what, (t - t_mark) * 1e3); t_mark = t; } };

    // -0- new optimisation variables (:1340-1367)
    for (int yp = n_opt_variables(); yp < node_len; ++yp) allocate_and_append_new_opt_variable_withpose(Matrix4d::Identity());
    for (int yp = n_opt_switch(); yp < loopedge_len; ++yp) allocate_and_append_new_edge_switch_var();

    // -1/-2- loop edges, intra- and inter-world (:1381-1559)
    std::vector<int32_t> c1, c2, sw;
    std::vector<double> T, w;
    std::vector<AddedEdge> pending;
    for (int e = prev_loopedge_len; e < loopedge_len; ++e) {
        const Matrix4d bTa = manager->getEdgePose(e);
        const double weight = manager->getEdgeWeight(e);
        const std::pair<int, int> paur = manager->getEdgeIdxInfo(e);
        const int a = paur.first, b = paur.second;
        if (a == b || a < 0 || b < 0 || a >= node_len || b >= node_len) continue;   // not a residual block Ceres could hold (one parameter block twice): dropped alone, not with its batch
        const int a_world = manager->which_world_is_this_node(a), b_world = manager->which_world_is_this_node(b);
        if (a_world < 0 || b_world < 0) continue;                                // an endpoint lies in a dead zone (:1400-1401)
        if (a_world != b_world && !manager->is_exist(b_world, a_world)) {
            // relative pose between the two worlds from ODOMETRY poses at first contact (:1459-1464)
            const Matrix4d wb_T_wa = (manager->getNodePose(b) * bTa) * manager->getNodePose(a).inverse();
            std::map<int, int> before, after;
            manager->getWorld2SetIDMap(before);
            manager->setPoseBetweenWorlds(b_world, a_world, wb_T_wa);           // the only place two sets can merge (:1489-1490)
            manager->getWorld2SetIDMap(after);
            changes_to_setid_on_set_union.clear();
            for (const auto& kv : before) {
                const auto it = after.find(kv.first);
                if (it != after.end() && it->second != kv.second) changes_to_setid_on_set_union[kv.first] = std::make_tuple(kv.second, it->second);
            }
        }
        // SixDOFErrorWithSwitchingConstraints(bTa, weight) on (q_b,t_b, q_a,t_a, s_e)  (:1550-1556)
        c1.push_back(b); c2.push_back(a); sw.push_back(e); w.push_back(weight);
        T.insert(T.end(), bTa.d.begin(), bTa.d.end());
        pending.push_back({b, a, weight, e});
    }
    if (!c1.empty()) {
        last_rc_ = pgo_add_switchable_edges(problem_, (int64_t)c1.size(), c1.data(), c2.data(), T.data(), w.data(), sw.data());
        if (last_rc_ != PGO_OK) return failed();
    }
    added_edges_.insert(added_edges_.end(), pending.begin(), pending.end());
    prev_loopedge_len = loopedge_len;
    mark("-0/1/2- variables, loop edges");

    // -3- odometry residues u <-> u-f, f = 1..5 (:1570-1639)
    const int su = solvedUntil();
    const int u_first = std::max(su + 1, odom_until_);     // = su + 1 unless an earlier wake-up failed after adding its odometry residues
    if (device_graph_construction_) {
        // K0 on the device: only the new raw VIO poses travel; measurement, quaternion and yaw weight are computed there
        int64_t resident = 0;
        last_rc_ = pgo_num_vio_poses(problem_, &resident);
        if (last_rc_ != PGO_OK) return failed();
        if (resident < node_len) {
            std::vector<double> fresh;
            fresh.reserve((size_t)(node_len - resident) * 16);
            for (int u = (int)resident; u < node_len; ++u) { const Matrix4d M = manager->getNodePose(u); fresh.insert(fresh.end(), M.d.begin(), M.d.end()); }
            last_rc_ = pgo_set_vio_poses(problem_, resident, node_len - resident, fresh.data());
            if (last_rc_ != PGO_OK) return failed();
        }
        mark("   -3- pgo_set_vio_poses");
        std::vector<int32_t> set_id((size_t)node_len, 0);
        for (int u = std::max(0, u_first - 5); u < node_len; ++u) set_id[u] = manager->find_setID_of_world_i(manager->which_world_is_this_node(u));
        int64_t before = 0, added = 0;
        if ((last_rc_ = pgo_num_relpose_edges(problem_, &before)) != PGO_OK) return failed();
        if (u_first < node_len && (last_rc_ = pgo_add_odometry_edges_from_vio(problem_, set_id.data(), u_first, node_len, 5, 1, &added)) != PGO_OK) return failed();
        odom_until_ = std::max(odom_until_, node_len);
        mark("   -3- pgo_add_odometry_edges");
        if (added > 0) {
            std::vector<int32_t> a1((size_t)added), a2((size_t)added);
            std::vector<double> rec((size_t)added * 8);
            if ((last_rc_ = pgo_get_relpose_edge_records(problem_, before, added, a1.data(), a2.data(), rec.data())) != PGO_OK) return failed();
            for (int64_t k = 0; k < added; ++k) added_edges_.push_back({a1[k], a2[k], rec[k * 8 + 7], -1});
        }
    } else {
        c1.clear(); c2.clear(); T.clear(); w.clear(); pending.clear();
        for (int u = u_first; u < node_len; ++u) {
            const int set_u = manager->find_setID_of_world_i(manager->which_world_is_this_node(u));
            for (int f = 1; f < 6; ++f) {
                const int world_umf = (u - f >= 0) ? manager->which_world_is_this_node(u - f) : -1;
                const int set_umf = manager->find_setID_of_world_i(world_umf);
                if (set_u < 0 || set_umf < 0) continue;                              // dead zone (:1583-1586)
                if (u - f < 0) continue;                                             // (:1588-1591)
                const Matrix4d u_M_umf = manager->getNodePose(u).inverse() * manager->getNodePose(u - f);      // (:1597-1599)
                const double yaw = yaw_degrees(u_M_umf);
                const double odom_edge_weight = std::pow(0.9, f) * std::exp(-yaw * yaw / 6.0);                // (:1603-1606)
                c1.push_back(u); c2.push_back(u - f); w.push_back(odom_edge_weight);
                T.insert(T.end(), u_M_umf.d.begin(), u_M_umf.d.end());
                pending.push_back({u, u - f, odom_edge_weight, -1});
            }
        }
        if (!c1.empty() && (last_rc_ = pgo_add_relpose_edges(problem_, (int64_t)c1.size(), c1.data(), c2.data(), T.data(), w.data())) != PGO_OK) return failed();
        added_edges_.insert(added_edges_.end(), pending.begin(), pending.end());
        odom_until_ = std::max(odom_until_, node_len);
    }

    mark("-3- odometry residues");
    // -4- initial guesses (:1649-1793)
    {
        const int s_until = su;
        int s_world = manager->which_world_is_this_node(s_until);
        if (s_world < 0) s_world = -s_world - 1;
        // device path: pose_u = left[sel] * w_M_u for the keyframes that take their guess from the VIO pose; table entry 0 = chaining
        // from the last solved pose (w_T_last * w_M_last^-1, :1770-1775), 1 = identity (:1756-1761), 2+k = k-th distinct wset_T_w (:1777-1780)
        std::vector<double> left;
        std::vector<int32_t> sel((size_t)node_len, -1);
        std::map<int, int> left_of_world;
        if (device_graph_construction_ && node_len > 0) {
            const Matrix4d I = Matrix4d::Identity();
            left.insert(left.end(), I.d.begin(), I.d.end());                     // entry 0 is filled in after the loop
            left.insert(left.end(), I.d.begin(), I.d.end());
        }
        auto guess_from_vio = [&](int u, int which, const Matrix4d& L, int world_u) {
            if (!device_graph_construction_) { update_opt_variable_with(u, L * manager->getNodePose(u)); return; }
            if (which == 2) {
                auto it = left_of_world.find(world_u);
                if (it == left_of_world.end()) { it = left_of_world.emplace(world_u, (int)(left.size() / 16)).first; left.insert(left.end(), L.d.begin(), L.d.end()); }
                which = it->second;
            }
            sel[u] = which;
        };
        for (int u = 0; u < node_len; ++u) {
            const int world_u = manager->which_world_is_this_node(u);
            const int set_u = manager->find_setID_of_world_i(world_u);
            if (set_u < 0) continue;                                             // kidnapped nodes (:1665-1668)
            Matrix4d wset_T_w = Matrix4d::Identity();
            if (set_u != world_u) {
                if (!manager->is_exist(set_u, world_u)) { last_rc_ = PGO_ERR_STATE; status_ = 0; return false; }   // the reference exit(3)s here
                wset_T_w = manager->getPoseBetweenWorlds(set_u, world_u);
            }
            const bool before = u <= s_until;
            const bool in_change_set = changes_to_setid_on_set_union.count(world_u) > 0;
            if (in_change_set && before) {
                if (set_u == s_world) { last_rc_ = PGO_ERR_STATE; status_ = 0; return false; }                      // the reference exit(8)s here
                const int old_setid = std::get<0>(changes_to_setid_on_set_union[world_u]);
                const int new_setid = std::get<1>(changes_to_setid_on_set_union[world_u]);
                // re-expresses an already OPTIMISED pose in the merged set's frame: host (rare, world merges only)
                update_opt_variable_with(u, manager->getPoseBetweenWorlds(new_setid, old_setid) * this->getNodePose(u));
            } else if (!before) {
                // both the in-change-set and the ordinary branch chain from the last solved pose inside its world, or map the
                // odometry pose into the set's frame otherwise (:1727-1753, :1767-1786)
                if (s_world == world_u) {
                    if (device_graph_construction_) sel[u] = 0;
                    else {
                        const Matrix4d last_M_u = manager->getNodePose(s_until).inverse() * manager->getNodePose(u);
                        update_opt_variable_with(u, this->getNodePose(s_until) * last_M_u);
                    }
                } else {
                    guess_from_vio(u, 2, wset_T_w, world_u);
                }
            } else if (s_until == 0) {
                guess_from_vio(u, 1, Matrix4d::Identity(), world_u);            // very first trigger (:1756-1761)
            }
        }
        if (device_graph_construction_ && node_len > 0) {
            // The chain anchor w_T_last must be the pose the sequential host loop would read at this point.  Keyframes <= s_until that
            // were re-expressed above are final already; on the very first trigger (s_until == 0) keyframe 0's own guess w_M_0 is still
            // deferred to the kernel (sel = 1), so it is resolved here — through the same (xyzw, t) round trip update_opt_variable_with
            // + getNodePose perform — instead of reading the Identity the variable was allocated with.
            Matrix4d w_T_last = this->getNodePose(s_until);
            if (s_until >= 0 && s_until < node_len && sel[s_until] >= 0) {
                Matrix4d L;
                std::copy(left.begin() + (size_t)sel[s_until] * 16, left.begin() + (size_t)sel[s_until] * 16 + 16, L.d.begin());
                double q[4], t3[3];
                eigenmat_to_raw_xyzw(L * manager->getNodePose(s_until), q, t3);
                raw_xyzw_to_eigenmat(q, t3, w_T_last);
            }
            const Matrix4d chain = w_T_last * manager->getNodePose(s_until).inverse();
            std::copy(chain.d.begin(), chain.d.end(), left.begin());
            std::lock_guard<std::mutex> lk(mutex_opt_vars);
            last_rc_ = pgo_initial_guess_from_vio(problem_, (int64_t)(left.size() / 16), left.data(), sel.data(), 0, node_len, _opt_quat_.data(), _opt_t_.data());
        }
        if (last_rc_ != PGO_OK) return failed();
    }

    mark("-4- initial guesses");
    // -5- node regularisation replaces the previous set (:1803-1877)
    regs_.clear();
    for (int ww = 0; ww < manager->n_worlds(); ++ww) {
        const int ww_setid = manager->find_setID_of_world_i(ww);
        const int ww_start = manager->nodeidx_of_world_i_started(ww), ww_end = manager->nodeidx_of_world_i_ended(ww);
        if (ww_start < 0 || ww_start >= node_len) continue;
        if (ww_setid >= 0 && ww_setid == ww) {
            const double regularization_weight = std::max(1.1, std::log(1.0 + ww_end - ww_start) / 2.0);            // (:1839)
            regs_.push_back({ww_start, regularization_weight, this->getNodePose(ww_start)});                          // (:1844-1848)
        }
    }
    {
        std::vector<int32_t> rn; std::vector<double> rw, rT;
        for (const AddedRegularizer& r : regs_) { rn.push_back(r.node); rw.push_back(r.weight); rT.insert(rT.end(), r.target.d.begin(), r.target.d.end()); }
        if ((last_rc_ = pgo_set_node_regularizers(problem_, (int64_t)rn.size(), rn.data(), rT.data(), rw.data())) != PGO_OK) return failed();
    }
    changes_to_setid_on_set_union.clear();                                       // (:1882)

    mark("-5- regularisers");
    // -6- solve WITHOUT holding the lock; single write-back at the end (:1891-1910)
    status_ = 2;
    std::vector<double> q, t, s;
    {
        std::lock_guard<std::mutex> lk(mutex_opt_vars);
        q = _opt_quat_; t = _opt_t_; s = _opt_switch_;
    }
    init_quat_ = q; init_t_ = t;
    last_rc_ = pgo_solve(problem_, q.data(), t.data(), s.empty() ? nullptr : s.data(), (int64_t)(t.size() / 3), (int64_t)s.size(), &summary_);
    mark("-6- pgo_solve");
    if (last_rc_ != PGO_OK) return failed();     // a library error (not a Ceres-style FAILURE): no write-back, solved_until stays
    {
        std::lock_guard<std::mutex> lk(mutex_opt_vars);
        if (summary_.termination_type != PGO_FAILURE) { _opt_quat_ = q; _opt_t_ = t; _opt_switch_ = s; }
        solved_until = node_len - 1;                                             // regardless of convergence (:1906-1910)
    }
    status_ = 0;
    prev_node_len = node_len;
    return true;
}

// ---------------------------------------------------------------- saveAsJSON (reference src/PoseGraphSLAM.cpp:1111-1207)
static std::string csv_matrix(const Matrix4d& M) { return matrix4d_to_csv(M); }            // shared with log_posegraph.json (GraphFormats.cpp)
static std::string prettyprintMatrix4d(const Matrix4d& M) { return prettyprint_matrix4d(M); }
bool PoseGraphSLAM::saveAsJSON(const std::string& base_path) const {
    FILE* f = std::fopen((base_path + "/log_optimized_poses.json").c_str(), "w");
    if (!f) return false;
    const int n = nNodes();
    std::fprintf(f, "{\n    \"meta_data\": {\"nNodes\": %d},\n    \"PoseGraphSLAM_nodes\": [", n);
    for (int i = 0; i < n; ++i) {
        const Matrix4d opt = getNodePose(i), odom = manager->getNodePose(i);
        std::fprintf(f, "%s\n        {\"node_i\": %d, \"wTc_opt\": \"%s\", \"wTc_opt_prettyprint\": \"%s\", \"w_T_c_odom\": \"%s\", \"w_T_c_odom_prettyprint\": \"%s\"}",
                     i ? "," : "", i, csv_matrix(opt).c_str(), prettyprintMatrix4d(opt).c_str(), csv_matrix(odom).c_str(), prettyprintMatrix4d(odom).c_str());
    }
    std::fprintf(f, "\n    ],\n    \"PoseGraphSLAM_loopedgeinfo\": [");
    const int ne = manager->getEdgeLen();
    for (int e = 0; e < ne; ++e) {
        const std::pair<int, int> ab = manager->getEdgeIdxInfo(e);
        const int a = ab.first, b = ab.second;
        std::fprintf(f, "%s\n        {\"getEdge_i\": %d, \"a\": %d, \"b\": %d, \"world_of_a\": %d, \"world_of_b\": %d, \"weight\": %.17g, \"getEdgePose\": \"%s\"",
                     e ? "," : "", e, a, b, manager->which_world_is_this_node(a), manager->which_world_is_this_node(b), manager->getEdgeWeight(e), prettyprintMatrix4d(manager->getEdgePose(e)).c_str());
        if (a < n && b < n) std::fprintf(f, ", \"getEdgePose_after_opt\": \"%s\"", prettyprintMatrix4d(getNodePose(b).inverse() * getNodePose(a)).c_str());
        if (e < n_opt_switch()) std::fprintf(f, ", \"switching_var_after_opt\": %.17g", get_loopedge_switching_variable_val(e));
        std::fprintf(f, "}");
    }
    std::fprintf(f, "\n    ]\n}\n");
    std::fclose(f);
    return true;
}

// ---------------------------------------------------------------- VectorGraphSource
void VectorGraphSource::ensure_world(int w) const {
    while ((int)set_of_.size() <= w) { set_of_.push_back((int)set_of_.size()); set_T_world_.push_back(Matrix4d::Identity()); }
}
int VectorGraphSource::n_worlds() const { return (int)world_first_.size(); }
int VectorGraphSource::nodeidx_of_world_i_started(int w) const { return (w >= 0 && (size_t)w < world_first_.size()) ? world_first_[w] : -1; }
int VectorGraphSource::nodeidx_of_world_i_ended(int w) const { return (w >= 0 && (size_t)w < world_last_.size()) ? world_last_[w] : -1; }
int VectorGraphSource::find_setID_of_world_i(int w) const { if (w < 0) return w; ensure_world(w); return set_of_[w]; }
bool VectorGraphSource::is_exist(int m, int n) const { if (m < 0 || n < 0) return false; ensure_world(std::max(m, n)); return set_of_[m] == set_of_[n]; }
Matrix4d VectorGraphSource::getPoseBetweenWorlds(int m, int n) const { ensure_world(std::max(m, n)); return set_T_world_[m].inverse() * set_T_world_[n]; }
void VectorGraphSource::setPoseBetweenWorlds(int m, int n, const Matrix4d& m_T_n) {
    ensure_world(std::max(m, n));
    if (set_of_[m] == set_of_[n]) return;
    // merge the set of n into the set of m (or the other way round) keeping the smaller set id as the root frame
    const int sm = set_of_[m], sn = set_of_[n];
    if (sm < sn) {
        const Matrix4d sm_T_sn = set_T_world_[m] * m_T_n * set_T_world_[n].inverse();
        for (size_t w = 0; w < set_of_.size(); ++w) if (set_of_[w] == sn) { set_T_world_[w] = sm_T_sn * set_T_world_[w]; set_of_[w] = sm; }
    } else {
        const Matrix4d sn_T_sm = set_T_world_[n] * m_T_n.inverse() * set_T_world_[m].inverse();
        for (size_t w = 0; w < set_of_.size(); ++w) if (set_of_[w] == sm) { set_T_world_[w] = sn_T_sm * set_T_world_[w]; set_of_[w] = sn; }
    }
}
void VectorGraphSource::getWorld2SetIDMap(std::map<int, int>& out) const {
    out.clear();
    const int n = n_worlds();
    if (n > 0) ensure_world(n - 1);
    for (int w = 0; w < n; ++w) out[w] = set_of_[w];
}

}  // namespace pgo_host

// ================================================================================================
// thin C wrapper so the parity tests (pytest/ctypes) can drive the C++ host side
// ================================================================================================
using namespace pgo_host;
struct pgo_host_session { VectorGraphSource src; PoseGraphSLAM* slam; std::vector<Matrix4d> loaded_w_T_c; };

extern "C" {
pgo_host_session* pgo_host_create(const pgo_options* opt) {
    pgo_host_session* s = new pgo_host_session();
    s->slam = new PoseGraphSLAM(&s->src, opt);
    if (!s->slam->ok()) { delete s->slam; delete s; return nullptr; }
    return s;
}
// A session that only holds the data source (no solver, no GPU): file-format work
pgo_host_session* pgo_host_create_source_only() { pgo_host_session* s = new pgo_host_session(); s->slam = nullptr; return s; }
// Attaches the solver to a source-only session (e.g. after pgo_host_load_posegraph_json); 1 on success
int pgo_host_attach_solver(pgo_host_session* s, const pgo_options* opt) {
    if (s->slam) return 1;
    s->slam = new PoseGraphSLAM(&s->src, opt);
    if (!s->slam->ok()) { delete s->slam; s->slam = nullptr; return 0; }
    return 1;
}
int pgo_host_source_n_nodes(pgo_host_session* s) { return s->src.getNodeLen(); }
int pgo_host_source_n_edges(pgo_host_session* s) { return s->src.getEdgeLen(); }
void pgo_host_source_get_node(pgo_host_session* s, int i, int* world, double* stamp, double* T16) {
    *world = s->src.which_world_is_this_node(i); *stamp = s->src.getNodeTimestamp(i);
    const Matrix4d T = s->src.getNodePose(i); std::copy(T.d.begin(), T.d.end(), T16);
}
void pgo_host_source_get_edge(pgo_host_session* s, int e, int* a, int* b, double* weight, double* bTa16, char* description, int description_cap) {
    const std::pair<int, int> p = s->src.getEdgeIdxInfo(e);
    *a = p.first; *b = p.second; *weight = s->src.getEdgeWeight(e);
    const Matrix4d T = s->src.getEdgePose(e); std::copy(T.d.begin(), T.d.end(), bTa16);
    if (description && description_cap > 0) { std::strncpy(description, s->src.getEdgeDescriptionString(e).c_str(), (size_t)description_cap - 1); description[description_cap - 1] = 0; }
}
void pgo_host_add_node_stamped(pgo_host_session* s, int world, const double* T16, double stamp) { Matrix4d T; std::copy(T16, T16 + 16, T.d.begin()); s->src.add_node(world, T, stamp); }
void pgo_host_add_loop_edge_described(pgo_host_session* s, int a, int b, const double* bTa16, double w, const char* description) {
    Matrix4d T; std::copy(bTa16, bTa16 + 16, T.d.begin()); s->src.add_loop_edge(a, b, T, w, description ? description : "");
}
// log_posegraph.json (NodeDataManager::saveAsJSON / loadFromJSON, reference src/NodeDataManager.cpp:503-754)
int pgo_host_save_posegraph_json(pgo_host_session* s, const char* base_path) { return save_posegraph_json(s->src, base_path) ? 1 : 0; }
int pgo_host_load_posegraph_json(pgo_host_session* s, const char* base_path, const uint8_t* edge_mask, int n_mask, char* err, int err_cap) {
    if (s->slam) { if (err && err_cap > 0) std::snprintf(err, (size_t)err_cap, "load into a source-only session, then attach the solver"); return 0; }
    std::vector<bool> mask;
    for (int i = 0; i < n_mask; ++i) mask.push_back(edge_mask[i] != 0);
    std::string e;
    const bool ok = load_posegraph_json(s->src, base_path, mask, &e);
    if (!ok && err && err_cap > 0) std::snprintf(err, (size_t)err_cap, "%s", e.c_str());
    return ok ? 1 : 0;
}
// solved_posegraph.json (Composer::saveStateToDisk / loadStateFromDisk, reference src/Composer.cpp:952-1177): corrected poses = the
// optimised ones when a solver is attached and has them, else the odometry poses
int pgo_host_save_solved_posegraph_json(pgo_host_session* s, const char* base_path) {
    std::vector<Matrix4d> poses;
    for (int i = 0; i < s->src.getNodeLen(); ++i) poses.push_back((s->slam && s->slam->nodePoseExists(i)) ? s->slam->getNodePose(i) : s->src.getNodePose(i));
    return save_solved_posegraph_json(s->src, poses, base_path) ? 1 : 0;
}
int pgo_host_load_solved_posegraph_json(pgo_host_session* s, const char* base_path, char* err, int err_cap) {
    if (s->slam) { if (err && err_cap > 0) std::snprintf(err, (size_t)err_cap, "load into a source-only session, then attach the solver"); return 0; }
    std::string e;
    const bool ok = load_solved_posegraph_json(s->src, s->loaded_w_T_c, base_path, &e);
    if (!ok && err && err_cap > 0) std::snprintf(err, (size_t)err_cap, "%s", e.c_str());
    return ok ? 1 : 0;
}
int pgo_host_n_loaded_poses(pgo_host_session* s) { return (int)s->loaded_w_T_c.size(); }
void pgo_host_get_loaded_pose(pgo_host_session* s, int i, double* T16) { std::copy(s->loaded_w_T_c[i].d.begin(), s->loaded_w_T_c[i].d.end(), T16); }
int pgo_host_source_set_id_of_world(pgo_host_session* s, int w) { return s->src.find_setID_of_world_i(w); }
void pgo_host_source_pose_between_worlds(pgo_host_session* s, int m, int n, double* T16) { const Matrix4d T = s->src.getPoseBetweenWorlds(m, n); std::copy(T.d.begin(), T.d.end(), T16); }
void pgo_host_source_merge_worlds(pgo_host_session* s, int m, int n, const double* m_T_n16) { Matrix4d T; std::copy(m_T_n16, m_T_n16 + 16, T.d.begin()); s->src.setPoseBetweenWorlds(m, n, T); }
// g2o export: keyframe poses = optimised when `optimized` and a solver is attached, else the VIO poses; edges = loop edges (b -> a, b_T_a,
// unit weight: the switchable functor ignores its weight, CeresResidues.h:198) followed by the odometry edges of the reference policy
// (f = 1..f_max, weight 0.9^f exp(-yaw^2/6)) for pairs whose endpoints are in live worlds
int pgo_host_export_g2o(pgo_host_session* s, const char* path, int optimized, int f_max) {
    const int n = s->src.getNodeLen();
    std::vector<Matrix4d> poses;
    for (int i = 0; i < n; ++i) poses.push_back((optimized && s->slam && s->slam->nodePoseExists(i)) ? s->slam->getNodePose(i) : s->src.getNodePose(i));
    std::vector<G2oEdge> edges;
    for (int e = 0; e < s->src.getEdgeLen(); ++e) { const std::pair<int, int> p = s->src.getEdgeIdxInfo(e); edges.push_back({p.second, p.first, s->src.getEdgePose(e), 1.0}); }
    for (int u = 0; u < n; ++u)
        for (int f = 1; f <= f_max; ++f) {
            if (u - f < 0 || s->src.find_setID_of_world_i(s->src.which_world_is_this_node(u)) < 0 || s->src.find_setID_of_world_i(s->src.which_world_is_this_node(u - f)) < 0) continue;
            const Matrix4d M = s->src.getNodePose(u).inverse() * s->src.getNodePose(u - f);
            const double yaw = yaw_degrees(M);
            edges.push_back({u, u - f, M, std::pow(0.9, f) * std::exp(-yaw * yaw / 6.0)});
        }
    return export_g2o(path, poses, edges) ? 1 : 0;
}
void pgo_host_destroy(pgo_host_session* s) { if (s) { delete s->slam; delete s; } }   // slam may be null (source-only)
void pgo_host_add_node(pgo_host_session* s, int world, const double* T16) { Matrix4d T; std::copy(T16, T16 + 16, T.d.begin()); s->src.add_node(world, T); }
void pgo_host_add_loop_edge(pgo_host_session* s, int a, int b, const double* bTa16, double w) { Matrix4d T; std::copy(bTa16, bTa16 + 16, T.d.begin()); s->src.add_loop_edge(a, b, T, w); }
void pgo_host_set_kidnapped(pgo_host_session* s, int k) { s->src.set_kidnapped(k != 0); }
void pgo_host_set_device_graph_construction(pgo_host_session* s, int on) { s->slam->set_device_graph_construction(on != 0); }
int pgo_host_load_state(pgo_host_session* s, int as_constants) { return (s->slam && s->slam->load_state(as_constants != 0)) ? 1 : 0; }
int pgo_host_trigger(pgo_host_session* s) { return s->slam->reinit_ceres_problem_onnewloopedge_optimize6DOF_once() ? 1 : 0; }
int pgo_host_n_nodes(pgo_host_session* s) { return s->slam->nNodes(); }
int pgo_host_solved_until(pgo_host_session* s) { return s->slam->solvedUntil(); }
int pgo_host_node_pose_exists(pgo_host_session* s, int i) { return s->slam->nodePoseExists(i) ? 1 : 0; }
void pgo_host_get_node_pose(pgo_host_session* s, int i, double* T16) { const Matrix4d T = s->slam->getNodePose(i); std::copy(T.d.begin(), T.d.end(), T16); }
double pgo_host_switch(pgo_host_session* s, int e) { return s->slam->get_loopedge_switching_variable_val(e); }
int pgo_host_n_added_edges(pgo_host_session* s) { return (int)s->slam->added_edges().size(); }
void pgo_host_get_added_edges(pgo_host_session* s, int32_t* c1, int32_t* c2, double* w, int32_t* sw) {
    const auto& v = s->slam->added_edges();
    for (size_t k = 0; k < v.size(); ++k) { c1[k] = v[k].c1; c2[k] = v[k].c2; w[k] = v[k].weight; sw[k] = v[k].switch_idx; }
}
int pgo_host_n_regularizers(pgo_host_session* s) { return (int)s->slam->regularizers().size(); }
void pgo_host_get_regularizers(pgo_host_session* s, int32_t* node, double* w, double* T16) {
    const auto& v = s->slam->regularizers();
    for (size_t k = 0; k < v.size(); ++k) { node[k] = v[k].node; w[k] = v[k].weight; std::copy(v[k].target.d.begin(), v[k].target.d.end(), T16 + 16 * k); }
}
void pgo_host_get_initial_guess(pgo_host_session* s, double* quat, double* t) {
    std::copy(s->slam->last_initial_quat().begin(), s->slam->last_initial_quat().end(), quat);
    std::copy(s->slam->last_initial_t().begin(), s->slam->last_initial_t().end(), t);
}
void pgo_host_get_summary(pgo_host_session* s, pgo_summary* out) { *out = s->slam->last_summary(); }
int pgo_host_last_error(pgo_host_session* s) { return s->slam->last_error(); }
int pgo_host_save_as_json(pgo_host_session* s, const char* base_path) { return s->slam->saveAsJSON(base_path) ? 1 : 0; }
}
