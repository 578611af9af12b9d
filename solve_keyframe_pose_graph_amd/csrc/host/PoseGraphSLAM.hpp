// PoseGraphSLAM.hpp — ROS-free host side above the C-ABI (include/pgo.h), mirroring the solver-facing interface of the
// reference's `PoseGraphSLAM` class: same method names, argument meaning and threading contract, so that a maintainer can
// swap the Ceres calls of the reference for libpgo and keep NodeDataManager, the ROS callbacks and the Worlds/kidnap
// bookkeeping untouched.
//
// Mirrored interface (reference file:line):
//   opt-variable storage, xyzw quaternion + translation + one switch per loop edge      src/PoseGraphSLAM.h:153-175
//   getNodePose / nodePoseExists / nNodes / getAllNodePose / solvedUntil                src/PoseGraphSLAM.cpp:178-224, PoseGraphSLAM.h:120
//   allocate_and_append_new_opt_variable_withpose / update_opt_variable_with            src/PoseGraphSLAM.cpp:226-335
//   allocate_and_append_new_edge_switch_var (init 0.99) / get_loopedge_switching_variable_val   :339-361, PoseGraphSLAM.h:219
//   one wake-up of reinit_ceres_problem_onnewloopedge_optimize6DOF(): steps -0- .. -6-   src/PoseGraphSLAM.cpp:1287-1940
// The data source is abstract: in the reference it is `NodeDataManager` + `Worlds` (NodeDataManager.h:95-111,179; Worlds.h:50-76).
#pragma once
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "pgo.h"

namespace pgo_host {

// Eigen::Matrix4d stand-in: 16 doubles, column-major.
struct Matrix4d {
    std::array<double, 16> d;
    double& operator()(int r, int c) { return d[c * 4 + r]; }
    double operator()(int r, int c) const { return d[c * 4 + r]; }
    static Matrix4d Identity();
    Matrix4d operator*(const Matrix4d& o) const;
    Matrix4d inverse() const;   // affine inverse [A^-1, -A^-1 t], A^-1 by cofactors (what Eigen's general inverse() gives for a pose)
};

// PoseManipUtils::{raw_xyzw_to_eigenmat, eigenmat_to_raw_xyzw, R2ypr}  (reference src/utils/PoseManipUtils.cpp:61-98,143-158)
void raw_xyzw_to_eigenmat(const double* quat, const double* t, Matrix4d& dst);
void eigenmat_to_raw_xyzw(const Matrix4d& T, double* quat, double* t);
double yaw_degrees(const Matrix4d& T);

// What the trigger reads from NodeDataManager / Worlds.  Method names follow the reference.
class GraphSource {
public:
    virtual ~GraphSource() {}
    virtual int getNodeLen() const = 0;
    virtual Matrix4d getNodePose(int i) const = 0;                  // odometry pose w_M_i in its own world
    virtual int which_world_is_this_node(int i) const = 0;          // = which_world_is_this(getNodeTimestamp(i)); negative while kidnapped
    virtual int getEdgeLen() const = 0;
    virtual Matrix4d getEdgePose(int e) const = 0;                  // b_T_a
    virtual double getEdgeWeight(int e) const = 0;
    virtual std::pair<int, int> getEdgeIdxInfo(int e) const = 0;    // (a = current, b = previous)
    virtual bool curr_kidnap_status() const = 0;
    virtual int n_worlds() const = 0;
    virtual int nodeidx_of_world_i_started(int w) const = 0;
    virtual int nodeidx_of_world_i_ended(int w) const = 0;
    // Worlds
    virtual int find_setID_of_world_i(int w) const = 0;             // negative for kidnapped / unknown worlds
    virtual bool is_exist(int m, int n) const = 0;
    virtual Matrix4d getPoseBetweenWorlds(int m, int n) const = 0;  // m_T_n
    virtual void setPoseBetweenWorlds(int m, int n, const Matrix4d& m_T_n) = 0;
    virtual void getWorld2SetIDMap(std::map<int, int>& out) const = 0;
};

// One record per residual block the trigger adds — what a reader of the reference would expect `AddResidualBlock` to receive.
struct AddedEdge { int c1, c2; double weight; int switch_idx; /* -1: SixDOFError */ };
struct AddedRegularizer { int node; double weight; Matrix4d target; };

class PoseGraphSLAM {
public:
    explicit PoseGraphSLAM(GraphSource* manager, const pgo_options* options = nullptr);
    ~PoseGraphSLAM();
    bool ok() const { return problem_ != nullptr; }
    // Steps -3-/-4- (odometry measurements, yaw weights, VIO-derived initial guesses) run as device kernels from the resident raw VIO
    // poses by default; `false` computes them on the host thread like the reference — kept for the parity tests.
    void set_device_graph_construction(bool on) { device_graph_construction_ = on; }

    // PoseGraphSLAM::load_state (reference src/PoseGraphSLAM.cpp:30-165): the keyframes the data source holds beyond the current
    // optimisation variables are a previously solved map (e.g. from solved_posegraph.json).  Each becomes an optimisation variable at
    // its pose in its world-set's frame (ws_T_w * w_T_c), marked constant when `optimization_variable_as_constants` (the reference's
    // SetParameterBlockConstant, :143-144), and solved_until moves to the last of them — so the trigger adds no odometry residues among
    // them and later loop closures localise new keyframes against the fixed map.
    bool load_state(bool optimization_variable_as_constants = true);

    // One wake-up of the reference's trigger loop body.  Returns true when a solve ran.
    bool reinit_ceres_problem_onnewloopedge_optimize6DOF_once();
    int get_reinit_ceres_problem_onnewloopedge_optimize6DOF_status() const { return status_; }

    // thread-safe readers (same names as the reference)
    const Matrix4d getNodePose(int i) const;
    bool nodePoseExists(int i) const;
    int nNodes() const;
    void getAllNodePose(std::vector<Matrix4d>& vec_w_T_ci) const;
    int solvedUntil() const;
    double get_loopedge_switching_variable_val(int i) const;
    // writes base_path + "/log_optimized_poses.json" with the keys of the reference's saveAsJSON (src/PoseGraphSLAM.cpp:1111-1207):
    // meta_data.nNodes, PoseGraphSLAM_nodes[{node_i, wTc_opt, w_T_c_odom (+ _prettyprint)}], PoseGraphSLAM_loopedgeinfo[{getEdge_i, a, b,
    // world_of_a, world_of_b, weight, getEdgePose, getEdgePose_after_opt, switching_var_after_opt}]; matrices as Eigen CSVFormat strings
    bool saveAsJSON(const std::string& base_path) const;

    // introspection for the parity tests
    const std::vector<AddedEdge>& added_edges() const { return added_edges_; }
    const std::vector<AddedRegularizer>& regularizers() const { return regs_; }
    const std::vector<double>& last_initial_quat() const { return init_quat_; }
    const std::vector<double>& last_initial_t() const { return init_t_; }
    const pgo_summary& last_summary() const { return summary_; }
    int last_error() const { return last_rc_; }

private:
    void allocate_and_append_new_opt_variable_withpose(const Matrix4d& pose);
    bool update_opt_variable_with(int i, const Matrix4d& pose);
    void allocate_and_append_new_edge_switch_var();
    int n_opt_variables() const;
    int n_opt_switch() const;

    GraphSource* manager;
    pgo_problem* problem_ = nullptr;
    mutable std::mutex mutex_opt_vars;
    std::vector<double> _opt_quat_, _opt_t_, _opt_switch_;   // xyzw ; xyz ; one per loop edge
    int solved_until = 0;
    int prev_loopedge_len = 0, prev_node_len = 0;
    int odom_until_ = 0;                 // odometry residues exist for every keyframe below this index
    int status_ = -1, last_rc_ = 0;
    bool device_graph_construction_ = true;
    std::map<int, std::tuple<int, int>> changes_to_setid_on_set_union;
    std::vector<AddedEdge> added_edges_;
    std::vector<AddedRegularizer> regs_;
    std::vector<double> init_quat_, init_t_;
    pgo_summary summary_;
};

// A plain in-memory GraphSource (stands in for NodeDataManager + Worlds in tests and examples).
class VectorGraphSource : public GraphSource {
public:
    // stamp < 0: keyframes are stamped idx * 0.1 s (the reference stores ros::Time; only equality and order matter here)
    void add_node(int world, const Matrix4d& w_M_i, double stamp = -1.0) {
        node_stamp_.push_back(stamp >= 0 ? stamp : 0.1 * (double)node_pose_.size());
        if (world >= 0) {   // first / last keyframe of every world, kept incrementally (the trigger asks per world, per wake-up)
            if ((size_t)world >= world_first_.size()) { world_first_.resize((size_t)world + 1, -1); world_last_.resize((size_t)world + 1, -1); }
            if (world_first_[world] < 0) world_first_[world] = (int)node_pose_.size();
            world_last_[world] = (int)node_pose_.size();
        }
        node_world_.push_back(world); node_pose_.push_back(w_M_i);
    }
    void add_loop_edge(int a, int b, const Matrix4d& b_T_a, double weight, const std::string& description = std::string()) {
        edge_ab_.push_back({a, b}); edge_pose_.push_back(b_T_a); edge_w_.push_back(weight); edge_desc_.push_back(description);
    }
    void set_kidnapped(bool k) { kidnapped_ = k; }
    void reset() { node_world_.clear(); node_pose_.clear(); node_stamp_.clear(); edge_ab_.clear(); edge_pose_.clear(); edge_w_.clear(); edge_desc_.clear(); set_of_.clear(); set_T_world_.clear(); world_first_.clear(); world_last_.clear(); kidnapped_ = false; }
    double getNodeTimestamp(int i) const { return node_stamp_[i]; }
    const std::string& getEdgeDescriptionString(int e) const { return edge_desc_[e]; }
    std::string disjoint_set_status() const;      // Worlds::disjoint_set_status (reference src/Worlds.cpp:333-370)

    int getNodeLen() const override { return (int)node_pose_.size(); }
    Matrix4d getNodePose(int i) const override { return node_pose_[i]; }
    int which_world_is_this_node(int i) const override { return node_world_[i]; }
    int getEdgeLen() const override { return (int)edge_ab_.size(); }
    Matrix4d getEdgePose(int e) const override { return edge_pose_[e]; }
    double getEdgeWeight(int e) const override { return edge_w_[e]; }
    std::pair<int, int> getEdgeIdxInfo(int e) const override { return edge_ab_[e]; }
    bool curr_kidnap_status() const override { return kidnapped_; }
    int n_worlds() const override;
    int nodeidx_of_world_i_started(int w) const override;
    int nodeidx_of_world_i_ended(int w) const override;
    int find_setID_of_world_i(int w) const override;
    bool is_exist(int m, int n) const override;
    Matrix4d getPoseBetweenWorlds(int m, int n) const override;
    void setPoseBetweenWorlds(int m, int n, const Matrix4d& m_T_n) override;
    void getWorld2SetIDMap(std::map<int, int>& out) const override;

private:
    void ensure_world(int w) const;
    std::vector<int> node_world_, world_first_, world_last_;
    std::vector<Matrix4d> node_pose_;
    std::vector<std::pair<int, int>> edge_ab_;
    std::vector<Matrix4d> edge_pose_;
    std::vector<double> edge_w_, node_stamp_;
    std::vector<std::string> edge_desc_;
    bool kidnapped_ = false;
    mutable std::vector<int> set_of_;            // world -> set id (= smallest world id of the merged set)
    mutable std::vector<Matrix4d> set_T_world_;  // pose of the world in its set's frame
};

}  // namespace pgo_host
