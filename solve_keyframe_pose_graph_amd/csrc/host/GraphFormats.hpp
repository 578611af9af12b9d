// GraphFormats.hpp — the on-disk formats either side of the solver path, ROS-free:
//   log_posegraph.json        what NodeDataManager::saveAsJSON writes and ::loadFromJSON reads  (reference src/NodeDataManager.cpp:503-754)
//   log_optimized_poses.json  what PoseGraphSLAM::saveAsJSON writes                              (reference src/PoseGraphSLAM.cpp:1111-1207; writer in PoseGraphSLAM.cpp)
//   *.g2o                     VERTEX_SE3:QUAT / EDGE_SE3:QUAT export for cross-checks with external solvers (not in the reference)
// plus the small JSON reader they need (the reference uses nlohmann::json, which this image does not have).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "PoseGraphSLAM.hpp"

namespace pgo_host {

// ---- minimal JSON document (objects keep insertion order irrelevant: lookups by key) ----
struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JsonValue> arr;
    std::map<std::string, JsonValue> obj;

    bool has(const std::string& k) const { return kind == Object && obj.count(k) > 0; }
    const JsonValue& at(const std::string& k) const;   // Null value when absent
    const JsonValue& at(size_t i) const;
    size_t size() const { return kind == Array ? arr.size() : (kind == Object ? obj.size() : 0); }
    int as_int(int dflt = 0) const { return kind == Number ? (int)num : dflt; }
    double as_double(double dflt = 0) const { return kind == Number ? num : dflt; }
    const std::string& as_string() const { return str; }
};
// Parses a complete JSON text; on failure returns false and describes the problem (with byte offset) in *err.
bool json_parse(const std::string& text, JsonValue& out, std::string* err);
std::string json_escape(const std::string& s);

// Eigen `IOFormat(FullPrecision, DontAlignCols, ",", ";")` of a Matrix4d and its inverse, PoseManipUtils::string_to_eigenmat
// (reference src/utils/PoseManipUtils.cpp:272-295): rows separated by ';', entries by ','.
std::string matrix4d_to_csv(const Matrix4d& M);
bool csv_to_matrix4d(const std::string& s, Matrix4d& M);
// PoseManipUtils::prettyprintMatrix4d (src/utils/PoseManipUtils.cpp:206-215)
std::string prettyprint_matrix4d(const Matrix4d& M);

// log_posegraph.json.  Keys as the reference writes them: meta_data{getNodeLen,getEdgeLen,n_worlds}, nodes[{timestamp, idx, world_id,
// wTc, wTc_pretty, cov}], loopedges[{idx0, idx1, timestamp0, timestamp1, world0_id, world1_id, code, b_T_a, b_T_a_pretty, weight,
// description}], world_info[{id, nodeidx_of_world_i_started, nodeidx_of_world_i_ended}], kidnap_info[...], disjoint_set_status.
bool save_posegraph_json(const VectorGraphSource& src, const std::string& base_path);
// Loads nodes (timestamp, wTc; "world_id" when present, else world 0) and loop edges (idx0 = a, idx1 = b, b_T_a, weight, description);
// edge_mask as the reference: empty = all, else edge i is loaded iff edge_mask[i].  Checks meta_data against the array sizes and
// the edge timestamps against the node timestamps like the reference (:659-666, :736-747) — returning false instead of exit(1).
bool load_posegraph_json(VectorGraphSource& src, const std::string& base_path, const std::vector<bool>& edge_mask, std::string* err);

// solved_posegraph.json — the saved state of a finished session that the reference writes in Composer::saveStateToDisk and reads back in
// Composer::loadStateFromDisk to continue on top of a previous map (reference src/Composer.cpp:952-1177, src/Worlds.cpp:449-667):
//   SolvedPoseGraph[{w_T_c{rows, cols, data, data_pretty}, worldID, setID_of_worldID, stampNSec, seq}]  corrected keyframe poses
//   KidnapTimestamps{kidnap_starts[{stampNSec}], kidnap_ends[{stampNSec}]}
//   WorldsData{rel_pose_between_worlds__wb_T_wa[{node_b, node_a, wb_T_wa{rows, cols, data, data_pretty}, info_wb_T_wa}],
//              vec_world_starts[{stampNSec}], vec_world_ends[{stampNSec}], disjoint_set{debug_string, log_string}}
// Matrices use RawFileIO::eigen_matrix_to_json's layout (entries ", "-separated, rows "\n"-separated); the disjoint set travels as the
// replayable command log "add_element:k;union_sets:max,min;" the reference parses (Worlds.cpp:560-640).
bool save_solved_posegraph_json(const VectorGraphSource& src, const std::vector<Matrix4d>& w_T_c, const std::string& base_path);
// Rebuilds a source from the file: one keyframe per SolvedPoseGraph entry (its corrected pose stands in for the odometry pose of the
// loaded map, its world and stamp restored), the world merges replayed from WorldsData; w_T_c receives the corrected poses — the
// caller marks those keyframes constant like PoseGraphSLAM::load_state (reference src/PoseGraphSLAM.cpp:143-144 -> pgo_set_nodes_constant).
bool load_solved_posegraph_json(VectorGraphSource& src, std::vector<Matrix4d>& w_T_c, const std::string& base_path, std::string* err);

// g2o export of a solved or unsolved graph: one VERTEX_SE3:QUAT per pose (x y z qx qy qz qw), one EDGE_SE3:QUAT per edge (c1 -> c2
// with the measurement c1_T_c2) and the 21 upper-triangular entries of the information matrix of (dt, dq.vec): w^2 on translation,
// 4 w^2 on the quaternion vector part (the reference residual is [dt; 2 dq.vec]·w, g2o's is [dt; dq.vec]).
struct G2oEdge { int c1, c2; Matrix4d c1_T_c2; double weight; };
bool export_g2o(const std::string& path, const std::vector<Matrix4d>& poses, const std::vector<G2oEdge>& edges);

}  // namespace pgo_host
