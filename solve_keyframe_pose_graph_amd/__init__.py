"""solve_keyframe_pose_graph_amd — MI355X-native 6-DoF pose-graph Levenberg-Marquardt solver.

Drop-in for the ONE hot path of mpkuse/solve_keyframe_pose_graph (the Ceres solve inside
PoseGraphSLAM::reinit_ceres_problem_onnewloopedge_optimize6DOF, reference src/PoseGraphSLAM.cpp:1251-1950,
with the cost functors of src/CeresResidues.h).  The product is the C-ABI library libpgo.so
(include/pgo.h, hand-written HIP for gfx950); this package is the thin host side above it.
"""
__all__ = ["capi", "graphgen", "sharding"]
