"""K0 (graph construction from raw VIO poses) on C4's 200k keyframes / ~1M odometry edges: kernel time vs the CPU restatement
of the reference's loop (oracle), and the whole-call times of the two ways to hand the edges to libpgo."""
import json, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import binding as orc
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util

g = graphgen.config("C4")
w_M = util.poses_to_matrices(g.init_q, g.init_t)
n = g.n_poses
out = {"keyframes": n}
t0 = time.perf_counter(); c1, c2, T, w = orc.odometry_edges_from_vio(w_M, None, 0, n, 5, True); out["cpu_oracle_loop_ms"] = (time.perf_counter() - t0) * 1e3
out["edges"] = len(c1)
P = capi.Problem()
t0 = time.perf_counter(); P.set_vio_poses(0, w_M); out["upload_vio_ms"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter(); na = P.add_odometry_edges_from_vio(None, 0, n); out["device_call_ms (idx H2D + kernel + records D2H)"] = (time.perf_counter() - t0) * 1e3
assert na == len(c1)
ms, by = P.time_vio_odometry_kernel(5, 50)
out["k0_kernel_ms"] = ms; out["k0_algorithmic_bytes"] = by; out["k0_GBps"] = by / ms / 1e6
P2 = capi.Problem()
t0 = time.perf_counter(); P2.add_relpose_edges(c1, c2, T, w); out["host_matrices_call_ms (Matrix4d -> records on the host)"] = (time.perf_counter() - t0) * 1e3
_, _, ra = P.relpose_edge_records(0, na); _, _, rb = P2.relpose_edge_records(0, na)
out["max_record_difference"] = float(np.abs(ra - rb).max())
print(json.dumps(out, indent=1))
