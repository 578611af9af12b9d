"""Replays a recorded session through the MI355X solver:  python -m solve_keyframe_pose_graph_amd.replay <dir> [--out <dir>] [--every K]

<dir>/log_posegraph.json is what the reference's NodeDataManager::saveAsJSON writes (src/NodeDataManager.cpp:503-628).  Keyframes are
streamed in index order, a loop edge arrives once both of its keyframes exist, and the trigger of
reinit_ceres_problem_onnewloopedge_optimize6DOF() (csrc/host/PoseGraphSLAM.cpp) is woken every K keyframes — it solves only when a new
loop edge has arrived, like the reference thread.  Writes <out>/log_optimized_poses.json (PoseGraphSLAM::saveAsJSON keys,
src/PoseGraphSLAM.cpp:1111-1207), <out>/log_posegraph.json and <out>/optimized.g2o.  Needs a GPU: there is no CPU fallback."""
import argparse
import json
import os
import sys
import time

from .pose_graph_slam import GraphSource, PoseGraphSLAM


def replay(src, every=50, **opt_kw):
    """src: a loaded GraphSource.  Returns (PoseGraphSLAM session, list of per-trigger records)."""
    n, m = src.n_nodes(), src.n_edges()
    edges = sorted((max(src.edge(e)[0], src.edge(e)[1]), e) for e in range(m))
    S = PoseGraphSLAM(**opt_kw)
    log, k = [], 0
    for i in range(n):
        w, st, T = src.node(i)
        S.add_node(w, T, stamp=st)
        while k < m and edges[k][0] <= i:
            a, b, wgt, bTa, desc = src.edge(edges[k][1])
            S.add_loop_edge(a, b, bTa, wgt, desc)
            k += 1
        if (i + 1) % every == 0 or i == n - 1:
            t0 = time.perf_counter()
            if S.reinit_ceres_problem_onnewloopedge_optimize6DOF_once():
                sm = S.summary()
                log.append({"keyframes": i + 1, "loop_edges": k, "lm_iterations": sm.num_iterations, "cg_iterations": sm.cg_iterations,
                            "initial_cost": sm.initial_cost, "final_cost": sm.final_cost, "wall_ms": (time.perf_counter() - t0) * 1e3})
    return S, log


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--every", type=int, default=50)
    a = ap.parse_args(argv)
    out = a.out or a.dir
    os.makedirs(out, exist_ok=True)
    src = GraphSource().load_posegraph_json(a.dir)
    S, log = replay(src, a.every)
    S.saveAsJSON(out)
    if os.path.abspath(out) != os.path.abspath(a.dir):
        S.save_posegraph_json(out)
    S.export_g2o(os.path.join(out, "optimized.g2o"), optimized=True)
    json.dump({"triggers": log}, sys.stdout, indent=1)
    print()
    return 0


if __name__ == "__main__":
    sys.exit(main())
