"""Overhead of the multi-rank machinery on ONE GPU: C3 solved plainly vs with a 1-rank RCCL communicator (every collective and every
extra kernel of the N-rank path issued, nothing to exchange).  The difference per CG iteration is the fixed cost the N-rank run pays
on top of its xGMI transfer time."""
import json, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from solve_keyframe_pose_graph_amd import capi, graphgen
from tests import util

g = graphgen.config("C3")
q, t, s = util.initial_state(g, True)
out = {}
import os
mg = len(sys.argv) > 1 and sys.argv[1] == "mg"      # "mg": library defaults on both sides (hybrid block-Jacobi / multigrid); default: block-Jacobi on both sides
for name in ("plain", "rccl_1rank"):
    P = util.pgo_problem(g, True) if mg else util.pgo_problem(g, True, mg_min_keyframes=0, coarse_aggregates=0)
    if name != "plain":
        P.comm_init(0, 1, capi.Problem.comm_unique_id())
    P.solve(q, t, s)                      # warm-up (graph build, hipGraph capture)
    _, _, _, sm = P.solve(q, t, s)
    out[name] = {"seconds_device": sm.seconds_device, "cg_iterations": int(sm.cg_iterations), "final_cost": sm.final_cost,
                 "us_per_cg_iteration": 1e6 * sm.seconds_device / sm.cg_iterations}
    if name != "plain":
        P.comm_destroy()
    P.close()
out["mode"] = ("defaults (hybrid multigrid)" if mg else "block-Jacobi") + (", RCCL chunks as hipGraphs" if os.environ.get("PGO_RCCL_GRAPH") == "1" else "")
out["extra_us_per_cg_iteration"] = out["rccl_1rank"]["us_per_cg_iteration"] - out["plain"]["us_per_cg_iteration"]
print(json.dumps(out, indent=1), flush=True)
open('gpurun_out/multi_overhead%s%s.json' % ('_mg' if mg else '', '_graph' if os.environ.get('PGO_RCCL_GRAPH') == '1' else ''), 'w').write(json.dumps(out, indent=1))
