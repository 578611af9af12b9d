#!/usr/bin/env python3
"""One MEASURED run of the CPU port (oracle/: Jet autodiff + Ceres-style LM + exact block-sparse Cholesky standing in for SPARSE_NORMAL_CHOLESKY, 1 thread — the
reference sets no num_threads, src/PoseGraphSLAM.cpp:1268-1272) on a BASELINE.json config at FULL size, so that bench.py's cpu_baseline can quote a number that is
neither scaled nor extrapolated (missing 4).  Hours of CPU for C3: run it once per round, commit the JSON under profiles/.

  python scripts/cpu_c3_full.py [config=C3] [iterations=3] [out=profiles/r04_cpu_c3_full.json]

`kind` stays "port": the up-looking block Cholesky with AMD ordering is not CHOLMOD's supernodal code; it is context for the GPU number, never credit."""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402
from tests import util  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r04_cpu_%s_full.json" % name.lower())
switchable = name != "C2"
g = graphgen.config(name)
O = util.oracle_problem(g, switchable)
q, t, s = util.initial_state(g, switchable)
opt = ob.default_options(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0, num_threads=1)
t0 = time.time()
_, _, _, sm = O.solve(q, t, s, opt)
wall = time.time() - t0
its = [dict(iteration=sm.iterations[k].iteration, cost=sm.iterations[k].cost, successful=int(sm.iterations[k].step_is_successful)) for k in range(sm.num_logged)]
cpu = ""
try:
    with open("/proc/cpuinfo") as f:
        cpu = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")][0]
except Exception:
    pass
rec = dict(config=name, n_poses=int(g.n_poses), n_edges=int(g.n_odom + g.n_loops), lm_iterations=int(sm.num_iterations), threads=1, wall_seconds=wall, seconds_total=sm.seconds_total,
           seconds_linear_solver=sm.seconds_linear_solver, seconds_jacobian=sm.seconds_jacobian, cholesky_fill_blocks=int(sm.chol_nnz_blocks),
           lm_iterations_per_second=sm.num_iterations / sm.seconds_total, seconds_per_lm_iteration=sm.seconds_total / max(1, sm.num_iterations),
           initial_cost=sm.initial_cost, final_cost=sm.final_cost, iterations=its, kind="port", host_cpu=cpu, host_cpus=os.cpu_count(), machine=platform.node(),
           note="oracle/pgo_oracle.cpp, 1 thread, exact up-looking block Cholesky (AMD ordering); measured at full size, nothing scaled")
with open(out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec))
