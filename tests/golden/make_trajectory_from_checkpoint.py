#!/usr/bin/env python3
"""Writes an N-iteration trajectory golden (the format of c3_ten_iterations.json) from the checkpoint a `converge` run of tests/golden/make_c3_trajectory.py keeps after
every iteration (<output>.state.npz: the state after iteration next_it - 1 and the whole log).  Used for tests/golden/c4_twenty_iterations.json: the run to convergence of C4 takes
about seven CPU-hours (35 iterations with up to 4 000 CG iterations each); its first twenty iterations — 2.4 hours — are the anchor the GPU suite compares with.
  python tests/golden/make_trajectory_from_checkpoint.py <state.npz> <config> <n_iterations> <output name>"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solve_keyframe_pose_graph_amd import graphgen  # noqa: E402

state, name, n_iter, out_name = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
z = np.load(state, allow_pickle=False)
log = json.loads(str(z["log"]))
assert int(z["next_it"]) == n_iter + 1 and len(log) == n_iter + 1, (int(z["next_it"]), len(log))      # the state IS the one after iteration n_iter
g = graphgen.config(name)
t, s = z["t"], z["s"]
assert t.shape[0] == g.n_poses and s.shape[0] == g.n_loops
out = dict(note="first %d iterations of a `converge` run of tests/golden/make_c3_trajectory.py (oracle Jacobians + scipy CG, rtol 1e-12, preconditioner mg, + Python restatement of the "
                "Ceres LM loop), written from its checkpoint by tests/golden/make_trajectory_from_checkpoint.py" % n_iter,
           config=name, n_poses=int(g.n_poses), n_edges=int(g.n_odom + g.n_loops), iterations=log, final_t_sample=t[::997].tolist(), final_s_sample=s[::997].tolist())
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), out_name), "w") as f:
    json.dump(out, f)
print("wrote", out_name, "final cost", log[-1]["cost"])
