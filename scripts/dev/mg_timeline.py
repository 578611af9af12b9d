"""Where do the ~10 us of a multigrid level kernel go?  Variant build with -DPGO_MG_TIMELINE: thread 0 of every workgroup of mg_down_kernel records the 100-MHz wall clock at its
phase boundaries (after a full wait for outstanding memory operations); per level: medians over the workgroups since the first entry.  Development aid; run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solve_keyframe_pose_graph_amd import _build  # noqa: E402

variant = os.path.join(ROOT, "build", "variants", "libpgo_mgtl.so")
os.makedirs(os.path.dirname(variant), exist_ok=True)
srcs = [os.path.join(_build.CSRC, s) for s in _build.HIP_SOURCES]
subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-DPGO_MG_TIMELINE", "-I", _build.INCLUDE, "-I", _build.CSRC, "-x", "hip"] + srcs + ["-o", variant, "-ldl"])
os.environ["PGO_LIBPGO_OVERRIDE"] = variant
from solve_keyframe_pose_graph_amd import capi, graphgen  # noqa: E402
from tests import util  # noqa: E402

g = graphgen.config(sys.argv[1] if len(sys.argv) > 1 else "C3")
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True)
P.solve_begin(q, t, s)
for _ in range(3):
    P.lm_step(ignore_termination=True)
ms, _ = P.time_kernel(7, 20)
print("level kernels of one cycle (instrumented build): %.2f us" % (ms * 1e3))
lib = capi.load()
n = 4 * 1024 * 8
buf = (C.c_ulonglong * n)()
assert lib.pgo_debug_mg_timeline(buf, n) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(4, 1024, 8).astype(np.float64)
P.solve_end(); P.close()
labels = ["entry", "tile info / row range in", "operands (r, d, agg_ptr, Dinv) in", "row product (col -> x, blocks) done", "gather + 2 barriers", "restriction sums stored", "x_next stored"]
for lvl in range(4):
    live = T[lvl, :, 0] > 0
    if not live.any():
        continue
    X = T[lvl][live]
    us = (X - X[:, 0].min()) / 100.0
    print("level %d: %d workgroups; microseconds since the first entry: median / p90 / max   (phase length, median)" % (lvl + 1, len(X)))
    for k, lab in enumerate(labels):
        col = us[:, k]
        if (X[:, k] == 0).all():
            continue
        prev = us[:, k - 1] if k else col
        print("  %-40s %6.2f %6.2f %6.2f   (%5.2f)" % (lab, np.median(col), np.percentile(col, 90), col.max(), np.median(col - prev)))
