"""Where do the ~26 us of the matrix-free matvec (mf_spmv_kernel) go?  Builds a VARIANT libpgo.so with -DPGO_MF_TIMELINE (thread 0 of every workgroup records the
100-MHz wall clock at its phase boundaries, after a full wait for outstanding memory operations), runs the matvec alone on C3 (pgo_time_kernel 4) and prints, over the
workgroups, the distribution of every phase's start time relative to the earliest kernel entry, and of its length.  Development aid; run on the GPU box."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solve_keyframe_pose_graph_amd import _build  # noqa: E402

variant = os.path.join(ROOT, "build", "variants", "libpgo_timeline.so")
if "--built" not in sys.argv:
    os.makedirs(os.path.dirname(variant), exist_ok=True)
    srcs = [os.path.join(_build.CSRC, s) for s in _build.HIP_SOURCES]
    subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-DPGO_MF_TIMELINE", "-I", _build.INCLUDE, "-I", _build.CSRC,
                           "-x", "hip"] + srcs + ["-o", variant, "-ldl"])
os.environ["PGO_LIBPGO_OVERRIDE"] = variant
from solve_keyframe_pose_graph_amd import capi, graphgen  # noqa: E402
from tests import util  # noqa: E402

name = [a for a in sys.argv[1:] if not a.startswith("--")]
name = name[0] if name else "C3"
g = graphgen.config(name)
q, t, s = util.initial_state(g, True)
P = util.pgo_problem(g, True)
P.solve_begin(q, t, s)
for _ in range(3):
    P.lm_step(ignore_termination=True)
ms, by = P.time_kernel(4, 20)
print("matvec alone (instrumented build): %.2f us, %.1f MB -> %.2f TB/s" % (ms * 1e3, by / 1e6, by / ms / 1e9))
lib = capi.load()
n = 1024 * 16
buf = (C.c_ulonglong * n)()
rc = lib.pgo_debug_mf_timeline(buf, n)
assert rc == 0, rc
T = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64)
P.solve_end(); P.close()
live = T[:, 0] > 0
T = T[live]
t0 = T[:, 0].min()
us = (T - t0) / 100.0      # 100 MHz -> microseconds
labels = ["entry", "re-reduction done", "t1 bounds", "t1 phase 0 + barrier", "t1 records in", "t1 far gathers in", "t1 products + barrier", "t1 phase-B loads in", "t1 end",
          "t2 bounds", "t2 phase 0 + barrier", "t2 records in", "t2 far gathers in", "t2 products + barrier", "t2 phase-B loads in", "t2 end"]
print("%d workgroups; time since the first workgroup's entry, microseconds: min / median / p90 / max   (length of the phase: median)" % len(T))
second = T[:, 9] > 0
for k, lab in enumerate(labels):
    col = us[:, k] if k < 9 else us[second, k]
    if len(col) == 0:
        continue
    prev = (us[:, k - 1] if k < 9 else us[second, k - 1]) if k > 0 else col
    print("%-24s %6.2f %6.2f %6.2f %6.2f   (%5.2f)" % (lab, col.min(), np.median(col), np.percentile(col, 90), col.max(), np.median(col - prev)))
print("workgroups with a second tile: %d" % int(second.sum()))
