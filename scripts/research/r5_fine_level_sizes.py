"""Host-only: level sizes of the hierarchy with and without the smoothed keyframe transition (pgo_options::mg_smoothed_fine) on the graph types of scripts/dev/r05/opt_types.py, beside
the measured gain or loss of the option (profiles/r05_smoothed_fine_measured.txt) — the raw material of a default "by the density of the resulting levels".  Uses the test shim
tests/native/libmg_host.so (built by tests/test_mg_hierarchy.py's fixture; aggregates of 8, then of 4, smoothed transition above level 1, loop discount 0).
  python scripts/research/r5_fine_level_sizes.py"""
import ctypes as C
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from solve_keyframe_pose_graph_amd import graphgen
so = os.path.join(ROOT, "tests", "native", "libmg_host.so")
src = os.path.join(ROOT, "tests", "native", "mg_host.cpp")
hdr = os.path.join(ROOT, "solve_keyframe_pose_graph_amd", "csrc", "pgo_mg_host.hpp")
if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.dirname(hdr), "-o", so, src])
lib = C.CDLL(so)
lib.mgh_build_fine.restype = C.c_void_p
lib.mgh_build_regroup.restype = C.c_void_p
lib.mgh_fine_nnzb.restype = C.c_longlong


def I32(x): return np.ascontiguousarray(x, dtype=np.int32)
def ptr(a, t): return a.ctypes.data_as(C.POINTER(t))


def sizes(g, fine):
    N = g.n_poses
    nf = np.ones(N, np.uint8); nf[0] = 0
    rc1, rc2, sc1, sc2 = I32(g.odom_c1), I32(g.odom_c2), I32(g.loop_c1), I32(g.loop_c2)
    rw = np.ascontiguousarray(g.odom_w, dtype=np.float64)
    args = (C.c_longlong(N), ptr(nf, C.c_ubyte), C.c_longlong(len(rc1)), ptr(rc1, C.c_int), ptr(rc2, C.c_int), ptr(rw, C.c_double), C.c_longlong(len(sc1)), ptr(sc1, C.c_int), ptr(sc2, C.c_int))
    if fine:
        h = lib.mgh_build_fine(*args, 3, 2, 512, 32, 12, 1, 64)
    else:
        one = np.ones(max(1, len(sc1)))
        h = lib.mgh_build_regroup(*args, ptr(one, C.c_double), ptr(one, C.c_double), 0, 3, 2, 512, 32, 12, 1, C.c_double(0.0), 64)
    h = C.c_void_p(h)
    out = []
    for l in range(lib.mgh_levels(h)):
        sz = np.zeros(6, np.int64)
        lib.mgh_sizes(h, l, ptr(sz, C.c_longlong))
        out.append((int(sz[0]), int(sz[1])))
    ps = None
    if fine:
        ss = np.zeros(3, np.int64)
        lib.mgh_smoothed_sizes(h, -1, ptr(ss, C.c_longlong))
        ps = int(ss[0])
    lib.mgh_free(h)
    return out, ps


cases = [("C3 (100k / 100k loops, outliers)", lambda: graphgen.config("C3"), "-20 %"),
         ("C4 (200k, 4 worlds)", lambda: graphgen.config("C4"), "-43 %"),
         ("60k keyframes, 6k loops (chain-like)", lambda: graphgen.generate(60000, 6000, odom_f_max=2, seed=7), "+28 %"),
         ("60k keyframes, 60k loops, no outliers", lambda: graphgen.generate(60000, 60000, odom_f_max=2, seed=8, outlier_frac=0.0), "+18 %"),
         ("50k keyframes, 25k loops, f=1..5 + yaw weights", lambda: graphgen.generate(50000, 25000, odom_f_max=5, apply_yaw_weight=True, seed=9), "-37 %"),
         ("40k keyframes, 40k plain loops", lambda: graphgen.generate(40000, 40000, odom_f_max=2, seed=10, outlier_frac=0.0), "-25 %"),
         ("20k keyframes, 20k loops", lambda: graphgen.generate(20000, 20000, odom_f_max=2, seed=3), "+25 %"),
         ("12k keyframes, 12k loops", lambda: graphgen.generate(12000, 12000, odom_f_max=2, seed=3), "+9 %")]
print("%-48s %-8s  %s" % ("graph", "measured", "levels (nodes: blocks), sparse levels only; total blocks; with the smoothed keyframe transition: the same, and Ps_0 blocks per keyframe"))
for name, make, gain in cases:
    g = make()
    a, _ = sizes(g, False)
    b, ps = sizes(g, True)
    ta, tb = sum(x[1] for x in a[:-1]), sum(x[1] for x in b[:-1])
    print("%-48s %-8s  %s = %d  |  %s = %d (x%.2f), Ps_0 %.2f" % (name, gain, " ".join("%d:%d" % x for x in a[:-1]), ta, " ".join("%d:%d" % x for x in b[:-1]), tb, tb / max(ta, 1), ps / g.n_poses), flush=True)
