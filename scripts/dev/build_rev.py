"""Builds libpgo of a git revision into build/variants/libpgo_<name>.so (sources taken with `git archive`), for bit-for-bit comparisons of a new build against an older one
on the same box: PGO_LIBPGO_OVERRIDE=build/variants/libpgo_<name>.so python -m tests.solve_digest C3 ...
  python scripts/dev/build_rev.py <revision> <name>"""
import os, subprocess, sys, tarfile, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solve_keyframe_pose_graph_amd import _build
rev, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "build", "rev_src", name)
os.makedirs(src, exist_ok=True)
tar = subprocess.check_output(["git", "archive", rev, "solve_keyframe_pose_graph_amd/csrc", "include"], cwd=ROOT)
tarfile.open(fileobj=io.BytesIO(tar)).extractall(src)
csrc = os.path.join(src, "solve_keyframe_pose_graph_amd", "csrc")
out = os.path.join(ROOT, "build", "variants", "libpgo_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on", "-I", os.path.join(src, "include"), "-I", csrc, "-x", "hip"]
                      + [os.path.join(csrc, s) for s in _build.HIP_SOURCES] + ["-o", out, "-ldl"])
print(out)
