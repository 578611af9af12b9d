"""A/B of a kernel variant inside ONE gpurun call (boxes differ by +-4 %): builds build/variants/libpgo_<name>.so with the given -D macros, then runs the measurement
script alternately with the product library and the variant (PGO_LIBPGO_OVERRIDE), `reps` times each, and prints every line the script printed.
  python scripts/dev/ab_variant.py <name> "<-Dmacros>" <reps> -- <script> [args...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solve_keyframe_pose_graph_amd import _build  # noqa: E402

name, macros, reps = sys.argv[1], sys.argv[2].split(), int(sys.argv[3])
cmd = sys.argv[sys.argv.index("--") + 1:]
variant = os.path.join(ROOT, "build", "variants", "libpgo_%s.so" % name)
os.makedirs(os.path.dirname(variant), exist_ok=True)
_build.compile_libpgo(variant, extra_flags=macros, obj_dir=os.path.join(ROOT, "build", "variants", "obj_" + name))
_build.build_libpgo()
for r in range(reps):
    for label, lib in (("product", None), (name, variant)):
        env = dict(os.environ)
        env.pop("PGO_LIBPGO_OVERRIDE", None)
        if lib:
            env["PGO_LIBPGO_OVERRIDE"] = lib
        out = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        for ln in out.stdout.splitlines():
            print("[%s #%d] %s" % (label, r, ln), flush=True)
