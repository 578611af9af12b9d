"""In-tree native builds: libpgo.so (HIP, gfx950) and libpgo_graphgen.so (host).  No JIT cache: the built
.so files live next to the sources so that they travel to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIBPGO = os.path.join(_HERE, "libpgo.so")
LIBGEN = os.path.join(_HERE, "libpgo_graphgen.so")

HIP_SOURCES = ["pgo_kernels.hip", "pgo_solver.hip"]
HIP_HEADERS = ["pgo_internal.hpp", "pgo_device_math.hpp", "pgo_mg_kernels.hpp", "pgo_mg_host.hpp", "pgo_comm_local.hpp"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libpgo cannot be built (there is no CPU fallback)")


def build_graphgen(force=False):
    src = os.path.join(CSRC, "pgo_graphgen.cpp")
    if force or _stale(LIBGEN, [src, os.path.join(INCLUDE, "pgo_graphgen.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE, "-o", LIBGEN, src])
    return LIBGEN


HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on"]
_ROOT = os.path.dirname(_HERE)
OBJ_DIR = os.path.join(_ROOT, "build", "obj")


def source_tree_hash(extra_flags=()):
    """sha256 over the sources libpgo.so is built from (names relative to the repo root, then contents, in a fixed order) and the compiler flags: what pgo_build_info()
    reports for a library built by build_libpgo(), recomputable from a checkout (bench.py prints both)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_HEADERS] + [os.path.join(INCLUDE, "pgo.h")]
    for f in files:
        h.update(os.path.relpath(f, _ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(HIP_FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def compile_libpgo(target, extra_flags=(), verbose=False, obj_dir=None):
    """Reproducible build: every translation unit to an object of its own with a FIXED compilation-unit id (clang otherwise draws a random one per run: two builds of the
    same sources differed in a few hundred bytes), source paths mapped relative to the repo root, no linker build-id — the same sources and flags give the same bytes whatever
    the checkout's path.  The two objects are compiled side by side."""
    obj_dir = obj_dir or OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    sha = source_tree_hash(extra_flags)
    procs, objs = [], []
    for src in HIP_SOURCES:
        stem = os.path.splitext(src)[0]
        obj = os.path.join(obj_dir, stem + ".o")
        cmd = [hipcc_path()] + HIP_FLAGS + list(extra_flags) + ["-ffile-prefix-map=%s=." % _ROOT, "-cuid=" + stem, '-DPGO_SOURCE_SHA256="%s"' % sha,
                                                                "-I", INCLUDE, "-I", CSRC, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", target, "-ldl", "-Wl,--build-id=none"])
    return target


def build_libpgo(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(INCLUDE, "pgo.h")]
    if force or _stale(LIBPGO, deps):
        compile_libpgo(LIBPGO, verbose=verbose)
    return LIBPGO


LIBHOST = os.path.join(_HERE, "libpgo_host.so")


def build_host(force=False):
    """C++ host side above the C-ABI (csrc/host/PoseGraphSLAM.*): plain g++, links libpgo.so next to it."""
    build_libpgo(force)
    srcs = [os.path.join(CSRC, "host", "PoseGraphSLAM.cpp"), os.path.join(CSRC, "host", "GraphFormats.cpp")]
    deps = srcs + [os.path.join(CSRC, "host", "PoseGraphSLAM.hpp"), os.path.join(CSRC, "host", "GraphFormats.hpp"), os.path.join(CSRC, "pgo_device_math.hpp"), os.path.join(INCLUDE, "pgo.h"), LIBPGO]
    if force or _stale(LIBHOST, deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", INCLUDE, "-I", CSRC, "-o", LIBHOST] + srcs +
                              ["-L", _HERE, "-l:libpgo.so", "-Wl,-rpath,$ORIGIN"])
    return LIBHOST


EXAMPLE_REPLAY = os.path.join(os.path.dirname(_HERE), "examples", "replay_session")


def build_examples(force=False):
    """examples/replay_session.cpp: the C++ host classes used directly (what a maintainer's code looks like)."""
    build_host(force)
    src = EXAMPLE_REPLAY + ".cpp"
    if force or _stale(EXAMPLE_REPLAY, [src, LIBHOST, os.path.join(CSRC, "host", "PoseGraphSLAM.hpp"), os.path.join(CSRC, "host", "GraphFormats.hpp")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", INCLUDE, "-I", CSRC, "-o", EXAMPLE_REPLAY, src, "-L", _HERE, "-l:libpgo_host.so", "-l:libpgo.so",
                               "-Wl,-rpath," + _HERE])
    return EXAMPLE_REPLAY


EXAMPLE_RANKS = os.path.join(os.path.dirname(_HERE), "examples", "ranks_in_process")


def build_example_ranks(force=False):
    """examples/ranks_in_process.cpp: N ranks of the sharded solver from one C++ process through the C-ABI alone (in-process communicator)."""
    build_libpgo(force); build_graphgen(force)
    src = EXAMPLE_RANKS + ".cpp"
    if force or _stale(EXAMPLE_RANKS, [src, LIBPGO, LIBGEN, os.path.join(INCLUDE, "pgo.h"), os.path.join(INCLUDE, "pgo_graphgen.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", INCLUDE, "-o", EXAMPLE_RANKS, src, "-L", _HERE, "-l:libpgo.so", "-l:libpgo_graphgen.so", "-Wl,-rpath," + _HERE])
    return EXAMPLE_RANKS


def build_all(force=False):
    build_graphgen(force)
    build_libpgo(force)
    build_host(force)
    build_examples(force)
    build_example_ranks(force)
