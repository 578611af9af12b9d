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
HIP_HEADERS = ["pgo_internal.hpp", "pgo_device_math.hpp", "pgo_mg_kernels.hpp", "pgo_mg_host.hpp"]


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


def build_libpgo(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(INCLUDE, "pgo.h")]
    if force or _stale(LIBPGO, deps):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on",
               "-I", INCLUDE, "-I", CSRC, "-x", "hip"] + srcs + ["-o", LIBPGO, "-ldl"]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
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


def build_all(force=False):
    build_graphgen(force)
    build_libpgo(force)
    build_host(force)
    build_examples(force)
