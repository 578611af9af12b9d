"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pgo.h declares; without a GPU the
product path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from solve_keyframe_pose_graph_amd import _build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgo_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_build.build_libpgo())
    names = header_functions(os.path.join(ROOT, "include", "pgo.h"))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.EXPORTS) == names


def test_library_records_the_hash_of_the_sources_it_was_built_from():
    """pgo_build_info(): the source-tree hash the build recipe compiled in equals the hash recomputed from this checkout (include/pgo.h; the recipe also makes the build
    reproducible byte for byte: fixed compilation-unit ids, paths relative to the repo root, no linker build-id)."""
    capi._lib = None
    in_lib, in_tree = capi.build_info()
    assert len(in_tree) == 64 and in_lib == in_tree, (in_lib, in_tree)


def test_two_builds_of_the_same_sources_are_byte_identical(tmp_path):
    """The build recipe (solve_keyframe_pose_graph_amd/_build.py: one object per translation unit with a FIXED -cuid, -ffile-prefix-map, -Wl,--build-id=none) is reproducible:
    two builds into different directories give the same bytes — and the same bytes as the in-tree libpgo.so, so a sha256 quoted in a bench line or under profiles/ can be
    recomputed from the commit (round-4 verdict: two builds differed in 2 468 bytes — clang's random compilation-unit ids)."""
    import hashlib
    shas = []
    for k in range(2):
        d = tmp_path / ("b%d" % k)
        d.mkdir()
        target = str(d / "libpgo.so")
        _build.compile_libpgo(target, obj_dir=str(d / "obj"))
        shas.append(hashlib.sha256(open(target, "rb").read()).hexdigest())
    assert shas[0] == shas[1]
    assert hashlib.sha256(open(_build.build_libpgo(), "rb").read()).hexdigest() == shas[0]


def test_graphgen_exports_every_declared_symbol():
    lib = C.CDLL(_build.build_graphgen())
    for n in header_functions(os.path.join(ROOT, "include", "pgo_graphgen.h")):
        assert hasattr(lib, n), n


def test_options_defaults_match_reference_ceres_settings():
    o = capi.default_options()
    assert o.max_num_iterations == 10            # reference src/PoseGraphSLAM.cpp:1272
    assert o.initial_trust_region_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.function_tolerance == 1e-6 and o.gradient_tolerance == 1e-10 and o.parameter_tolerance == 1e-8
    assert o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32 and o.jacobi_scaling == 1
    assert C.sizeof(capi.Summary) > 256 * C.sizeof(capi.Iteration)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(capi.PgoError) as e:
        capi.Problem()
    assert e.value.code == -2   # PGO_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "solve_keyframe_pose_graph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "capi.py" and "oracle" not in txt.lower(), os.path.join(dirpath, f)


def test_abi_version_and_the_host_only_entry_points_of_round_6():
    """No GPU needed: the ABI version the ctypes view was written against, the sizes of the structs it mirrors (pgo_sharding_stats has no size query: its layout is pinned here by
    the field the library would write last), and the in-process communicator's group object — created, aborted, destroyed; refused for worlds the kernels cannot take."""
    lib = capi.load()
    assert lib.pgo_abi_version() == capi.ABI_VERSION == 7
    assert C.sizeof(capi.ShardingStats) == 200 and capi.ShardingStats.exchanges_per_bj_iteration.offset == 156 and capi.ShardingStats.bytes_allreduce_replicated_setup.offset == 192
    o = capi.default_options()
    assert o.mg_min_keyframes == 5000 and o.mg_min_keyframes_switchable == 5000 and o.mg_smoothed_fine == -1 and o.mg_dist_min_rows == 8192 and o.mg_fine_filter == 0 and o.mg_dist_setup == 1
    g = capi.local_group_create(4)
    assert g.value
    capi.local_group_abort(g)
    capi.local_group_destroy(g)
    for bad in (0, 17):
        with pytest.raises(capi.PgoError):
            capi.local_group_create(bad)
