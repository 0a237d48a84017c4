"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, its POD structs match the bindings, it fails loudly without a HIP device, and the product
never touches the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "frequensee.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._capi.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libfrequensee.so does not export {s}"
    assert sorted(pkg._capi.EXPORTS) == syms


def test_every_export_is_core_or_extended():
    """include/frequensee.h sorts its entry points into the CORE boundary and the EXTENDED tier: every
    declared symbol appears in exactly one of the two lists of the header's opening comment."""
    src = open(os.path.join(ROOT, "include", "frequensee.h")).read()
    head = src[:src.index("#ifndef FREQUENSEE_H")]
    core = set(re.findall(r"fs_[a-z0-9_]+", head[head.index("CORE:"):head.index("EXTENDED =")]))
    ext = set(re.findall(r"fs_[a-z0-9_]+", head[head.index("EXTENDED:"):head.index("(tests/test_capi_cpu.py")]))
    assert not (core & ext), sorted(core & ext)
    syms = set(header_symbols())
    assert core | ext == syms, (sorted(syms - core - ext), sorted((core | ext) - syms))
    assert 25 <= len(core) <= 35


def test_no_getenv_on_a_per_frame_path():
    """the library reads its FS_* knobs once (context creation, scene commit, first launch of a kernel family): every getenv
    outside fs_context_create / the builders is a function-local static initialiser"""
    csrc = os.path.join(ROOT, "audio-pathtracer_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hip", ".hpp")) or f in ("fs_bvh.cpp",):
            continue
        txt = open(os.path.join(csrc, f)).read()
        if f == "fs_capi_context.cpp":      # everything inside fs_context_create
            body = txt[txt.index("int fs_context_create("):txt.index("const char* fs_context_advice(")]
            assert txt.count("getenv") == body.count("getenv"), f
            continue
        for line in txt.splitlines():
            if "getenv" in line and not line.lstrip().startswith("//"):
                assert "static const" in line or "const char* env = std::getenv(\"FS_RCCL_LIB\")" in line or "const char* e = std::getenv" in line, (f, line.strip())


def test_struct_layouts_match_header(pkg):
    lib = pkg._capi.load()
    assert lib.fs_abi_version() == 5
    p = pkg.default_params()
    c = pkg.default_config()
    assert p.struct_size == C.sizeof(pkg._capi.Params)
    assert c.struct_size == C.sizeof(pkg._capi.Config)
    # defaults are the constants compiled into the reference
    assert (p.num_rays, p.depth, p.russian_roulette) == (2000, 0, 1)
    assert p.rr_prob == pytest.approx(0.9) and p.max_trace_dist == 1e6 and p.surface_offset == pytest.approx(0.1)
    assert p.dist_divisor == 1000.0 and p.min_seg == 1.0 and p.prob_exponent == pytest.approx(0.1)
    assert p.energy_clamp == 1.0 and p.energy_gain == 10.0 and p.sound_speed == 343.0
    assert all(abs(a - 0.05) < 1e-9 for a in p.air_absorption)
    assert (c.num_bands, c.sample_rate, c.num_channels, c.world_size) == (1, 48000, 2, 1)


def test_pipeline_counters_layout_and_null_arguments(pkg):
    """fs_pipeline_counters (round 5): the ctypes mirror has the header's layout — struct_size + reserved + 11 counters of 64 bits —
    and the measurement entry points reject null arguments without a context (no device needed)."""
    lib = pkg._capi.load()
    header = open(os.path.join(ROOT, "include", "frequensee.h")).read()
    body = header[header.index("typedef struct fs_pipeline_counters {"):header.index("} fs_pipeline_counters;")]
    names = re.findall(r"uint64_t ([a-z_, ]+);", body)
    fields = [n.strip() for grp in names for n in grp.split(",")]
    mirror = [k for k, _ in pkg._capi.PipelineCounters._fields_ if k not in ("struct_size", "reserved")]
    assert fields == mirror, (fields, mirror)
    assert C.sizeof(pkg._capi.PipelineCounters) == 8 + 8 * len(fields)
    pc = pkg._capi.PipelineCounters()
    pc.struct_size = C.sizeof(pkg._capi.PipelineCounters)
    assert lib.fs_get_pipeline_counters(None, C.byref(pc)) == pkg._capi.ERR_INVALID_ARGUMENT
    assert lib.fs_get_streams(None, None, None) == pkg._capi.ERR_INVALID_ARGUMENT
    assert lib.fs_comm_info(None, None, None, None) == pkg._capi.ERR_INVALID_ARGUMENT


def test_fails_loudly_without_a_device(pkg):
    """No CPU fallback: without a HIP device the context reports FS_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(pkg.FrequenSeeError) as ei:
        pkg.Context(num_bands=1)
    assert ei.value.code == pkg._capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_invalid_config_rejected(pkg):
    lib = pkg._capi.load()
    h = C.c_void_p()
    bad = pkg.default_config(num_bands=9)
    assert lib.fs_context_create(C.byref(bad), C.byref(h)) == pkg._capi.ERR_INVALID_ARGUMENT
    bad = pkg.default_config(rank=2, world_size=2)
    assert lib.fs_context_create(C.byref(bad), C.byref(h)) == pkg._capi.ERR_INVALID_ARGUMENT
    bad = pkg.default_config()
    bad.struct_size = 4
    assert lib.fs_context_create(C.byref(bad), C.byref(h)) == pkg._capi.ERR_INVALID_ARGUMENT


def test_product_never_references_the_oracle():
    pkg_dir = os.path.join(ROOT, "audio-pathtracer_amd")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for line in txt.splitlines():
                    if re.search(r"^\s*(import|from)\s+oracle|fs_oracle|libfs_oracle|fso_", line):
                        raise AssertionError(f"{f}: product references the oracle: {line.strip()}")


def test_scene_generators_exact_counts(scene_factory):
    assert scene_factory("shoebox").num_triangles == 12
    assert scene_factory("starter_room").num_triangles == 5000
    om = scene_factory("old_mine")
    assert om.num_triangles == 100000 and om.absorption.shape == (8, 8)
    assert om.extra_sources.shape == (8, 3)
    assert 0.05 <= om.absorption.min() and om.absorption.max() <= 0.9


def test_every_export_is_explained_to_the_integrator():
    """INTEGRATION.md's last table names the reference interface every entry point of include/frequensee.h stands for."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "frequensee.h")).read()
    names = set(re.findall(r"^(?:int|void|const char\*) (fs_[a-z_]+)\(", header, re.M))
    assert len(names) >= 60
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc and n.replace("fs_", "fs_").rsplit("_async", 1)[0] not in doc)
    assert not missing, missing
