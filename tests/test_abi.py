"""The C-ABI library loads and exports every symbol include/restir_rt.h (the reference-facing boundary) and
include/restir_rt_internal.h (strip driver / tools / tests) declare (no compute calls: this runs without a GPU), PODs
have the reference's sizes, and the product has no CPU fallback."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"^(?:int|size_t|const char\*)\s+(rt_[a-z_0-9]+)\s*\(", src, re.M)))


def _declared():
    return sorted(_declared_in("restir_rt.h") + _declared_in("restir_rt_internal.h"))


def test_library_exports_every_declared_symbol():
    from cedec_2024_rt_amd import api

    lib = api.load_library()
    public, internal = _declared_in("restir_rt.h"), _declared_in("restir_rt_internal.h")
    assert not set(public) & set(internal)
    for names, mirror in ((public, api.PUBLIC_EXPORTS), (internal, api.INTERNAL_EXPORTS)):
        missing = [n for n in names if not hasattr(lib, n)]
        assert not missing, missing
        assert sorted(mirror) == names
    # the public header is the reference-facing boundary, not the strip driver's and the tools' internals
    assert 30 <= len(public) <= 40, len(public)
    for n in ("rt_create", "rt_scene_set", "rt_camera_lookat", "rt_options_set", "rt_clear", "rt_raycast", "rt_generate_candidate",
              "rt_temporal_resampling", "rt_save_temporal_reservoir", "rt_spatial_resampling", "rt_resolve", "rt_tone_mapping",
              "rt_frame", "rt_mg_create", "rt_mg_frame", "rt_mg_destroy", "rt_upload", "rt_download", "rt_timing", "rt_ray_count"):
        assert n in public, n
    assert not [n for n in public if n.startswith(("rt_halo_", "rt_frame_stage", "rt_exp_", "rt_wire", "rt_tuning", "rt_trace"))]


def test_public_header_stands_alone_and_the_host_app_needs_no_more_for_a_frame():
    """restir_rt.h compiles as C and as C++ on its own; the internal header includes it."""
    import subprocess

    inc = os.path.join(ROOT, "include")
    for lang, cc in (("c", "gcc"), ("c++", "g++")):
        for h in ("restir_rt.h", "restir_rt_internal.h"):
            subprocess.run([cc, "-fsyntax-only", "-Wall", "-Werror", "-x", lang, os.path.join(inc, h)], check=True)


def test_product_library_carries_no_experiment_kernels():
    """the A/B forms (per-lane gather / LDS-staged / software-pipelined spatial pass, fused last pass + resolve,
    resolve as a stream, ray queue, PLOC builder ...) are code objects of librestir_rt_exp.so only; both libraries export the
    same C-ABI."""
    import subprocess

    from cedec_2024_rt_amd import api

    exp = api.load_library(exp=True)
    assert not [n for n in _declared() if not hasattr(exp, n)]
    assert exp.rt_build_id().decode().endswith("-exp") and not api.build_id(exp=False).endswith("-exp")
    A_B_ONLY = ("k_spatial_gather", "k_spatial_lds", "k_spatial_pipe", "k_spatial_resolve", "k_resolve_stream", "k_trace_queue", "k_raycast_half",
                "k_candidate_visibility", "k_ploc_nn", "k_raycast_quad")

    def kernels(path):
        out = subprocess.run(["strings", "-a", path], capture_output=True, text=True, check=True).stdout
        return {k for k in A_B_ONLY if k in out}

    assert kernels(api.LIB_PATH) == set(), "A/B-only kernels in the product library"
    assert kernels(api.EXP_LIB_PATH) == set(A_B_ONLY)


def test_pod_sizes_match_reference():
    from cedec_2024_rt_amd import types

    assert types.TRIANGLE.itemsize == 60 and types.VISIBILITY.itemsize == 16 and types.RESERVOIR.itemsize == 76
    assert types.OPTIONS.itemsize == 48 and types.RAYGEN.itemsize == 36
    assert types.RESERVOIR.fields["w_sum"][1] == 64 and types.RESERVOIR.fields["M"][1] == 72
    assert types.OPTIONS.fields["use_shadowed_target_function"][1] == 44
    d = types.default_options()
    assert d["ris_sample_count"][0] == 32 and d["use_temporal_resampling"][0] == 0 and d["use_visibility_reuse"][0] == 1


def test_no_cpu_fallback_without_gpu():
    import torch

    from cedec_2024_rt_amd import api

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.RtError):
        api.Renderer(16, 16)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "cedec_2024_rt_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "restir_oracle" not in txt and "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
    for f in ("include/restir_rt.h", "include/restir_rt_internal.h"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read().lower().replace("design.md \"oracle\"", "")


def test_build_id_is_the_hash_of_sources_and_flags():
    """rt_build_id (what bench.py matches profiles/spatial_pmc_latest.json on) is reproducible from the checkout: the SHA-256
    the Makefile bakes in, recomputed here from the same files and flags. A clean rebuild of the same sources gives the same id
    (the SHA-256 of the .so does not: VERDICT r02)."""
    import hashlib
    import subprocess

    from cedec_2024_rt_amd import api

    csrc = os.path.join(ROOT, "cedec_2024_rt_amd", "csrc")
    want = subprocess.run(["make", "-s", "-C", csrc, "--eval", "print-id: ; @echo $(BUILD_ID)", "print-id"], capture_output=True, text=True).stdout.strip()
    got = api.build_id(exp=False)
    assert re.fullmatch(r"[0-9a-f]{16}", got), got
    assert got == want, f"the library was built from other sources than the checkout holds ({got} vs {want}): run __graft_entry__.build()"
    # and it is a hash over the listed sources: a change of any of them changes it
    mk = open(os.path.join(csrc, "Makefile")).read()
    listed = re.search(r"^SOURCES = (.*)$", mk, re.M).group(1).split()
    assert {"restir_rt.hip", "frame_kernels.h", "bvh.h", "bvh_build_device.h", "rt_device.h", "portable_math.h", "strip_mg.cpp"} <= set(listed)
    h = hashlib.sha256()
    for f in listed:
        h.update(open(os.path.join(csrc, f), "rb").read())
    assert len(h.hexdigest()) == 64
