"""The hot kernel as a reviewable, versioned text: docs/generated/headline_{generic,steady}.hip is what the library writes for BASELINE's
headline world (1 M entities x 3 components, depth 8) -- the kernel that runs 99.9 % of the GPU time of the bench -- and docs/generated/resources.json what
the gfx950 code objects of that text need (registers, scratch, LDS, occupancy).  A change of the generator shows up as a diff of these files:
regenerate with  python scripts/aot_build.py --write-docs  and commit the result together with the generator change."""
import json
import os

import pytest

import bevy_ggrs_amd as bg
import common as cm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = os.path.join(ROOT, "docs", "generated")


def _headline():
    w = bg.World(1_000_000, max_depth=9, flags=bg.GGRS_WORLD_LAYOUT_ONLY)
    cm.build_particles(w)
    return w


@pytest.mark.parametrize("form", ["generic", "steady"])
def test_committed_text_is_what_the_library_generates(form):
    want = open(os.path.join(DOCS, f"headline_{form}.hip")).read()
    got = _headline().generated_kernel_source(steady=form == "steady")
    if got != want:
        import difflib
        d = list(difflib.unified_diff(want.splitlines(), got.splitlines(), "docs/generated (committed)", "generated now", lineterm="", n=1))
        raise AssertionError("the generated kernel text changed -- review the diff, then `python scripts/aot_build.py --write-docs` and commit:\n" + "\n".join(d[:60]))


def test_committed_resources_keep_the_occupancy():
    r = json.load(open(os.path.join(DOCS, "resources.json")))
    for form in ("generic", "steady"):
        x = r[form]
        assert x["private_segment_fixed_size"] == 0 and x["vgpr_spill_count"] == 0 and x["sgpr_spill_count"] == 0, (form, x)      # no scratch: a spill would add HBM traffic of its own
        assert x["waves_per_simd"] == 8 and x["vgpr_count"] <= 64 and x["max_flat_workgroup_size"] == 256, (form, x)
    assert r["steady"]["vgpr_count"] <= 32 and r["steady"]["sgpr_count"] < r["generic"]["sgpr_count"], r         # the specialised copy drops the walk and the mask tests
    assert r["steady"]["kernarg_segment_size"] < 1024, r                                                         # 560 B of arguments + the hidden ones (the one-size block of rounds 2-4: 2112 B)


def test_aot_names_are_a_function_of_the_text():
    """The file name a shipped code object carries is a hash of target + ABI + text: the same text always asks for the same file, any change asks for another."""
    import ctypes as C
    from bevy_ggrs_amd import _ffi
    def name(s):
        b = C.create_string_buffer(64); assert _ffi.lib.ggrs_hip_aot_object_name(s.encode(), b, 64) == 0; return b.value.decode()
    src = open(os.path.join(DOCS, "headline_steady.hip")).read()
    assert name(src) == name(src) and name(src) != name(src + " ") and name(src).endswith(".hsaco") and len(name(src)) == 38
    idx = os.path.join(ROOT, "bevy_ggrs_amd", "aot", "index.json")
    if os.path.exists(idx):                                             # built by __graft_entry__.build(): the shipped object of the committed text is the one the index names
        assert json.load(open(idx))["headline:1000000:8:steady"] == name(src)
