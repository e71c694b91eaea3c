#!/usr/bin/env python
"""Ahead-of-time code objects of the generated request-group kernel: `make -C bevy_ggrs_amd/csrc aot`.

For every (schema, entities, depth) named on the command line (default: the worlds bench.py measures) a GGRS_WORLD_LAYOUT_ONLY world -- no GPU
needed -- hands out the text of its generic kernel and of the copy specialised for its steady SyncTest tick
(ggrs_hip_generated_kernel_source); hipcc compiles both for gfx950 under the library's floating-point contract, and the objects land in
bevy_ggrs_amd/aot/ under the names the library asks for (ggrs_hip_aot_object_name: a hash of target + ABI + text).  A world whose text hashes
to a shipped object loads it -- before the run-time compiler is even looked for -- so a deployment without libhiprtc.so keeps the fast path
for the shapes it shipped (everything else falls back to one launch per request, as ggrs_hip_world_kernel_info reports).

  --write-docs   also writes docs/generated/<name>_{generic,steady}.hip + resources.json (registers, scratch, LDS, occupancy) for the
                 headline world: the reviewable text of the hot kernel (tests/test_generated_golden.py pins it byte for byte)
"""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "--genco", "--no-gpu-bundle-output",
         "-include", "hip/hip_runtime.h", "-x", "hip"]
DEFAULT = ["headline:1000000:8", "allhot:1000000:8", "full:1000000:8", "headline:10000:8", "headline:100000:8"]


def world(schema, n, depth):
    import bevy_ggrs_amd as bg
    import common as cm
    w = bg.World(n, max_depth=depth + 1, flags=bg.GGRS_WORLD_LAYOUT_ONLY)
    cm.build_particles(w, schema=schema)
    return w


def aot_name(src):
    from bevy_ggrs_amd import _ffi
    buf = C.create_string_buffer(64)
    assert _ffi.lib.ggrs_hip_aot_object_name(src.encode(), buf, 64) == 0
    return buf.value.decode()


def compile_to(src, path):
    with tempfile.NamedTemporaryFile("w", suffix=".hip", delete=False) as f:
        f.write(src)
    try:
        subprocess.check_call([HIPCC] + FLAGS + [f.name, "-o", path + ".tmp"])
        os.replace(path + ".tmp", path)
    finally:
        os.unlink(f.name)


def resources(path):
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True, check=True).stdout
    r = {k[1:]: int(re.search(re.escape(k) + r":\s*(\d+)", notes)[1]) for k in (".vgpr_count", ".sgpr_count", ".agpr_count", ".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count",
                                                                                  ".group_segment_fixed_size", ".kernarg_segment_size", ".max_flat_workgroup_size")}
    gran = (r["vgpr_count"] + 7) // 8 * 8
    r["waves_per_simd"] = min(8, 512 // max(gran, 8))            # gfx950: 512 VGPRs per SIMD lane, allocation granule 8, at most 8 waves per SIMD
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("worlds", nargs="*", default=DEFAULT, help="schema:entities:depth (schema: headline | full | allhot)")
    ap.add_argument("--out", default=os.path.join(ROOT, "bevy_ggrs_amd", "aot"))
    ap.add_argument("--write-docs", action="store_true")
    ap.add_argument("--if-missing", action="store_true", help="compile only the objects that are not there yet (what __graft_entry__.build() asks for)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    index = {}
    for spec in a.worlds:
        schema, n, depth = spec.split(":"); n, depth = int(n), int(depth)
        w = world(schema, n, depth)
        for form, steady in (("generic", False), ("steady", True)):
            src = w.generated_kernel_source(steady=steady)
            name = aot_name(src)
            path = os.path.join(a.out, name)
            if not (a.if_missing and os.path.exists(path)):
                compile_to(src, path)
            index[f"{spec}:{form}"] = name
            if a.write_docs and spec == "headline:1000000:8":
                d = os.path.join(ROOT, "docs", "generated"); os.makedirs(d, exist_ok=True)
                open(os.path.join(d, f"headline_{form}.hip"), "w").write(src)
                res = json.load(open(os.path.join(d, "resources.json"))) if os.path.exists(os.path.join(d, "resources.json")) else {}
                res[form] = resources(path)
                json.dump(res, open(os.path.join(d, "resources.json"), "w"), indent=1, sort_keys=True)
        w.close()
    json.dump(index, open(os.path.join(a.out, "index.json"), "w"), indent=1, sort_keys=True)
    for f in os.listdir(a.out):                                       # objects of earlier generator versions: nothing asks for their names any more
        if f.endswith(".hsaco") and f not in index.values(): os.remove(os.path.join(a.out, f))
    print(f"{len(index)} code objects under {a.out}")


if __name__ == "__main__":
    main()
