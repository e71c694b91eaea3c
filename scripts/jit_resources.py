#!/usr/bin/env python
"""Compile a generated-kernel source (or any hiprtc-style HIP source) for gfx950 with hiprtc -- no GPU needed -- and print the
kernel's register / LDS usage from the code object's metadata.  usage: jit_resources.py file.hip [more.hip ...]"""
import ctypes as C
import re
import subprocess
import sys
import tempfile

rtc = C.CDLL("libhiprtc.so")
OPTS = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-ffp-contract=off", b"-fno-fast-math", b"-fhip-fp32-correctly-rounded-divide-sqrt"]


def build(src: bytes) -> bytes:
    prog = C.c_void_p()
    assert rtc.hiprtcCreateProgram(C.byref(prog), src, b"k.hip", 0, None, None) == 0
    arr = (C.c_char_p * len(OPTS))(*OPTS)
    rc = rtc.hiprtcCompileProgram(prog, len(OPTS), arr)
    if rc != 0:
        n = C.c_size_t(); rtc.hiprtcGetProgramLogSize(prog, C.byref(n)); log = C.create_string_buffer(n.value); rtc.hiprtcGetProgramLog(prog, log)
        raise SystemExit(log.value.decode()[-4000:])
    n = C.c_size_t(); rtc.hiprtcGetCodeSize(prog, C.byref(n)); code = C.create_string_buffer(n.value); rtc.hiprtcGetCode(prog, code)
    return code.raw


for path in sys.argv[1:]:
    co = build(open(path, "rb").read())
    with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
        f.write(co)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size", ".vgpr_spill_count", ".max_flat_workgroup_size")
    print(path, {k: (re.search(re.escape(k) + r":\s*(\d+)", notes) or [None, None])[1] for k in keys}, "hsaco:", f.name)
