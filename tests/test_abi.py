"""The C-ABI library loads and exports every symbol include/ggrs_hip.h declares (no compute
calls: there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

import bevy_ggrs_amd as bg
from bevy_ggrs_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ggrs_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ggrs_hip_\w+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    syms = _declared_symbols()
    assert len(syms) >= 35
    lib = C.CDLL(_ffi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in ggrs_hip.h but not exported"
        assert s in _ffi.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_ffi.SIGNATURES) == set(syms)


def test_abi_version_and_struct_sizes():
    assert _ffi.lib.ggrs_hip_abi_version() == 9
    assert C.sizeof(_ffi.Request) == 72 and C.sizeof(_ffi.SpawnSystemDesc) == 128
    assert C.sizeof(_ffi.SystemDesc) == 72
    assert C.sizeof(_ffi.WorldDesc) == 48
    assert C.sizeof(_ffi.BranchSpawn) == 40 and C.sizeof(_ffi.BranchStep) == 64


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(bg.GgrsHipError) as e:
        bg.World(128)
    assert e.value.code == bg.GGRS_E_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "bevy_ggrs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the CPU oracle", "").replace("CPU\n oracle", "") or f == "world.py", (dp, f)


def test_rust_ffi_declares_every_header_symbol():
    """rust/bevy_ggrs_hip/src/ffi.rs (un-built source: no Rust toolchain in this image) must stay in lock-step with
    include/ggrs_hip.h: same entry points, same number of arguments, same ABI version and request/system constants."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "ggrs_hip.h")).read()
    rs = open(os.path.join(root, "rust", "bevy_ggrs_hip", "src", "ffi.rs")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    c_fns = {m.group(1): m.group(2) for m in re.finditer(r"\b(ggrs_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr_nc)}
    rs_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (ggrs_hip_[a-z0-9_]+)\(([^)]*)\)", rs)}
    assert set(c_fns) == set(rs_fns), set(c_fns) ^ set(rs_fns)
    def argc(a):
        a = a.strip()
        return 0 if a in ("", "void") else a.count(",") + 1
    for name in c_fns:
        assert argc(c_fns[name]) == argc(rs_fns[name]), name
    for const in re.findall(r"#define (GGRS_(?:E|REQ|SYS|WORLD|COMP|DESPAWN)_[A-Z0-9_]+|GGRS_OK|GGRS_HIP_ABI_VERSION)\s+(-?\d+)u?", hdr):
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % const[0], rs)
        assert m and int(m.group(1)) == int(const[1]), const


_C2RS = {"int": "c_int", "uint64_t": "u64", "uint32_t": "u32", "int32_t": "i32", "int64_t": "i64", "uint8_t": "u8", "uint16_t": "u16", "float": "f32", "double": "f64",
         "char": "c_char", "void": "c_void"}


def _c_type_to_rust(t: str, name_has_array: bool = False) -> str:
    """`const ggrs_request*` -> `*const ggrs_request`, `void**` -> `*mut *mut c_void`, `const void* const*` -> `*const *const c_void`;
    an array parameter (`uint64_t out[2]`) is a pointer to its element."""
    import re
    t = re.sub(r"\s+", " ", t.strip())
    toks = re.findall(r"\*|const|[A-Za-z_]\w*", t)
    base = [x for x in toks if x not in ("*", "const")]
    assert len(base) == 1, t
    out = _C2RS.get(base[0], base[0])
    # walk the declarator left to right: a `const` belongs to what precedes the next `*` (or to the base when it leads)
    pending_const = False
    seen_base = False
    for x in toks:
        if x == "const": pending_const = True
        elif x == "*":
            out = ("*const " if pending_const else "*mut ") + out
            pending_const = False
        else:
            seen_base = True
    if name_has_array:
        out = ("*const " if pending_const else "*mut ") + out
    return out


def _parse_header():
    import re
    hdr = open(os.path.join(ROOT, "include", "ggrs_hip.h")).read()
    h = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))
    fns = {}
    for m in re.finditer(r"^\s*([A-Za-z_][\w\s\*]*?)\b(ggrs_hip_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", h, flags=re.M):
        ret, name, args = m.group(1).strip(), m.group(2), re.sub(r"\s+", " ", m.group(3)).strip()
        params = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                arr = bool(re.search(r"\[[^\]]*\]$", a))
                a = re.sub(r"\[[^\]]*\]$", "", a).strip()
                mm = re.match(r"(.*?)([A-Za-z_]\w*)$", a)
                params.append((mm.group(2), _c_type_to_rust(mm.group(1), arr)))
        fns[name] = (None if ret == "void" else _c_type_to_rust(ret), params)
    structs = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(ggrs_\w+)\s*;", h, flags=re.S):
        fields = []
        for line in m.group(1).split(";"):
            line = re.sub(r"\s+", " ", line).strip()
            if not line: continue
            mm = re.match(r"(.*?)([A-Za-z_]\w*)\s*(\[[^\]]*\])?$", line)
            ty = _c_type_to_rust(mm.group(1))
            if mm.group(3): ty = f"[{ty}; {mm.group(3)[1:-1].strip()}]"
            fields.append((mm.group(2), ty))
        structs[m.group(2)] = fields
    return fns, structs


def _parse_ffi_rs():
    import re
    rs = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "bevy_ggrs_hip", "src", "ffi.rs")).read())
    norm = lambda t: re.sub(r"\s+", " ", t.strip())
    fns = {}
    for m in re.finditer(r"pub fn (ggrs_hip_[a-z0-9_]+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rs):
        params = [(a.split(":", 1)[0].strip(), norm(a.split(":", 1)[1])) for a in m.group(2).split(",") if a.strip()]
        fns[m.group(1)] = (norm(m.group(3)) if m.group(3) else None, params)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (ggrs_\w+)\s*\{(.*?)\}", rs, flags=re.S):
        fields = [(f.split(":", 1)[0].replace("pub", "").strip(), norm(f.split(":", 1)[1])) for f in m.group(2).split(",") if ":" in f]
        structs[m.group(1)] = fields
    return fns, structs


def test_rust_ffi_signatures_and_struct_layouts_match_the_header():
    """Beyond names and argument counts: every parameter's NAME and TYPE, every return type and every field of the
    three argument structs in rust/bevy_ggrs_hip/src/ffi.rs must be what a bindgen run over include/ggrs_hip.h would emit -- checked by
    translating the header's C declarators to Rust (`const ggrs_request*` -> `*const ggrs_request`, `uint64_t out[2]` -> `*mut u64`,
    `uint32_t comp[GGRS_CUSTOM_MAX_BINDINGS]` -> `[u32; GGRS_CUSTOM_MAX_BINDINGS]`) and comparing token for token."""
    c_fns, c_structs = _parse_header()
    r_fns, r_structs = _parse_ffi_rs()
    assert len(c_fns) >= 60 and set(c_fns) == set(r_fns)
    for name, (ret, params) in c_fns.items():
        r_ret, r_params = r_fns[name]
        assert r_ret == ret, f"{name}: returns {r_ret} in ffi.rs, {ret} per the header"
        assert r_params == params, f"{name}: ffi.rs {r_params} vs header {params}"
    for sname in ("ggrs_world_desc", "ggrs_system_desc", "ggrs_custom_system_desc", "ggrs_spawn_system_desc", "ggrs_request", "ggrs_branch_spawn", "ggrs_branch_step"):
        assert sname in c_structs and sname in r_structs, sname
        c = [(n, t.replace("[u32; 4]", "[u32; 4]")) for n, t in c_structs[sname]]
        assert r_structs[sname] == c, f"{sname}: ffi.rs {r_structs[sname]} vs header {c}"
    # the translator itself, on the declarator shapes the header uses
    assert _c_type_to_rust("const void* const*") == "*const *const c_void" and _c_type_to_rust("ggrs_world**") == "*mut *mut ggrs_world"
    assert _c_type_to_rust("const uint8_t", True) == "*const u8" and _c_type_to_rust("uint64_t", True) == "*mut u64" and _c_type_to_rust("const char*") == "*const c_char"


def test_rust_shim_uses_only_declared_symbols_and_owns_the_session_driver():
    """rust/bevy_ggrs_hip/src/lib.rs (un-built source) must (a) use only `ffi::` items that ffi.rs declares and the
    header exports, (b) install its OWN session-driving system -- the stock bevy_ggrs plugin would call the stock
    CPU handle_requests (/root/reference/src/schedule_systems.rs:98,123,156) and never drive the device world --,
    (c) mirror the no-session reset (schedule_systems.rs:70-78) on the device, (d) collect an enqueued batch's
    checksums BEFORE the next advance_frame() (SyncTest compares there), (e) hand spawn payloads and the Spectator
    confirmed-frame rule through."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = open(os.path.join(root, "rust", "bevy_ggrs_hip", "src", "lib.rs")).read()
    rs = open(os.path.join(root, "rust", "bevy_ggrs_hip", "src", "ffi.rs")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "ggrs_hip.h")).read(), flags=re.S)
    code = re.sub(r"//[^\n]*", "", lib)                                   # strip comments: prose may mention anything
    used_fns = set(re.findall(r"ffi::(ggrs_hip_[a-z0-9_]+)", code))
    used_consts = set(re.findall(r"ffi::(GGRS_[A-Z0-9_]+)", code))
    assert len(used_fns) >= 18, used_fns
    for f in used_fns:
        assert re.search(r"pub fn %s\(" % f, rs), f"{f} used by lib.rs but not declared in ffi.rs"
        assert re.search(r"\b%s\s*\(" % f, hdr), f"{f} used by lib.rs but not declared in ggrs_hip.h"
    for c in used_consts:
        assert re.search(r"pub const %s:" % c, rs), f"{c} used by lib.rs but not declared in ffi.rs"
    # (b) own driver, stock plugin NOT added
    assert "add_plugins(bevy_ggrs::GgrsPlugin" not in code
    assert re.search(r"add_systems\(PreUpdate,\s*\(drive_session::<C>", code)
    drive = code[code.index("pub fn drive_session"):code.index("pub fn handle_requests")]
    assert "handle_requests::<C>(reqs, world)" in drive
    # (c) no-session reset reaches the device
    assert "ggrs_hip_set_frame(hip.raw, 0)" in drive and "ggrs_hip_set_confirmed(hip.raw, 1, -1)" in drive and "ggrs_hip_set_depth(hip.raw, 8)" in drive
    # (d) collect precedes the step (advance_frame) inside the loop
    loop = drive[drive.index("while pacer.take_step(fps)"):]
    assert loop.index("collect_in_flight(world)") < loop.index("step_session::<C>")
    # (e) spawn payloads + spectator rule
    hr = code[code.index("pub fn handle_requests"):]
    assert "q.spawn_vx = vx.as_ptr()" in hr and "Kind::Spectator" in hr and "ggrs_hip_set_synctest_check_distance(raw, 0)" in hr


def test_bench_gpus_dry_run_builds_rank_commands():
    """`bench.py --gpus N` without a launcher starts its own ranks: --dry-run prints one JSON line per rank --
    rank r with RANK = LOCAL_RANK = r (bench.py binds device LOCAL_RANK), the same WORLD_SIZE / MASTER_ADDR / MASTER_PORT, the
    same command line -- and under a launcher (WORLD_SIZE set) it does not spawn again."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    ranks = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(ranks) == 2
    for r, d in enumerate(ranks):
        assert d["env"]["RANK"] == d["env"]["LOCAL_RANK"] == str(r) and d["env"]["WORLD_SIZE"] == "2"
        assert d["env"]["MASTER_ADDR"] == "127.0.0.1" and d["env"]["MASTER_PORT"] == ranks[0]["env"]["MASTER_PORT"]
        assert d["cmd"][-4:] == ["--gpus", "2", "--steps", "7"] and d["cmd"][1].endswith("bench.py")
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert len([l for l in single.stdout.splitlines() if l.startswith("{")]) == 1
    launched = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                              env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), timeout=120)
    assert len([l for l in launched.stdout.splitlines() if l.startswith("{")]) == 1       # a rank of an existing launch: no re-spawn


def test_kernel_info_without_a_device():
    """ggrs_hip_world_kernel_info works on a GGRS_WORLD_LAYOUT_ONLY world: the run-time compiler's state is queryable anywhere."""
    w = bg.World(1000, max_depth=4, flags=bg.GGRS_WORLD_LAYOUT_ONLY)
    w.register_component("X", 4, 1)
    info = w.kernel_info()
    assert info["sealed"] == "0" and info["hiprtc"].startswith(("loaded", "missing")) and "request_group_kernel" in info and info["row_versions"] == "on"
    assert info["specialised_kernel"] == "none yet" and w.specialise_wait() is False              # nothing to wait for: no request group has run


def test_every_environment_knob_is_documented():
    """Every GGRS_* variable the library reads (csrc/) has a row in INTEGRATION.md's knob table."""
    import glob, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "bevy_ggrs_amd", "csrc", "*.h*")):
        names |= set(re.findall(r'"(GGRS_[A-Z0-9_]+)"', open(f).read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert 8 <= len(names) <= 15, sorted(names)            # at most 15 environment knobs
    assert not [n for n in sorted(names) if n not in doc]


def test_limits_agree_between_the_header_the_kernels_and_the_oracle():
    """GGRS_MAX_* are spelled once in the header; the device code sizes its mask / checksum tables from the header's constant and the CPU oracle carries its own
    copy of the limits (it links nothing of the product): the three must say the same, or a world at the limit is accepted by one side and refused by the other."""
    import re
    hdr = open(os.path.join(ROOT, "include", "ggrs_hip.h")).read()
    lim = {k: int(v) for k, v in re.findall(r"#define (GGRS_MAX_[A-Z_]+)\s+(\d+)", hdr)}
    assert lim["GGRS_MAX_COMPONENTS"] == 32 and lim["GGRS_MAX_WORDS"] == 16 and lim["GGRS_MAX_PLAYERS"] == 16 and lim["GGRS_MAX_INPUT_BYTES"] == 16
    ker = open(os.path.join(ROOT, "bevy_ggrs_amd", "csrc", "kernels.hpp")).read()
    assert "constexpr int MAX_COMPS = GGRS_MAX_COMPONENTS;" in ker and "constexpr int MAX_MASKS = MAX_COMPS + 1;" in ker
    assert "constexpr int MAX_CKS = MAX_COMPS;" in ker and "constexpr int GEN_MAX_CKS = MAX_COMPS;" in ker
    assert not re.search(r"\b(off_present|n_units|unit_base)\[16\]", ker)
    ora = open(os.path.join(ROOT, "oracle", "ggrs_oracle.cpp")).read()
    m = re.search(r"enum \{ MAX_COMPS = (\d+), MAX_WORDS = (\d+), MAX_UNITS = (\d+), MAX_SYSTEMS = (\d+) \}", ora)
    assert m and [int(x) for x in m.groups()] == [lim["GGRS_MAX_COMPONENTS"], lim["GGRS_MAX_WORDS"], lim["GGRS_MAX_CKS_UNITS"], lim["GGRS_MAX_SYSTEMS"]]
