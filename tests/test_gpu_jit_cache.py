"""The on-disk code-object cache of generated kernels (kernel_gen.hpp: GGRS_JIT_CACHE_DIR): a second PROCESS sealing the same world
loads the module from disk instead of compiling it; a truncated / garbage file is ignored and replaced; `0` turns the cache off."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import bevy_ggrs_amd as bg
import common as cm
w = bg.World(20_000, max_depth=9)
ids = cm.build_particles(w)
vel, ttl = cm.synthetic_particles(5000, ttl="despawn")
t0 = time.perf_counter()
cm.spawn_particles(w, ids, 5000, vel, ttl)                    # seals the world: generates + builds (or loads) its kernel
seal_s = time.perf_counter() - t0
drv = cm.SyncTestDriver(w, 3)
for _ in range(6): drv.tick((0,))
print("RESULT " + json.dumps({{"seal_s": seal_s, "kernel": w.kernel_info()["request_group_kernel"], "cs": [hex(c) for _f, c in drv.all_checksums]}}))
'''


def _child(cache_dir):
    env = dict(os.environ, GGRS_JIT_CACHE_DIR=cache_dir)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=300, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[-1][7:])


def test_second_process_loads_the_code_object_from_disk(tmp_path):
    d = str(tmp_path / "jit")
    first = _child(d)
    files = [f for f in os.listdir(d) if f.endswith(".hsaco")]
    assert first["kernel"].startswith("ggrs_jit_tick") and len(files) >= 1, (first, files)
    stamp = {f: os.path.getmtime(os.path.join(d, f)) for f in files}
    second = _child(d)
    assert second["cs"] == first["cs"]
    assert {f: os.path.getmtime(os.path.join(d, f)) for f in files} == stamp, "a cache hit must not rewrite the file"
    assert second["seal_s"] < first["seal_s"], (first["seal_s"], second["seal_s"])          # no hiprtc compile the second time
    # a corrupted entry is not trusted: the world still seals (recompiles) and the entry is replaced by a good one
    victim = os.path.join(d, files[0])
    open(victim, "wb").write(b"not a code object")
    third = _child(d)
    assert third["cs"] == first["cs"] and os.path.getsize(victim) > 1000


def test_cache_can_be_turned_off(tmp_path):
    out = _child("0")
    assert out["kernel"].startswith("ggrs_jit_tick")
    assert not os.path.exists(os.path.join(os.getcwd(), "0"))
