#!/usr/bin/env python
"""Inputs of the reference fixture run (rust/fixtures): the synthetic velocities / ttls of BASELINE configs 2 and 3 exactly as
tests/common.py generates them (numpy default_rng(123), ttl = 1 + slot % 300) and the scripted input bytes of config 1, as
little-endian binary files the Rust side reads."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
from test_box_game import input_script  # noqa: E402

CONFIGS = {"config2": 10_000, "config3": 1_000_000}


def main():
    out = os.path.join(ROOT, "rust", "fixtures", "inputs")
    os.makedirs(out, exist_ok=True)
    for tag, n in CONFIGS.items():
        vel, ttl = cm.synthetic_particles(n, ttl="despawn")
        vel.astype("<f4").tofile(os.path.join(out, f"{tag}_vel.bin"))
        ttl.astype("<u8").tofile(os.path.join(out, f"{tag}_ttl.bin"))
    np.array([input_script(t, 2) for t in range(40)], dtype=np.uint8).tofile(os.path.join(out, "config1_inputs.bin"))
    print("wrote", out)


if __name__ == "__main__":
    main()
