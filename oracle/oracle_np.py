"""numpy twin of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A second, independent restatement (vectorised numpy, plus a pure-Python big-int SeaHash)
of the arithmetic in oracle/ggrs_oracle.cpp, used only by tests/ to cross-check the C++
oracle before it is trusted as the checker for the HIP path.  Nothing under
bevy_ggrs_amd/ may import this module.

Reference call sites restated here (paths relative to /root/reference):
  * SeaHash 4.1 stream hasher (third-party crate, Cargo.toml:24; not vendored) as used by
    src/snapshot/mod.rs:318-320, component_checksum.rs:67-108, entity_checksum.rs:29-52,
    checksum.rs:38-44,88-99
  * examples/stress_tests/particles.rs:272-289 (update_particles, despawn_particles)
  * src/time.rs:63-87 (Time<GgrsTime> delta)

Parity pinning: seahash's published vector hash("to be or not to be") ==
1988685042348123509; everything else is parity-unpinned by reference-supplied vectors
(see oracle/ggrs_oracle.cpp header).
"""
from __future__ import annotations

import numpy as np

M64 = (1 << 64) - 1
P = 0x6EED0E9DA4D94A4F
K = (0x16F11FE89B0D677C, 0xB480A793D8E6C86C, 0x6FE2E5AAF078EBC9, 0x14F994A4C5259381)


# ----------------------------------------------------------------------------- pure Python
def diffuse(x: int) -> int:
    x = (x * P) & M64
    x ^= (x >> 32) >> (x >> 60)
    return (x * P) & M64


class SeaHasher:
    """seahash::SeaHasher (stream form)."""

    def __init__(self):
        self.s = list(K)
        self.written = 0
        self.tail = b""

    def _push(self, w: int):
        a = diffuse(self.s[0] ^ w)
        self.s = [self.s[1], self.s[2], self.s[3], a]
        self.written += 8

    def write(self, data: bytes):
        buf = self.tail + bytes(data)
        n_full = len(buf) // 8
        for i in range(n_full):
            self._push(int.from_bytes(buf[8 * i:8 * i + 8], "little"))
        self.tail = buf[8 * n_full:]
        return self

    def write_u64(self, v: int):
        return self.write(int(v & M64).to_bytes(8, "little"))

    def write_u32(self, v: int):
        return self.write(int(v & 0xFFFFFFFF).to_bytes(4, "little"))

    def finish(self) -> int:
        nt = len(self.tail)
        a = diffuse(self.s[0] ^ int.from_bytes(self.tail, "little")) if nt else self.s[0]
        return diffuse(a ^ self.s[1] ^ self.s[2] ^ self.s[3] ^ ((self.written + nt) & M64))


def seahash_buffer(data: bytes) -> int:
    """seahash::hash (4-lane buffer form) -- must equal the stream form for every input."""
    a, b, c, d = K
    n = len(data)
    i = 0

    def rd(off, ln):
        return int.from_bytes(data[off:off + ln], "little")

    while n - i >= 32:
        a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8))
        c = diffuse(c ^ rd(i + 16, 8)); d = diffuse(d ^ rd(i + 24, 8))
        i += 32
    rem = n - i
    lanes = [a, b, c, d]
    k = 0
    while rem > 0:
        ln = min(8, rem)
        lanes[k] = diffuse(lanes[k] ^ rd(i, ln))
        i += ln; rem -= ln; k += 1
    a, b, c, d = lanes
    a ^= b; c ^= d; a ^= c; a ^= n
    return diffuse(a)


def entity_checksum(active: int, total: int) -> int:
    """entity_checksum.rs:29-52"""
    return SeaHasher().write_u64(active).write_u64(total).finish()


def checksum_part_from_u32(v: int) -> int:
    """ChecksumPart::from_value(&v: u32), checksum.rs:38-44"""
    return SeaHasher().write_u32(v).finish()


# ----------------------------------------------------------------------------- numpy (vectorised)
_P = np.uint64(P)


def np_diffuse(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x * _P
        x = x ^ ((x >> np.uint64(32)) >> (x >> np.uint64(60)))
        return x * _P


def np_inner_hash_units(units: list[np.ndarray]) -> np.ndarray:
    """custom_hasher(component): SeaHash stream over a list of u32 unit columns."""
    n = len(units)
    s = [np.uint64(k) for k in K]
    written = 0
    k = 0
    while k + 1 < n:
        w = units[k].astype(np.uint64) | (units[k + 1].astype(np.uint64) << np.uint64(32))
        a = np_diffuse(s[0] ^ w)
        s = [s[1], s[2], s[3], a]
        written += 8
        k += 2
    if k < n:
        a = np_diffuse(s[0] ^ units[k].astype(np.uint64))
        nt = 4
    else:
        a = s[0]
        nt = 0
    return np_diffuse(a ^ s[1] ^ s[2] ^ s[3] ^ np.uint64(written + nt))


def np_inner_hash_fields(fields: list[tuple[np.ndarray, int]]) -> np.ndarray:
    """custom_hasher(component) for fields of ANY width: `checksum_hasher()` (snapshot/mod.rs:318-320, a SeaHasher) fed
    `Hash::hash` of each field in turn -- derive(Hash) writes a u8 / bool as 1 byte, a u16 as 2, a u32 / f32::to_bits as 4, a u64 /
    usize as 8, all little-endian.  SeaHasher::write buffers bytes until it has 8, mixes that word into the rotating 4-lane state
    (a, b, c, d) <- (b, c, d, diffuse(a ^ word)), and `finish` mixes what is left of the buffer, then the byte count.  Every entity
    has the same field layout, so the buffer fill level is one Python int and the buffers one uint64 column.
    fields: [(column as unsigned integers, bytes the field contributes)]."""
    n = len(fields[0][0]) if fields else 0
    s = [np.full(n, k, dtype=np.uint64) for k in K]
    tail = np.zeros(n, dtype=np.uint64)
    ntail = 0
    written = 0
    for col, nb in fields:
        v = col.astype(np.uint64)
        if nb < 8:
            v = v & np.uint64((1 << (8 * nb)) - 1)
        tail = tail | (v << np.uint64(8 * ntail)) if ntail < 8 else tail
        tot = ntail + nb
        if tot >= 8:
            a = np_diffuse(s[0] ^ tail)
            s = [s[1], s[2], s[3], a]
            written += 8
            used = 8 - ntail                                   # bytes of this field that completed the word
            tail = (v >> np.uint64(8 * used)) if used < 8 else np.zeros(n, dtype=np.uint64)
            ntail = tot - 8
        else:
            ntail = tot
    a = np_diffuse(s[0] ^ tail) if ntail else s[0]
    return np_diffuse(a ^ s[1] ^ s[2] ^ s[3] ^ np.uint64(written + ntail))


def np_entity_part(order: np.ndarray, inner: np.ndarray) -> np.ndarray:
    s = [np.uint64(k) for k in K]
    b = np_diffuse(s[0] ^ order.astype(np.uint64))
    c = np_diffuse(s[1] ^ inner)
    return np_diffuse(s[2] ^ s[3] ^ b ^ c ^ np.uint64(16))


def np_component_checksum(order: np.ndarray, units: list[np.ndarray]) -> int:
    """component_checksum.rs:67-108 over the given (already filtered) entities."""
    x = 0
    if len(order):
        parts = np_entity_part(order, np_inner_hash_units(units))
        x = int(np.bitwise_xor.reduce(parts))
    return SeaHasher().write_u64(x).finish()


def dt_bits(fps: int, frame: int) -> int:
    """time.rs:63-87 + Duration::as_secs_f32 for the advance INTO `frame`."""
    d = frame * 1_000_000_000 // fps - (frame - 1) * 1_000_000_000 // fps
    secs, nanos = divmod(d, 1_000_000_000)
    r = np.float32(secs) + np.float32(nanos) / np.float32(1_000_000_000)
    return int(np.float32(r).view(np.uint32))


def np_particles_update(tx, ty, tz, vx, vy, vz, dt_bits_: int, g=(0.0, -200.0, 0.0)):
    """particles.rs:272-280 on float32 arrays (in place); numpy never fuses mul+add."""
    dt = np.uint32(dt_bits_).view(np.float32)
    for t, v, gk in ((tx, vx, g[0]), (ty, vy, g[1]), (tz, vz, g[2])):
        v += np.float32(gk) * dt
        t += v * dt


# ----------------------------------------------------------------------------- box_game
def box_move(t, v, inp, dt_bits: int, accel=18.0, max_speed=3.0, friction=0.0018, half_width=None):
    """move_cube_system (examples/box_game/box_game.rs:154-206), vectorised over entities.
    t, v: float32 arrays (n, 3) (translation / Velocity), inp: uint8 array (n,) -- the input byte of the
    entity's player.  Returns new (t, v).  Every numpy float32 op is one IEEE operation (no fusing);
    FRICTION.powf(dt) goes through numpy's float32 power (libm powf)."""
    f = np.float32
    dt = np.array([dt_bits], dtype=np.uint32).view(np.float32)[0]
    if half_width is None:
        half_width = (f(5.0) - f(0.2)) * f(0.5)                      # (PLANE_SIZE - CUBE_SIZE) * 0.5
    accel, max_speed, friction, half_width = f(accel), f(max_speed), f(friction), f(half_width)
    t = np.array(t, dtype=np.float32, copy=True)
    v = np.array(v, dtype=np.float32, copy=True)
    up, down, left, right = [(inp & b) != 0 for b in (1, 2, 4, 8)]
    adt = accel * dt
    fp = np.power(friction, dt, dtype=np.float32)
    vx, vy, vz = v[:, 0], v[:, 1], v[:, 2]
    vz = np.where(up & ~down, vz - adt, vz)
    vz = np.where(~up & down, vz + adt, vz)
    vx = np.where(left & ~right, vx - adt, vx)
    vx = np.where(~left & right, vx + adt, vx)
    vz = np.where(~up & ~down, vz * fp, vz)
    vx = np.where(~left & ~right, vx * fp, vx)
    vy = vy * fp
    len_sq = (vx * vx + vy * vy) + vz * vz
    over = len_sq > max_speed * max_speed
    with np.errstate(divide="ignore", invalid="ignore"):
        l = np.sqrt(len_sq)
        vx = np.where(over, max_speed * (vx / l), vx)
        vy = np.where(over, max_speed * (vy / l), vy)
        vz = np.where(over, max_speed * (vz / l), vz)
    x = t[:, 0] + vx * dt
    y = t[:, 1] + vy * dt
    z = t[:, 2] + vz * dt
    x = np.where(x < -half_width, -half_width, x); x = np.where(x > half_width, half_width, x)
    z = np.where(z < -half_width, -half_width, z); z = np.where(z > half_width, half_width, z)
    return (np.stack([x, y, z], axis=1).astype(np.float32), np.stack([vx, vy, vz], axis=1).astype(np.float32))
