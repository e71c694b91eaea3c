"""SeaHash known answers: pins the oracle's hash arithmetic (third-party seahash 4.1, not under
/root/reference) before the oracle is trusted as the checker."""
import ctypes as C

import numpy as np

from oracle import oracle_np as onp
from oracle.binding import lib

PUBLISHED = (b"to be or not to be", 1988685042348123509)   # seahash crate docs


def _cxx_stream(data: bytes, chunks=None):
    if chunks is None:
        return lib.gor_seahash_stream(data, len(data), None, 0)
    arr = (C.c_uint32 * len(chunks))(*chunks)
    return lib.gor_seahash_stream(data, len(data), arr, len(chunks))


def test_published_vector_all_formulations():
    data, want = PUBLISHED
    assert lib.gor_seahash_buffer(data, len(data)) == want
    assert _cxx_stream(data) == want
    assert onp.seahash_buffer(data) == want
    assert onp.SeaHasher().write(data).finish() == want


def test_stream_equals_buffer_for_every_length_and_chunking():
    rng = np.random.default_rng(0)
    for n in list(range(0, 100)) + [127, 128, 129, 1000]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = onp.seahash_buffer(data)
        assert lib.gor_seahash_buffer(data, n) == want
        assert _cxx_stream(data) == want
        # random chunkings (write_u8 / write_u32 / write_u64 mixes)
        for _ in range(3):
            chunks, left = [], n
            while left:
                c = int(min(left, rng.choice([1, 2, 3, 4, 4, 8, 8, 13])))
                chunks.append(c); left -= c
            assert _cxx_stream(data, chunks) == want
            h = onp.SeaHasher(); off = 0
            for c in chunks:
                h.write(data[off:off + c]); off += c
            assert h.finish() == want


def test_derived_vectors_frozen():
    # derived from the restated algorithm, frozen here
    assert onp.entity_checksum(1, 1) == 0x7C846906B6E5A068 == lib.gor_entity_checksum(1, 1)
    assert onp.checksum_part_from_u32(42) == 0x352173BD5A4BA44B
    # ChecksumPart::from_value(&42u32) determinism (checksum.rs:109-113)
    assert onp.checksum_part_from_u32(42) == onp.checksum_part_from_u32(42)


def test_component_checksum_formula_cxx_vs_numpy():
    """component_checksum.rs:77-95 shapes: inner over u32 units, pair(order, inner), finalize."""
    rng = np.random.default_rng(1)
    for n_units in (1, 2, 3, 4, 5, 6, 10):
        units = [rng.integers(0, 2**32, 257, dtype=np.uint64).astype(np.uint32) for _ in range(n_units)]
        inner = onp.np_inner_hash_units(units)
        for i in (0, 1, 100, 256):
            u = (C.c_uint32 * n_units)(*[int(c[i]) for c in units])
            assert lib.gor_inner_hash_units(u, n_units) == int(inner[i])
            h = onp.SeaHasher()
            for c in units: h.write_u32(int(c[i]))
            assert h.finish() == int(inner[i])
        order = np.arange(257, dtype=np.uint64)
        parts = onp.np_entity_part(order, inner)
        for i in (0, 7, 256):
            assert lib.gor_entity_part(i, int(inner[i])) == int(parts[i])
            assert onp.SeaHasher().write_u64(i).write_u64(int(inner[i])).finish() == int(parts[i])
    assert lib.gor_finalize_part(0) == onp.SeaHasher().write_u64(0).finish()


def test_dt_bits_rule():
    """time.rs:63-87: ns deltas 666/667/667 at 60 fps -> two f32 values."""
    seen = set()
    for f in range(1, 400):
        b = lib.gor_dt_bits(60, f)
        assert b == onp.dt_bits(60, f)
        seen.add(b)
    assert seen == {0x3C888888, 0x3C888889}
    assert [lib.gor_dt_bits(60, f) for f in (1, 2, 3, 4)] == [0x3C888888, 0x3C888889, 0x3C888889, 0x3C888888]


def test_field_hasher_of_any_width_is_the_byte_stream_hasher():
    """oracle_np.np_inner_hash_fields (vectorised: u8 / u16 / u32 / u64 fields, one fill level for all entities) against the scalar
    SeaHasher fed the same little-endian bytes one field at a time, over random field layouts -- and against the u32-unit hasher."""
    import numpy as np
    from oracle import oracle_np as onp
    rng = np.random.default_rng(1)
    n = 129
    for trial in range(60):
        widths = [int(x) for x in rng.choice([1, 2, 4, 8], size=rng.integers(1, 8))]
        cols = [rng.integers(0, 2 ** (8 * w) if w < 8 else 2 ** 63, n, dtype=np.uint64) for w in widths]
        got = onp.np_inner_hash_fields(list(zip(cols, widths)))
        for e in (0, 17, n - 1):
            h = onp.SeaHasher()
            for c, w in zip(cols, widths): h.write(int(c[e]).to_bytes(8, "little")[:w])
            assert int(got[e]) == h.finish(), (widths, e)
    u = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32) for _ in range(5)]
    assert np.array_equal(onp.np_inner_hash_units(u), onp.np_inner_hash_fields([(x, 4) for x in u]))
