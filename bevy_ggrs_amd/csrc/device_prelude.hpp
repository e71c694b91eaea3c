// device_prelude.hpp -- device code shared VERBATIM by the statically compiled kernels (kernels.hpp includes it as code)
// and by the kernels generated at run time (ggrs_hip.hip includes it as a string and hands it to hiprtc): one text, so the
// SeaHash arithmetic and the box_game step cannot drift between the two.  The whole body is the argument of
// GGRS_SHARED_CODE(...): comments are fine (they are gone before macro expansion), preprocessor directives are not.
GGRS_SHARED_CODE(
constexpr int LT_SHIFT = 13;   // log2 of the slots of a LAYOUT tile (8 workgroup tiles)
constexpr int LAYOUT_TILE = 1 << LT_SHIFT;

// ------------------------------------------------------------------ SeaHash (seahash 4.1)
// Reference call sites: snapshot/mod.rs:318-320, component_checksum.rs:77-95,
// entity_checksum.rs:35-43.  Arithmetic restated from the crate's published algorithm.
constexpr uint64_t SEA_P = 0x6eed0e9da4d94a4fULL;
constexpr uint64_t SEA_K0 = 0x16f11fe89b0d677cULL, SEA_K1 = 0xb480a793d8e6c86cULL,
                   SEA_K2 = 0x6fe2e5aaf078ebc9ULL, SEA_K3 = 0x14f994a4c5259381ULL;

__host__ __device__ __forceinline__ uint64_t sea_diffuse(uint64_t x) {
    x *= SEA_P;
    // x ^= (x >> 32) >> (x >> 60), spelled in 32-bit terms: the shifted value is the high half, the shift count its top 4 bits, and
    // only the low half changes -- two 32-bit shifts and one 32-bit XOR where the 64-bit form costs a v_lshrrev_b64 per diffuse
    const uint32_t hi = (uint32_t)(x >> 32);
    x ^= (uint64_t)(hi >> (hi >> 28));
    x *= SEA_P;
    return x;
}
// SeaHasher::new(); write_u32(x); write_u32(y); write_u32(z); finish()  (12 bytes: one full
// word + a 4-byte tail) -- particles.rs:107-120 / 207-222.
__host__ __device__ __forceinline__ uint64_t sea_inner3(uint32_t x, uint32_t y, uint32_t z) {
    uint64_t A = sea_diffuse(SEA_K0 ^ ((uint64_t)x | ((uint64_t)y << 32)));
    uint64_t a = sea_diffuse(SEA_K1 ^ (uint64_t)z);
    return sea_diffuse(a ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ULL);
}
// the same with the tail's diffuse handed in: a = sea_diffuse(SEA_K1 ^ z) depends on z only, so a caller that knows z did not
// change since it last hashed this entity (translation.z / velocity.z of a 2-D simulation) reuses it
__host__ __device__ __forceinline__ uint64_t sea_tail3(uint32_t z) { return sea_diffuse(SEA_K1 ^ (uint64_t)z); }
__host__ __device__ __forceinline__ uint64_t sea_inner3_with_tail(uint32_t x, uint32_t y, uint64_t a) {
    uint64_t A = sea_diffuse(SEA_K0 ^ ((uint64_t)x | ((uint64_t)y << 32)));
    return sea_diffuse(a ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ULL);
}
// SeaHasher::new(); write_u64(order); write_u64(inner); finish()  -- component_checksum.rs:81-90
__host__ __device__ __forceinline__ uint64_t sea_pair(uint64_t order, uint64_t inner) {
    uint64_t B = sea_diffuse(SEA_K0 ^ order);
    uint64_t C = sea_diffuse(SEA_K1 ^ inner);
    return sea_diffuse(SEA_K2 ^ SEA_K3 ^ B ^ C ^ 16ULL);
}
// the same with B = diffuse(K0 ^ order) precomputed: it depends on the slot only, so one value serves
// every checksummed component of the entity and every Save of a fused group
__host__ __device__ __forceinline__ uint64_t sea_order_lane(uint64_t order) { return sea_diffuse(SEA_K0 ^ order); }
__host__ __device__ __forceinline__ uint64_t sea_pair_pre(uint64_t B, uint64_t inner) {
    uint64_t C = sea_diffuse(SEA_K1 ^ inner);
    return sea_diffuse(SEA_K2 ^ SEA_K3 ^ B ^ C ^ 16ULL);
}
// SeaHasher::new(); write_u64(x); finish()  -- component_checksum.rs:92-95
__host__ __device__ __forceinline__ uint64_t sea_one(uint64_t x) {
    uint64_t A = sea_diffuse(SEA_K0 ^ x);
    return sea_diffuse(SEA_K1 ^ SEA_K2 ^ SEA_K3 ^ A ^ 8ULL);
}
// SeaHasher as a stream (snapshot/mod.rs:318-320 `checksum_hasher()`): `write` appends the low nb bytes of v, little-endian,
// exactly like Hasher::write(&v.to_le_bytes()[..nb]) -- 8-byte words go through diffuse as they fill, the rest waits in a
// tail buffer.  derive(Hash) writes a u8 / bool as 1 byte, u16 as 2, u32 / f32::to_bits as 4, u64 / usize as 8.
struct SeaStream {
    uint64_t s0 = SEA_K0, s1 = SEA_K1, s2 = SEA_K2, s3 = SEA_K3, written = 0, tail = 0;
    uint32_t ntail = 0;
    __host__ __device__ __forceinline__ void write(uint64_t v, uint32_t nb) {      // 1 <= nb <= 8
        if (nb < 8) v &= (1ULL << (8 * nb)) - 1ULL;
        tail |= v << (8 * ntail);                                    // ntail < 8 always
        const uint32_t tot = ntail + nb;
        if (tot >= 8) {
            const uint64_t a = sea_diffuse(s0 ^ tail);
            s0 = s1; s1 = s2; s2 = s3; s3 = a; written += 8;
            const uint32_t used = 8 - ntail;                         // bytes of v that went into the full word: 1..8
            tail = used >= 8 ? 0ULL : (v >> (8 * used));
            ntail = tot - 8;
        } else ntail = tot;
    }
    __host__ __device__ __forceinline__ void unit(uint32_t u) { write(u, 4); }
    __host__ __device__ __forceinline__ uint64_t finish() const {
        const uint64_t a = ntail ? sea_diffuse(s0 ^ tail) : s0;
        return sea_diffuse(a ^ s1 ^ s2 ^ s3 ^ (written + ntail));
    }
};

struct Header {           // first 256 B of every packed state block
    uint64_t len;         // RollbackOrdered::len -- slots ever spawned
    int32_t frame;        // RollbackFrameCount the block was saved at
    uint32_t pad0;
    uint64_t active;      // live Rollback entities (filled by k_finalize)
    uint64_t checksum[2];
};

// XOR of v over the 64 lanes of the wave, returned wave-uniform.  DPP butterflies inside each row of 16 (quad_perm, row_half_mirror,
// row_mirror), then row_bcast:15 / row_bcast:31 carry the row totals up to lane 63: 6 v_xor_b32_dpp per 32-bit half and one
// readlane -- plain VALU, no LDS round trips (a __shfl_xor ladder is 12 ds_bpermute with ~6 dependent LDS latencies).
__device__ __forceinline__ uint32_t wave_xor32(uint32_t v) {
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);     // quad_perm:[1,0,3,2]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);     // quad_perm:[2,3,0,1]
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);    // row_half_mirror
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);    // row_mirror: every lane holds its row's XOR
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_xor(uint64_t v) {
    return ((uint64_t)wave_xor32((uint32_t)(v >> 32)) << 32) | wave_xor32((uint32_t)v);
}
constexpr uint8_t BOX_INPUT_UP = 1 << 0, BOX_INPUT_DOWN = 1 << 1, BOX_INPUT_LEFT = 1 << 2, BOX_INPUT_RIGHT = 1 << 3;   // box_game.rs:13-16
__device__ __forceinline__ void box_move_math(float& x, float& y, float& z, float& vx, float& vy, float& vz, uint8_t in,
                                              float dt, float fp, float accel, float max_speed, float half_width) {
    const bool up = in & BOX_INPUT_UP, down = in & BOX_INPUT_DOWN, left = in & BOX_INPUT_LEFT, right = in & BOX_INPUT_RIGHT;
    const float adt = __fmul_rn(accel, dt);
    if (up && !down) vz = __fsub_rn(vz, adt);
    if (!up && down) vz = __fadd_rn(vz, adt);
    if (left && !right) vx = __fsub_rn(vx, adt);
    if (!left && right) vx = __fadd_rn(vx, adt);
    if (!up && !down) vz = __fmul_rn(vz, fp);
    if (!left && !right) vx = __fmul_rn(vx, fp);
    vy = __fmul_rn(vy, fp);
    const float len_sq = __fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz));
    if (len_sq > __fmul_rn(max_speed, max_speed)) {
        // NOT __fsqrt_rn: clang's HIP header maps it to __ocml_native_sqrt_f32 (approximate) unless
        // OCML_BASIC_ROUNDED_OPERATIONS is defined; sqrtf is the correctly rounded one (Makefile pins
        // -fhip-fp32-correctly-rounded-divide-sqrt, the default)
        const float l = sqrtf(len_sq);
        vx = __fmul_rn(max_speed, vx / l);
        vy = __fmul_rn(max_speed, vy / l);
        vz = __fmul_rn(max_speed, vz / l);
    }
    x = __fadd_rn(x, __fmul_rn(vx, dt)); y = __fadd_rn(y, __fmul_rn(vy, dt)); z = __fadd_rn(z, __fmul_rn(vz, dt));
    const float lo = -half_width, hi = half_width;
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    if (z < lo) z = lo;
    if (z > hi) z = hi;
}

// ------------------------------------------------------------------ fold-forward (host_groups.hpp)
// One CHUNK of one row of per-workgroup checksum partials -- entries [lo, hi) (entry e at p[e * istride]) of what the PREVIOUS launch of the
// stream left in device memory -- becomes one value in pinned host memory, followed by its tag: XOR for a component's entity hashes (component_checksum.rs:88-89),
// the sum for the live counts (entity_checksum.rs:40).  Run by the first workgroups of the next request-group launch (256 threads), or by
// k_ff_fold when no launch follows.  A chunk is at most 1024 values: 4 eight-byte loads in flight per lane, ONE trip -- the role must not cost
// the tile role registers (the kernel's allocation is the maximum over both: with 16 loads in flight it grew from 28 to 41 VGPRs and the
// 1 M launch from 48.9 to 50.1 us, profiles/r05b).  `cell`: 16 bytes, 16-byte aligned, in pinned host memory -- {value, tag} -- written by ONE store.
constexpr uint32_t FF_CHUNK = 1024;
typedef uint32_t ff_u32x4 __attribute__((ext_vector_type(4)));
// self_seq != 0 (SELF-FOLD: the rows are the running launch's own): an entry is a 16-byte cell {value, self_seq} that its tile workgroup wrote with ONE sc1 store;
// a cell whose tag is not there yet is read again (bounded: *ok = false after 2 s) -- the value travels with its tag, so neither side waits for a store to be
// acknowledged or touches a cache as a whole.
__device__ __forceinline__ void ff_fold_row(const uint64_t* p, uint32_t istride, uint32_t lo, uint32_t hi, bool is_cnt, uint64_t* cell, uint64_t seq, uint64_t self_seq = 0) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    __shared__ unsigned long long ff_acc;
    __shared__ uint32_t ff_bad;
    if (tid == 0) { ff_acc = 0ull; ff_bad = 0u; }
    __syncthreads();
    uint64_t x = 0, sum = 0;
    if (self_seq) {
        const unsigned long long t0 = wall_clock64();
        // The fold workgroups are resident from the launch's start: until the chunk's LAST entry is there -- the tile workgroup dispatched last of the chunk's -- only
        // one lane looks, at that one cell, every ~1.7 us (all lanes polling all cells from the start cost a 4 M launch 80 us: profiles/r06w); then every lane reads its
        // cells, again until each tag is there (dispatch order is not completion order)
        if (tid == 0 && hi > lo) {
            const ff_u32x4* const c = reinterpret_cast<const ff_u32x4*>(p) + (uint64_t)(hi - 1u) * istride;
            for (;;) {
                ff_u32x4 q;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(c) : "memory");
                if ((((uint64_t)q.w << 32) | q.z) == self_seq || wall_clock64() - t0 > 200000000ull) break;
                __builtin_amdgcn_s_sleep(64);
            }
        }
        __syncthreads();
        for (uint32_t i = lo + tid; i < hi; i += 256u) {
            const ff_u32x4* const c = reinterpret_cast<const ff_u32x4*>(p) + (uint64_t)i * istride;
            for (;;) {
                ff_u32x4 q;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(c) : "memory");
                if ((((uint64_t)q.w << 32) | q.z) == self_seq) { const uint64_t v = ((uint64_t)q.y << 32) | q.x; x ^= v; sum += v; break; }
                if (wall_clock64() - t0 > 200000000ull) { ff_bad = 1u; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
    } else {
    constexpr int INFL = 4;
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += (uint32_t)INFL * 256u) {
        uint64_t v[INFL];
_Pragma("unroll")
        for (int u = 0; u < INFL; ++u) { const uint32_t i = i0 + (uint32_t)u * 256u; v[u] = i < hi ? p[(uint64_t)i * istride] : 0ULL; }
_Pragma("unroll")
        for (int u = 0; u < INFL; ++u) { x ^= v[u]; sum += v[u]; }
    }
    }
    if (is_cnt) { if (sum) atomicAdd(&ff_acc, (unsigned long long)sum); }          // workgroup-uniform branch
    else { x = wave_xor(x); if (lane == 0) atomicXor(&ff_acc, (unsigned long long)x); }
    __syncthreads();
    if (tid == 0 && !ff_bad) {                                                       // (a self-fold that timed out publishes nothing: the host reports the missing tags)
        // Value and tag leave as ONE 16-byte system-scope store (global_store_dwordx4 sc0 sc1: write-through to the fabric, no cache maintenance) into one
        // naturally aligned 16-byte cell: one write transaction on the way to host memory, so there is no order between two stores to rely on -- a host that
        // sees the tag sees the value that travelled with it.  (Rounds 4-5 issued two relaxed stores with s_waitcnt vmcnt(0) between them, which is ordered
        // on this PCIe path but not by the HIP memory model.)  NOT a system-scope release either: that is a write-back of the XCD's whole L2 (buffer_wbl2) --
        // issued here by 12 workgroups per XCD while the tile workgroups of the same launch stream their first Save THROUGH that L2, it cost the 1 M launch
        // 2.5 us (48.9 -> 51.4 us, profiles/r05d).
        const uint64_t v = (uint64_t)ff_acc;
        const ff_u32x4 q = {(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)seq, (uint32_t)(seq >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(cell), "v"(q) : "memory");
    }
}
)
