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
    x ^= (x >> 32) >> (x >> 60);
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
// generic stream over n u32 units (all writes on this path are multiples of 4 bytes)
struct SeaStream {
    uint64_t s0 = SEA_K0, s1 = SEA_K1, s2 = SEA_K2, s3 = SEA_K3, written = 0;
    uint32_t lo = 0; bool have_lo = false;
    __host__ __device__ __forceinline__ void unit(uint32_t u) {
        if (!have_lo) { lo = u; have_lo = true; return; }
        uint64_t a = sea_diffuse(s0 ^ ((uint64_t)lo | ((uint64_t)u << 32)));
        s0 = s1; s1 = s2; s2 = s3; s3 = a; written += 8; have_lo = false;
    }
    __host__ __device__ __forceinline__ uint64_t finish() const {
        uint64_t a = have_lo ? sea_diffuse(s0 ^ (uint64_t)lo) : s0;
        return sea_diffuse(a ^ s1 ^ s2 ^ s3 ^ (written + (have_lo ? 4ULL : 0ULL)));
    }
};

struct Header {           // first 256 B of every packed state block
    uint64_t len;         // RollbackOrdered::len -- slots ever spawned
    int32_t frame;        // RollbackFrameCount the block was saved at
    uint32_t pad0;
    uint64_t active;      // live Rollback entities (filled by k_finalize)
    uint64_t checksum[2];
};

__device__ __forceinline__ uint64_t wave_xor(uint64_t v) {
    for (int o = 32; o >= 1; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
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
)
