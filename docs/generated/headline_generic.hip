typedef unsigned long uint64_t; typedef unsigned int uint32_t; typedef unsigned short uint16_t; typedef unsigned char uint8_t;
typedef long int64_t; typedef int int32_t;
typedef unsigned long long ggrs_u64; typedef unsigned int ggrs_u32;
#define GGRS_G __attribute__((address_space(1)))
// a wave-uniform pointer pinned into an SGPR pair: `sgpr_base(p) + lane_offset_u32` selects the saddr form of
// global_load / global_store (no 64-bit VALU address arithmetic, no 64-bit address registers per word)
__device__ __forceinline__ GGRS_G unsigned char* sgpr_base(const unsigned char* p) { unsigned long x = (unsigned long)p; asm volatile("" : "+s"(x)); return (GGRS_G unsigned char*)x; }
// stores of one word to `base + lo` (base wave-uniform in an SGPR pair, lo a 32-bit lane offset), written as inline asm: the
// compiler otherwise materialises a 64-bit VGPR address per store -- into ONE register pair it recomputes before every store,
// which serialises a snapshot's store burst behind VALU address arithmetic (two extra VALU ops per stored word)
#define GGRS_ST(NAME, INSN, T, C) __device__ __forceinline__ void NAME(const unsigned char* base, uint32_t lo, T v) { const unsigned long b = (unsigned long)base; asm volatile(INSN " %0, %1, %2" : : "v"(lo), C(v), "s"(b) : "memory"); }
GGRS_ST(st1, "global_store_byte", uint32_t, "v") GGRS_ST(st2, "global_store_short", uint32_t, "v") GGRS_ST(st4, "global_store_dword", uint32_t, "v") GGRS_ST(st8, "global_store_dwordx2", uint64_t, "v")
#undef GGRS_ST
#define GGRS_ST(NAME, INSN, T, C) __device__ __forceinline__ void NAME(const unsigned char* base, uint32_t lo, T v) { const unsigned long b = (unsigned long)base; asm volatile(INSN " %0, %1, %2 nt" : : "v"(lo), C(v), "s"(b) : "memory"); }
GGRS_ST(st1nt, "global_store_byte", uint32_t, "v") GGRS_ST(st2nt, "global_store_short", uint32_t, "v") GGRS_ST(st4nt, "global_store_dword", uint32_t, "v") GGRS_ST(st8nt, "global_store_dwordx2", uint64_t, "v")
#undef GGRS_ST
// a batch member's record (GgrsJitArgs::mtab): written by the host before the launch, never by a kernel -- read through the constant address space,
// i.e. with scalar loads (the record's address is wave-uniform: blockIdx.z)
#define GGRS_K __attribute__((address_space(4)))
__device__ __forceinline__ uint64_t mb_u64(const GGRS_K unsigned char* mb, uint32_t off) { return *(const GGRS_K uint64_t*)(mb + off); }
__device__ __forceinline__ uint32_t mb_u32(const GGRS_K unsigned char* mb, uint32_t off) { return *(const GGRS_K uint32_t*)(mb + off); }
__device__ __forceinline__ uint32_t mb_u8(const GGRS_K unsigned char* mb, uint32_t off) { return *(const GGRS_K unsigned char*)(mb + off); }
// value tags keep one 32-bit tag per COLUMN in lane `column` of a register: a wave-uniform 64-bit column mask therefore IS the set of lanes to touch.  These
// run one instruction under that mask (exec narrowed, the instruction, exec restored) instead of building a per-lane condition from the mask with VALU
// shifts and compares -- the generated kernel is bound by its vector ALUs.  (`s_and_b64` writes SCC and the statements SAY so: without the clobber the compiler
// kept a condition in SCC across them -- s_bitcmp1 before the asm, s_cselect behind it -- in copies specialised for some shapes: profiles/r06ff)
__device__ __forceinline__ uint64_t uni64(uint64_t m) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m); }   // wave-uniform by construction: say so
__device__ __forceinline__ void set_lanes(uint32_t& v, uint64_t lanes_, uint32_t x_) { const uint64_t lanes = uni64(lanes_); const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x_); uint64_t sv_; asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %2\n\tv_mov_b32 %1, %3\n\ts_mov_b64 exec, %0" : "=&s"(sv_), "+v"(v) : "s"(lanes), "s"(x) : "scc"); }
__device__ __forceinline__ void store_lanes(GGRS_G uint32_t* p, uint32_t v, uint64_t lanes_) { const uint64_t lanes = uni64(lanes_); uint64_t sv_; asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(sv_) : "v"(p), "v"(v), "s"(lanes) : "memory", "scc"); }
// SPAWNS DECIDED ON THE DEVICE: the workgroups of a COOPERATIVE launch (all resident) meet through mailbox words {epoch:32 | value:32}, written and polled
// as relaxed agent-scope atomics (sc1: through to where every XCD reads them).  The value travels INSIDE the word it is waited on, so no rendezvous needs a
// release/acquire pair -- on gfx950 those are a writeback / an invalidate of a whole L2 each (measured: ~100 us per barrier with an acquire in the poll loop).
// Bounded: a second of wall clock, then the launch reports an error instead of hanging the device
__device__ __forceinline__ void sp_post(ggrs_u64* p, ggrs_u32 ep, ggrs_u32 v) { __hip_atomic_store(p, ((ggrs_u64)ep << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool sp_await(ggrs_u64* p, ggrs_u32 ep, ggrs_u32& v, unsigned long long t0_) {
    for (;;) {
        const ggrs_u64 x_ = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((ggrs_u32)(x_ >> 32) == ep) { v = (ggrs_u32)x_; return true; }
        if (wall_clock64() - t0_ > 100000000ull) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
namespace ggrs {
constexpr int LT_SHIFT = 13; constexpr int LAYOUT_TILE = 1 << LT_SHIFT; constexpr uint64_t SEA_P = 0x6eed0e9da4d94a4fULL; constexpr uint64_t SEA_K0 = 0x16f11fe89b0d677cULL, SEA_K1 = 0xb480a793d8e6c86cULL, SEA_K2 = 0x6fe2e5aaf078ebc9ULL, SEA_K3 = 0x14f994a4c5259381ULL; __host__ __device__ __forceinline__ uint64_t sea_diffuse(uint64_t x) { x *= SEA_P; const uint32_t hi = (uint32_t)(x >> 32); x ^= (uint64_t)(hi >> (hi >> 28)); x *= SEA_P; return x; } __host__ __device__ __forceinline__ uint64_t sea_inner3(uint32_t x, uint32_t y, uint32_t z) { uint64_t A = sea_diffuse(SEA_K0 ^ ((uint64_t)x | ((uint64_t)y << 32))); uint64_t a = sea_diffuse(SEA_K1 ^ (uint64_t)z); return sea_diffuse(a ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ULL); } __host__ __device__ __forceinline__ uint64_t sea_tail3(uint32_t z) { return sea_diffuse(SEA_K1 ^ (uint64_t)z); } __host__ __device__ __forceinline__ uint64_t sea_inner3_with_tail(uint32_t x, uint32_t y, uint64_t a) { uint64_t A = sea_diffuse(SEA_K0 ^ ((uint64_t)x | ((uint64_t)y << 32))); return sea_diffuse(a ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ULL); } __host__ __device__ __forceinline__ uint64_t sea_pair(uint64_t order, uint64_t inner) { uint64_t B = sea_diffuse(SEA_K0 ^ order); uint64_t C = sea_diffuse(SEA_K1 ^ inner); return sea_diffuse(SEA_K2 ^ SEA_K3 ^ B ^ C ^ 16ULL); } __host__ __device__ __forceinline__ uint64_t sea_order_lane(uint64_t order) { return sea_diffuse(SEA_K0 ^ order); } __host__ __device__ __forceinline__ uint64_t sea_pair_pre(uint64_t B, uint64_t inner) { uint64_t C = sea_diffuse(SEA_K1 ^ inner); return sea_diffuse(SEA_K2 ^ SEA_K3 ^ B ^ C ^ 16ULL); } __host__ __device__ __forceinline__ uint64_t sea_one(uint64_t x) { uint64_t A = sea_diffuse(SEA_K0 ^ x); return sea_diffuse(SEA_K1 ^ SEA_K2 ^ SEA_K3 ^ A ^ 8ULL); } struct SeaStream { uint64_t s0 = SEA_K0, s1 = SEA_K1, s2 = SEA_K2, s3 = SEA_K3, written = 0, tail = 0; uint32_t ntail = 0; __host__ __device__ __forceinline__ void write(uint64_t v, uint32_t nb) { if (nb < 8) v &= (1ULL << (8 * nb)) - 1ULL; tail |= v << (8 * ntail); const uint32_t tot = ntail + nb; if (tot >= 8) { const uint64_t a = sea_diffuse(s0 ^ tail); s0 = s1; s1 = s2; s2 = s3; s3 = a; written += 8; const uint32_t used = 8 - ntail; tail = used >= 8 ? 0ULL : (v >> (8 * used)); ntail = tot - 8; } else ntail = tot; } __host__ __device__ __forceinline__ void unit(uint32_t u) { write(u, 4); } __host__ __device__ __forceinline__ uint64_t finish() const { const uint64_t a = ntail ? sea_diffuse(s0 ^ tail) : s0; return sea_diffuse(a ^ s1 ^ s2 ^ s3 ^ (written + ntail)); } }; struct Header { uint64_t len; int32_t frame; uint32_t pad0; uint64_t active; uint64_t checksum[2]; }; __device__ __forceinline__ uint32_t wave_xor32(uint32_t v) { v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false); v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); } __device__ __forceinline__ uint64_t wave_xor(uint64_t v) { return ((uint64_t)wave_xor32((uint32_t)(v >> 32)) << 32) | wave_xor32((uint32_t)v); } constexpr uint8_t BOX_INPUT_UP = 1 << 0, BOX_INPUT_DOWN = 1 << 1, BOX_INPUT_LEFT = 1 << 2, BOX_INPUT_RIGHT = 1 << 3; __device__ __forceinline__ void box_move_math(float& x, float& y, float& z, float& vx, float& vy, float& vz, uint8_t in, float dt, float fp, float accel, float max_speed, float half_width) { const bool up = in & BOX_INPUT_UP, down = in & BOX_INPUT_DOWN, left = in & BOX_INPUT_LEFT, right = in & BOX_INPUT_RIGHT; const float adt = __fmul_rn(accel, dt); if (up && !down) vz = __fsub_rn(vz, adt); if (!up && down) vz = __fadd_rn(vz, adt); if (left && !right) vx = __fsub_rn(vx, adt); if (!left && right) vx = __fadd_rn(vx, adt); if (!up && !down) vz = __fmul_rn(vz, fp); if (!left && !right) vx = __fmul_rn(vx, fp); vy = __fmul_rn(vy, fp); const float len_sq = __fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)); if (len_sq > __fmul_rn(max_speed, max_speed)) { const float l = sqrtf(len_sq); vx = __fmul_rn(max_speed, vx / l); vy = __fmul_rn(max_speed, vy / l); vz = __fmul_rn(max_speed, vz / l); } x = __fadd_rn(x, __fmul_rn(vx, dt)); y = __fadd_rn(y, __fmul_rn(vy, dt)); z = __fadd_rn(z, __fmul_rn(vz, dt)); const float lo = -half_width, hi = half_width; if (x < lo) x = lo; if (x > hi) x = hi; if (z < lo) z = lo; if (z > hi) z = hi; } constexpr uint32_t FF_CHUNK = 1024; typedef uint32_t ff_u32x4 __attribute__((ext_vector_type(4))); __device__ __forceinline__ void ff_fold_row(const uint64_t* p, uint32_t istride, uint32_t lo, uint32_t hi, bool is_cnt, uint64_t* cell, uint64_t seq, uint64_t self_seq = 0) { const uint32_t tid = threadIdx.x, lane = tid & 63u; __shared__ unsigned long long ff_acc; __shared__ uint32_t ff_bad; if (tid == 0) { ff_acc = 0ull; ff_bad = 0u; } __syncthreads(); uint64_t x = 0, sum = 0; if (self_seq) { const unsigned long long t0 = wall_clock64(); if (tid == 0 && hi > lo) { const ff_u32x4* const c = reinterpret_cast<const ff_u32x4*>(p) + (uint64_t)(hi - 1u) * istride; for (;;) { ff_u32x4 q; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(c) : "memory"); if ((((uint64_t)q.w << 32) | q.z) == self_seq || wall_clock64() - t0 > 200000000ull) break; __builtin_amdgcn_s_sleep(64); } } __syncthreads(); for (uint32_t i = lo + tid; i < hi; i += 256u) { const ff_u32x4* const c = reinterpret_cast<const ff_u32x4*>(p) + (uint64_t)i * istride; for (;;) { ff_u32x4 q; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(q) : "v"(c) : "memory"); if ((((uint64_t)q.w << 32) | q.z) == self_seq) { const uint64_t v = ((uint64_t)q.y << 32) | q.x; x ^= v; sum += v; break; } if (wall_clock64() - t0 > 200000000ull) { ff_bad = 1u; break; } __builtin_amdgcn_s_sleep(8); } } } else { constexpr int INFL = 4; for (uint32_t i0 = lo + tid; i0 < hi; i0 += (uint32_t)INFL * 256u) { uint64_t v[INFL]; _Pragma("unroll") for (int u = 0; u < INFL; ++u) { const uint32_t i = i0 + (uint32_t)u * 256u; v[u] = i < hi ? p[(uint64_t)i * istride] : 0ULL; } _Pragma("unroll") for (int u = 0; u < INFL; ++u) { x ^= v[u]; sum += v[u]; } } } if (is_cnt) { if (sum) atomicAdd(&ff_acc, (unsigned long long)sum); } else { x = wave_xor(x); if (lane == 0) atomicXor(&ff_acc, (unsigned long long)x); } __syncthreads(); if (tid == 0 && !ff_bad) { const uint64_t v = (uint64_t)ff_acc; const ff_u32x4 q = {(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)seq, (uint32_t)(seq >> 32)}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(cell), "v"(q) : "memory"); } }
}
using namespace ggrs;
#define GGRS_INPUT_CONFIRMED 0
#define GGRS_INPUT_PREDICTED 1
#define GGRS_INPUT_DISCONNECTED 2
struct GgrsInputs { const unsigned char* p; ggrs_u32 ib; __device__ unsigned char operator[](int h) const { return p[(ggrs_u32)h * ib]; } };
struct GgrsFrame {
    float dt; int frame; ggrs_u32 n_inputs, input_bytes;
    GgrsInputs input; const unsigned char* status;
    float fparam[4]; long long iparam[2];
    __device__ const unsigned char* input_ptr(int h) const { return input.p + (ggrs_u32)h * input_bytes; }
    __device__ unsigned char input_u8(int h) const { return input_ptr(h)[0]; }
    __device__ unsigned short input_u16(int h) const { const unsigned char* q = input_ptr(h); return (unsigned short)(q[0] | (q[1] << 8)); }
    __device__ ggrs_u32 input_u32(int h) const { const unsigned char* q = input_ptr(h); return (ggrs_u32)q[0] | ((ggrs_u32)q[1] << 8) | ((ggrs_u32)q[2] << 16) | ((ggrs_u32)q[3] << 24); }
    __device__ ggrs_u64 input_u64(int h) const { const unsigned char* q = input_ptr(h); ggrs_u64 v = 0; for (int b = 0; b < 8; ++b) v |= (ggrs_u64)q[b] << (8 * b); return v; }
    __device__ int input_status(int h) const { return status[h]; }
};
struct GgrsEntity {
    ggrs_u64 slot; ggrs_u64 w[8]; int kill; int spawn_n;
    __device__ float& f32(int i) { return *reinterpret_cast<float*>(&w[i]); }
    __device__ ggrs_u32& u32(int i) { return *reinterpret_cast<ggrs_u32*>(&w[i]); }
    __device__ int& i32(int i) { return *reinterpret_cast<int*>(&w[i]); }
    __device__ ggrs_u64& u64(int i) { return w[i]; }
    __device__ unsigned short& u16(int i) { return *reinterpret_cast<unsigned short*>(&w[i]); }
    __device__ unsigned char& u8(int i) { return *reinterpret_cast<unsigned char*>(&w[i]); }
    __device__ void despawn() { if (kill == 0) kill = 1; }
    __device__ void despawn_rollback() { kill = 2; }
    __device__ void spawn(int n) { spawn_n = n < 0 ? 0 : (n > 255 ? 255 : n); }   /* commands.spawn(..) x n from THIS entity's system call: see ggrs_hip_add_spawn_system, GGRS_SPAWN_PAYLOAD_PARENT */
};
struct GgrsComponent {
    ggrs_u64 slot; ggrs_u64 w[16];
    __device__ float f32(int i) const { return __uint_as_float((ggrs_u32)w[i]); }
    __device__ ggrs_u32 u32(int i) const { return (ggrs_u32)w[i]; }
    __device__ int i32(int i) const { return (int)(ggrs_u32)w[i]; }
    __device__ ggrs_u64 u64(int i) const { return w[i]; }
    __device__ unsigned short u16(int i) const { return (unsigned short)w[i]; }
    __device__ unsigned char u8(int i) const { return (unsigned char)w[i]; }
};
struct GgrsHasher {                                   // SeaHasher::new() + Hasher::write_*; finish()
    ggrs::SeaStream s;
    __device__ void write_u8(unsigned char v) { s.write(v, 1); }
    __device__ void write_u16(unsigned short v) { s.write(v, 2); }
    __device__ void write_u32(ggrs_u32 v) { s.write(v, 4); }
    __device__ void write_i32(int v) { s.write((ggrs_u32)v, 4); }
    __device__ void write_u64(ggrs_u64 v) { s.write(v, 8); }
    __device__ void write_usize(ggrs_u64 v) { s.write(v, 8); }
    __device__ void write_f32_bits(float v) { s.write(__float_as_uint(v), 4); }
    __device__ ggrs_u64 finish() const { return s.finish(); }
};
struct GgrsWords {
    ggrs_u64 w[16];
    __device__ float& f32(int i) { return *reinterpret_cast<float*>(&w[i]); }
    __device__ ggrs_u32& u32(int i) { return *reinterpret_cast<ggrs_u32*>(&w[i]); }
    __device__ int& i32(int i) { return *reinterpret_cast<int*>(&w[i]); }
    __device__ ggrs_u64& u64(int i) { return w[i]; }
    __device__ unsigned short& u16(int i) { return *reinterpret_cast<unsigned short*>(&w[i]); }
    __device__ unsigned char& u8(int i) { return *reinterpret_cast<unsigned char*>(&w[i]); }
    __device__ float f32(int i) const { return __uint_as_float((ggrs_u32)w[i]); }
    __device__ ggrs_u32 u32(int i) const { return (ggrs_u32)w[i]; }
    __device__ int i32(int i) const { return (int)(ggrs_u32)w[i]; }
    __device__ ggrs_u64 u64(int i) const { return w[i]; }
    __device__ unsigned short u16(int i) const { return (unsigned short)w[i]; }
    __device__ unsigned char u8(int i) const { return (unsigned char)w[i]; }
};
struct GgrsJitArgs {
    const unsigned char* src;
    unsigned char* live;
    const unsigned char* mtab;
    ggrs_u64* parts;
    const ggrs_u64* ff_rows;
    ggrs_u64* ff_out;
    ggrs_u64 ff_seq;
    ggrs_u64 live_rows;
    ggrs_u64 load_rows;
    ggrs_u64 op_bits;
    ggrs_u64 len;
    unsigned char* save_dst[10];
    ggrs_u64 save_rows[10];
    ggrs_u64 save_len[10];
    int save_frame[10];
    ggrs_u32 save_pmask[10];
    ggrs_u32 live_pmask;
    ggrs_u32 nt_loads;
    ggrs_u32 n_ops;
    ggrs_u32 n_saves;
    ggrs_u32 n_steps;
    ggrs_u32 src_is_live;
    ggrs_u32 skip_live;
    ggrs_u32 dp_s;
    ggrs_u32 part_stride;
    ggrs_u32 part_tstride;
    ggrs_u32 nt;
    ggrs_u32 n_units;
    ggrs_u32 cached_saves;
    ggrs_u32 ff_blocks;
    ggrs_u32 ff_nvals;
    ggrs_u32 ff_g;
    ggrs_u32 ff_stride;
    ggrs_u32 ff_istride;
    ggrs_u32 ff_split;
    ggrs_u32 ff_self;
    ggrs_u32 dt_bits[11];
    int step_frame[11];
};
static_assert(sizeof(GgrsJitArgs) == 576, "host/device argument block mismatch");
static_assert(__builtin_offsetof(GgrsJitArgs, src) == 0, "argument block: offset of src");
static_assert(__builtin_offsetof(GgrsJitArgs, live) == 8, "argument block: offset of live");
static_assert(__builtin_offsetof(GgrsJitArgs, mtab) == 16, "argument block: offset of mtab");
static_assert(__builtin_offsetof(GgrsJitArgs, parts) == 24, "argument block: offset of parts");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_rows) == 32, "argument block: offset of ff_rows");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_out) == 40, "argument block: offset of ff_out");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_seq) == 48, "argument block: offset of ff_seq");
static_assert(__builtin_offsetof(GgrsJitArgs, live_rows) == 56, "argument block: offset of live_rows");
static_assert(__builtin_offsetof(GgrsJitArgs, load_rows) == 64, "argument block: offset of load_rows");
static_assert(__builtin_offsetof(GgrsJitArgs, op_bits) == 72, "argument block: offset of op_bits");
static_assert(__builtin_offsetof(GgrsJitArgs, len) == 80, "argument block: offset of len");
static_assert(__builtin_offsetof(GgrsJitArgs, save_dst) == 88, "argument block: offset of save_dst");
static_assert(__builtin_offsetof(GgrsJitArgs, save_rows) == 168, "argument block: offset of save_rows");
static_assert(__builtin_offsetof(GgrsJitArgs, save_len) == 248, "argument block: offset of save_len");
static_assert(__builtin_offsetof(GgrsJitArgs, save_frame) == 328, "argument block: offset of save_frame");
static_assert(__builtin_offsetof(GgrsJitArgs, save_pmask) == 368, "argument block: offset of save_pmask");
static_assert(__builtin_offsetof(GgrsJitArgs, live_pmask) == 408, "argument block: offset of live_pmask");
static_assert(__builtin_offsetof(GgrsJitArgs, nt_loads) == 412, "argument block: offset of nt_loads");
static_assert(__builtin_offsetof(GgrsJitArgs, n_ops) == 416, "argument block: offset of n_ops");
static_assert(__builtin_offsetof(GgrsJitArgs, n_saves) == 420, "argument block: offset of n_saves");
static_assert(__builtin_offsetof(GgrsJitArgs, n_steps) == 424, "argument block: offset of n_steps");
static_assert(__builtin_offsetof(GgrsJitArgs, src_is_live) == 428, "argument block: offset of src_is_live");
static_assert(__builtin_offsetof(GgrsJitArgs, skip_live) == 432, "argument block: offset of skip_live");
static_assert(__builtin_offsetof(GgrsJitArgs, dp_s) == 436, "argument block: offset of dp_s");
static_assert(__builtin_offsetof(GgrsJitArgs, part_stride) == 440, "argument block: offset of part_stride");
static_assert(__builtin_offsetof(GgrsJitArgs, part_tstride) == 444, "argument block: offset of part_tstride");
static_assert(__builtin_offsetof(GgrsJitArgs, nt) == 448, "argument block: offset of nt");
static_assert(__builtin_offsetof(GgrsJitArgs, n_units) == 452, "argument block: offset of n_units");
static_assert(__builtin_offsetof(GgrsJitArgs, cached_saves) == 456, "argument block: offset of cached_saves");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_blocks) == 460, "argument block: offset of ff_blocks");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_nvals) == 464, "argument block: offset of ff_nvals");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_g) == 468, "argument block: offset of ff_g");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_stride) == 472, "argument block: offset of ff_stride");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_istride) == 476, "argument block: offset of ff_istride");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_split) == 480, "argument block: offset of ff_split");
static_assert(__builtin_offsetof(GgrsJitArgs, ff_self) == 484, "argument block: offset of ff_self");
static_assert(__builtin_offsetof(GgrsJitArgs, dt_bits) == 488, "argument block: offset of dt_bits");
static_assert(__builtin_offsetof(GgrsJitArgs, step_frame) == 532, "argument block: offset of step_frame");
#line 1 "ggrs_jit_tick"
extern "C" __global__ __launch_bounds__(256) void ggrs_jit_tick(GgrsJitArgs a) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform, and the compiler knows it
    // FOLD-FORWARD role: the first ff_blocks workgroups (a multiple of 8: the XCD mapping below is unchanged) do not own a tile -- each folds one row
    // of partials the PREVIOUS launch on this stream left in device memory and hands the value, then its tag, to the host
    if (blockIdx.x < a.ff_blocks) {
        if (blockIdx.y == 0u && blockIdx.z == 0u && blockIdx.x < a.ff_nvals) {                   // ff_nvals = rows x chunks per row (ff_split)
            const uint32_t row = blockIdx.x / a.ff_split, ck = blockIdx.x % a.ff_split, per = (a.ff_g + a.ff_split - 1u) / a.ff_split;
            ff_fold_row((const uint64_t*)a.ff_rows + (uint64_t)row * a.ff_stride * (a.ff_self ? 2u : 1u), a.ff_istride,     // (self-fold: 16-byte cells)
                        ck * per, min(a.ff_g, (ck + 1u) * per), (row % 3u) == 2u,
                        (uint64_t*)a.ff_out + 2u * blockIdx.x, (uint64_t)a.ff_seq, a.ff_self ? (uint64_t)a.ff_seq : 0ull);   // cell blockIdx.x: {value, tag}
        }
        return;
    }
    const uint32_t bx = blockIdx.x - a.ff_blocks, gx = gridDim.x - a.ff_blocks;
    // batch members (blockIdx.z) with records: what differs between the launch's groups comes from member z's record, the rest from the argument block
    const GGRS_K unsigned char* const mb = a.mtab ? (const GGRS_K unsigned char*)(unsigned long)(a.mtab + (uint64_t)blockIdx.z * 304ull) : (const GGRS_K unsigned char*)0ul;
    const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
    const uint32_t o_first = a.dp_s ? blockIdx.y * a.dp_s : 0u;          // depth-parallel roles: this workgroup's share of the outputs
    const uint32_t o_last = a.dp_s ? min(o_first + a.dp_s, a.n_saves + 1u) : a.n_saves + 1u;
    const bool my_live = o_last == a.n_saves + 1u;
    if (a.dp_s && o_first == a.n_saves && !writes_live) return;
    // per-workgroup checksum partials [Save][component .. live count]: the waves fold into LDS
    __shared__ ggrs_u64 s_acc[16 * 3];
    for (uint32_t i = tid; i < 16u * 3u; i += 256u) s_acc[i] = 0;
    extern __shared__ ggrs_u64 s_lane[];                                  // [Save][checksummed component][lane]: a.n_saves * 2 * 64 cells (dynamic LDS)
    for (uint32_t i = tid; i < a.n_saves * 128u; i += 256u) s_lane[i] = 0;
    __syncthreads();
    // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (observed placement; used for speed only), and each XCD has its own
    // L2.  Handing XCD x the x-th CONTIGUOUS eighth of the tiles makes the workgroups that write neighbouring 1 KiB pieces of a
    // row share one L2, which merges them into long runs before they go to memory -- instead of every L2 seeing every 8th piece.
    const uint32_t g8 = gx >> 3;                                              // the grid is 8 x ceil(tiles / 8) workgroups (+ the fold-forward ones)
    const uint32_t tile = (bx & 7u) * g8 + (bx >> 3);
    if (tile * 4u >= a.n_units) return;                                       // padding workgroup of the last eighth
    {
    const uint32_t gu = tile * 4u + wave;                                     // this wave's 64-slot unit == its mask word
    const uint64_t e0 = (uint64_t)gu * 64u + lane;                             // this lane's slot
    const bool in_len = (uint64_t)gu * 64u < a.len;                                 // wave-uniform
    // word c of slot e lives at col_off[c] + (e >> 13) * tile_stride + (e & 8191) * word_bytes: the layout tile is the
    // wave's (uniform: SGPRs), the lane contributes one 32-bit offset per word size -> saddr-form accesses
    const uint64_t tbase = (uint64_t)(gu >> 7) * 491520ull;
    const uint32_t ei = (gu & 127u) * 64u + lane, lo1 = ei, lo2 = ei * 2u, lo4 = ei * 4u, lo8 = ei * 8u;
    (void)lo1; (void)lo2; (void)lo4; (void)lo8;
    const uint64_t wi8 = (uint64_t)gu * 8u;                                    // byte offset of this wave's mask word: bit `lane` is this slot
    const uint32_t sh = lane;
    const uint64_t mk_alive = *reinterpret_cast<const uint64_t*>(a.src + 256ull + wi8);
    bool alive_0 = (mk_alive >> sh) & 1ull;
    const uint64_t mk0 = *reinterpret_cast<const uint64_t*>(a.src + 126208ull + wi8);
    const bool p0_0 = (mk0 >> sh) & 1ull;
    const uint64_t mk1 = *reinterpret_cast<const uint64_t*>(a.src + 252160ull + wi8);
    const bool p1_0 = (mk1 >> sh) & 1ull;
    const uint64_t mk2 = *reinterpret_cast<const uint64_t*>(a.src + 378112ull + wi8);
    const bool p2_0 = (mk2 >> sh) & 1ull;
#define o0(blk) (sgpr_base((blk) + (507904ull + tbase)) + lo4)
#define b0(blk) ((blk) + (507904ull + tbase))
    uint32_t w0_0 = 0;
#define o1(blk) (sgpr_base((blk) + (540672ull + tbase)) + lo4)
#define b1(blk) ((blk) + (540672ull + tbase))
    uint32_t w1_0 = 0;
#define o2(blk) (sgpr_base((blk) + (573440ull + tbase)) + lo4)
#define b2(blk) ((blk) + (573440ull + tbase))
    uint32_t w2_0 = 0;
#define o3(blk) (sgpr_base((blk) + (770048ull + tbase)) + lo4)
#define b3(blk) ((blk) + (770048ull + tbase))
    uint32_t w3_0 = 0;
#define o4(blk) (sgpr_base((blk) + (802816ull + tbase)) + lo4)
#define b4(blk) ((blk) + (802816ull + tbase))
    uint32_t w4_0 = 0;
#define o5(blk) (sgpr_base((blk) + (835584ull + tbase)) + lo4)
#define b5(blk) ((blk) + (835584ull + tbase))
    uint32_t w5_0 = 0;
#define o6(blk) (sgpr_base((blk) + (868352ull + tbase)) + lo4)
#define b6(blk) ((blk) + (868352ull + tbase))
    uint32_t w6_0 = 0;
#define o7(blk) (sgpr_base((blk) + (901120ull + tbase)) + lo4)
#define b7(blk) ((blk) + (901120ull + tbase))
    uint32_t w7_0 = 0;
#define o8(blk) (sgpr_base((blk) + (933888ull + tbase)) + lo4)
#define b8(blk) ((blk) + (933888ull + tbase))
    uint32_t w8_0 = 0;
#define o9(blk) (sgpr_base((blk) + (966656ull + tbase)) + lo4)
#define b9(blk) ((blk) + (966656ull + tbase))
    uint32_t w9_0 = 0;
#define o10(blk) (sgpr_base((blk) + (606208ull + tbase)) + lo4)
#define b10(blk) ((blk) + (606208ull + tbase))
    uint32_t w10_0 = 0;
#define o11(blk) (sgpr_base((blk) + (638976ull + tbase)) + lo4)
#define b11(blk) ((blk) + (638976ull + tbase))
    uint32_t w11_0 = 0;
#define o12(blk) (sgpr_base((blk) + (671744ull + tbase)) + lo4)
#define b12(blk) ((blk) + (671744ull + tbase))
    uint32_t w12_0 = 0;
#define o13(blk) (sgpr_base((blk) + (704512ull + tbase)) + lo8)
#define b13(blk) ((blk) + (704512ull + tbase))
    uint64_t w13_0 = 0;
    if (in_len) {
        if (a.load_rows == 0x3c07ull) {
          if (a.nt_loads) {
            w0_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o0(a.src));
            w1_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o1(a.src));
            w2_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o2(a.src));
            w10_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o10(a.src));
            w11_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o11(a.src));
            w12_0 = __builtin_nontemporal_load((const GGRS_G uint32_t*)o12(a.src));
            w13_0 = __builtin_nontemporal_load((const GGRS_G uint64_t*)o13(a.src));
          } else {
            w0_0 = *(const GGRS_G uint32_t*)o0(a.src);
            w1_0 = *(const GGRS_G uint32_t*)o1(a.src);
            w2_0 = *(const GGRS_G uint32_t*)o2(a.src);
            w10_0 = *(const GGRS_G uint32_t*)o10(a.src);
            w11_0 = *(const GGRS_G uint32_t*)o11(a.src);
            w12_0 = *(const GGRS_G uint32_t*)o12(a.src);
            w13_0 = *(const GGRS_G uint64_t*)o13(a.src);
          }
        } else {
            if ((a.load_rows >> 0u) & 1ull) w0_0 = *(const GGRS_G uint32_t*)o0(a.src);
            if ((a.load_rows >> 1u) & 1ull) w1_0 = *(const GGRS_G uint32_t*)o1(a.src);
            if ((a.load_rows >> 2u) & 1ull) w2_0 = *(const GGRS_G uint32_t*)o2(a.src);
            if ((a.load_rows >> 3u) & 1ull) w3_0 = *(const GGRS_G uint32_t*)o3(a.src);
            if ((a.load_rows >> 4u) & 1ull) w4_0 = *(const GGRS_G uint32_t*)o4(a.src);
            if ((a.load_rows >> 5u) & 1ull) w5_0 = *(const GGRS_G uint32_t*)o5(a.src);
            if ((a.load_rows >> 6u) & 1ull) w6_0 = *(const GGRS_G uint32_t*)o6(a.src);
            if ((a.load_rows >> 7u) & 1ull) w7_0 = *(const GGRS_G uint32_t*)o7(a.src);
            if ((a.load_rows >> 8u) & 1ull) w8_0 = *(const GGRS_G uint32_t*)o8(a.src);
            if ((a.load_rows >> 9u) & 1ull) w9_0 = *(const GGRS_G uint32_t*)o9(a.src);
            if ((a.load_rows >> 10u) & 1ull) w10_0 = *(const GGRS_G uint32_t*)o10(a.src);
            if ((a.load_rows >> 11u) & 1ull) w11_0 = *(const GGRS_G uint32_t*)o11(a.src);
            if ((a.load_rows >> 12u) & 1ull) w12_0 = *(const GGRS_G uint32_t*)o12(a.src);
            if ((a.load_rows >> 13u) & 1ull) w13_0 = *(const GGRS_G uint64_t*)o13(a.src);
        }
    }
    const uint64_t ordB_0 = sea_order_lane(e0);
    uint32_t mt0 = (uint32_t)((((uint64_t)w2_0 >> 0u) & 0xffffffffull) << 0u); uint64_t ma0 = a.n_saves ? sea_diffuse(SEA_K1 ^ (uint64_t)mt0) : 0ull;   // memoised tail of checksum spec 0
    uint32_t mt1 = (uint32_t)((((uint64_t)w12_0 >> 0u) & 0xffffffffull) << 0u); uint64_t ma1 = a.n_saves ? sea_diffuse(SEA_K1 ^ (uint64_t)mt1) : 0ull;   // memoised tail of checksum spec 1
    uint32_t si = 0, sj = 0;
    for (uint32_t op = 0; op < a.n_ops; ++op) {
        if (!((a.op_bits >> op) & 1ull)) {
            // ---------------- SaveWorld
            if (si < o_first) { ++si; continue; }                          // another role's snapshot
            if (si >= o_last) break;
            const uint64_t alive_now = __ballot(alive_0);
            unsigned char* dst = mb ? (unsigned char*)mb_u64(mb, 0u + 8u * si) : a.save_dst[si];
            if (dst) {
                uint64_t rows = mb ? mb_u64(mb, 80u + 8u * si) : a.save_rows[si];
                const uint32_t pmask_s = mb ? mb_u32(mb, 256u + 4u * si) : a.save_pmask[si];
                if (in_len) {
                    if (a.nt && !((a.cached_saves >> si) & 1u)) {
                        if (rows == 0x3c07ull) {
                            st4nt(b0(dst), lo4, w0_0);
                            st4nt(b1(dst), lo4, w1_0);
                            st4nt(b2(dst), lo4, w2_0);
                            st4nt(b10(dst), lo4, w10_0);
                            st4nt(b11(dst), lo4, w11_0);
                            st4nt(b12(dst), lo4, w12_0);
                            st8nt(b13(dst), lo8, w13_0);
                        } else {
                            if ((rows >> 0u) & 1ull) st4nt(b0(dst), lo4, w0_0);
                            if ((rows >> 1u) & 1ull) st4nt(b1(dst), lo4, w1_0);
                            if ((rows >> 2u) & 1ull) st4nt(b2(dst), lo4, w2_0);
                            if ((rows >> 3u) & 1ull) st4nt(b3(dst), lo4, w3_0);
                            if ((rows >> 4u) & 1ull) st4nt(b4(dst), lo4, w4_0);
                            if ((rows >> 5u) & 1ull) st4nt(b5(dst), lo4, w5_0);
                            if ((rows >> 6u) & 1ull) st4nt(b6(dst), lo4, w6_0);
                            if ((rows >> 7u) & 1ull) st4nt(b7(dst), lo4, w7_0);
                            if ((rows >> 8u) & 1ull) st4nt(b8(dst), lo4, w8_0);
                            if ((rows >> 9u) & 1ull) st4nt(b9(dst), lo4, w9_0);
                            if ((rows >> 10u) & 1ull) st4nt(b10(dst), lo4, w10_0);
                            if ((rows >> 11u) & 1ull) st4nt(b11(dst), lo4, w11_0);
                            if ((rows >> 12u) & 1ull) st4nt(b12(dst), lo4, w12_0);
                            if ((rows >> 13u) & 1ull) st8nt(b13(dst), lo8, w13_0);
                        }
                    } else {
                        if (rows == 0x3c07ull) {
                            st4(b0(dst), lo4, w0_0);
                            st4(b1(dst), lo4, w1_0);
                            st4(b2(dst), lo4, w2_0);
                            st4(b10(dst), lo4, w10_0);
                            st4(b11(dst), lo4, w11_0);
                            st4(b12(dst), lo4, w12_0);
                            st8(b13(dst), lo8, w13_0);
                        } else {
                            if ((rows >> 0u) & 1ull) st4(b0(dst), lo4, w0_0);
                            if ((rows >> 1u) & 1ull) st4(b1(dst), lo4, w1_0);
                            if ((rows >> 2u) & 1ull) st4(b2(dst), lo4, w2_0);
                            if ((rows >> 3u) & 1ull) st4(b3(dst), lo4, w3_0);
                            if ((rows >> 4u) & 1ull) st4(b4(dst), lo4, w4_0);
                            if ((rows >> 5u) & 1ull) st4(b5(dst), lo4, w5_0);
                            if ((rows >> 6u) & 1ull) st4(b6(dst), lo4, w6_0);
                            if ((rows >> 7u) & 1ull) st4(b7(dst), lo4, w7_0);
                            if ((rows >> 8u) & 1ull) st4(b8(dst), lo4, w8_0);
                            if ((rows >> 9u) & 1ull) st4(b9(dst), lo4, w9_0);
                            if ((rows >> 10u) & 1ull) st4(b10(dst), lo4, w10_0);
                            if ((rows >> 11u) & 1ull) st4(b11(dst), lo4, w11_0);
                            if ((rows >> 12u) & 1ull) st4(b12(dst), lo4, w12_0);
                            if ((rows >> 13u) & 1ull) st8(b13(dst), lo8, w13_0);
                        }
                    }
                }
                if (lane == 0) {
                    *reinterpret_cast<uint64_t*>(dst + 256ull + wi8) = alive_now;
                    if ((pmask_s >> 0u) & 1u) *reinterpret_cast<uint64_t*>(dst + 126208ull + wi8) = mk0;
                    if ((pmask_s >> 1u) & 1u) *reinterpret_cast<uint64_t*>(dst + 252160ull + wi8) = mk1;
                    if ((pmask_s >> 2u) & 1u) *reinterpret_cast<uint64_t*>(dst + 378112ull + wi8) = mk2;
                }
                if (gu == 0 && lane == 0) {
                    Header h; h.len = mb ? mb_u64(mb, 160u + 8u * si) : a.save_len[si]; h.frame = a.save_frame[si]; h.pad0 = 0; h.active = 0; h.checksum[0] = 0; h.checksum[1] = 0;
                    *reinterpret_cast<Header*>(dst) = h;
                }
            }
            ggrs_u64* acc = s_acc + si * 3u;                                 // this Save's partials of the workgroup (LDS)
            {   // ComponentChecksumPlugin::update (component_checksum.rs:77-90): per-entity hash, paired with the order index
                uint64_t hx = 0;
                { const uint32_t tv = (uint32_t)((((uint64_t)w2_0 >> 0u) & 0xffffffffull) << 0u);
                  if (__ballot(tv != mt0) != 0ull) { mt0 = tv; ma0 = sea_diffuse(SEA_K1 ^ (uint64_t)tv); }
                  const uint64_t A = sea_diffuse(SEA_K0 ^ ((((uint64_t)w0_0 >> 0u) & 0xffffffffull) << 0u | (((uint64_t)w1_0 >> 0u) & 0xffffffffull) << 32u));
                  const uint64_t inner = sea_diffuse(ma0 ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ull);
                  hx = (alive_0 && p0_0) ? sea_pair_pre(ordB_0, inner) : 0ull; }
                atomicXor(&s_lane[(si * 2u + 0u) * 64u + lane], (ggrs_u64)hx);
            }
            {   // ComponentChecksumPlugin::update (component_checksum.rs:77-90): per-entity hash, paired with the order index
                uint64_t hx = 0;
                { const uint32_t tv = (uint32_t)((((uint64_t)w12_0 >> 0u) & 0xffffffffull) << 0u);
                  if (__ballot(tv != mt1) != 0ull) { mt1 = tv; ma1 = sea_diffuse(SEA_K1 ^ (uint64_t)tv); }
                  const uint64_t A = sea_diffuse(SEA_K0 ^ ((((uint64_t)w10_0 >> 0u) & 0xffffffffull) << 0u | (((uint64_t)w11_0 >> 0u) & 0xffffffffull) << 32u));
                  const uint64_t inner = sea_diffuse(ma1 ^ SEA_K2 ^ SEA_K3 ^ A ^ 12ull);
                  hx = (alive_0 && p1_0) ? sea_pair_pre(ordB_0, inner) : 0ull; }
                atomicXor(&s_lane[(si * 2u + 1u) * 64u + lane], (ggrs_u64)hx);
            }
            if (lane == 0) atomicAdd(&acc[2], (ggrs_u64)__popcll(alive_now));
            ++si;
            if (si >= o_last) break;
        } else {
            // ---------------- AdvanceWorld: the registered systems, in order
            const float dt = __uint_as_float(a.dt_bits[sj]);
            if (alive_0 && p0_0 && p1_0) {                                     // particles.rs:272-280
                { const float nv = __uint_as_float(w10_0) + __uint_as_float(0x00000000u) * dt; w10_0 = __float_as_uint(nv); w0_0 = __float_as_uint(__uint_as_float(w0_0) + nv * dt); }
                { const float nv = __uint_as_float(w11_0) + __uint_as_float(0xc3480000u) * dt; w11_0 = __float_as_uint(nv); w1_0 = __float_as_uint(__uint_as_float(w1_0) + nv * dt); }
                { const float nv = __uint_as_float(w12_0) + __uint_as_float(0x00000000u) * dt; w12_0 = __float_as_uint(nv); w2_0 = __float_as_uint(__uint_as_float(w2_0) + nv * dt); }
            }
            if (alive_0 && p2_0) { w13_0 -= 1; if (w13_0 == 0) alive_0 = false; }      // particles.rs:282-289
            ++sj;
        }
    }
    // ---- the live world, written once
    if (my_live && writes_live) {
        const uint64_t alive_now = __ballot(alive_0);
        unsigned char* const live_p = mb ? (unsigned char*)mb_u64(mb, 240u) : a.live;
        uint64_t live_rows_v = mb ? mb_u64(mb, 248u) : a.live_rows;
        const uint32_t live_pm_v = mb ? mb_u32(mb, 296u) : a.live_pmask;
        if (in_len) {
            if (live_rows_v == 0x3c07ull) {
                st4(b0(live_p), lo4, w0_0);
                st4(b1(live_p), lo4, w1_0);
                st4(b2(live_p), lo4, w2_0);
                st4(b10(live_p), lo4, w10_0);
                st4(b11(live_p), lo4, w11_0);
                st4(b12(live_p), lo4, w12_0);
                st8(b13(live_p), lo8, w13_0);
            } else {
                if ((live_rows_v >> 0u) & 1ull) st4(b0(live_p), lo4, w0_0);
                if ((live_rows_v >> 1u) & 1ull) st4(b1(live_p), lo4, w1_0);
                if ((live_rows_v >> 2u) & 1ull) st4(b2(live_p), lo4, w2_0);
                if ((live_rows_v >> 3u) & 1ull) st4(b3(live_p), lo4, w3_0);
                if ((live_rows_v >> 4u) & 1ull) st4(b4(live_p), lo4, w4_0);
                if ((live_rows_v >> 5u) & 1ull) st4(b5(live_p), lo4, w5_0);
                if ((live_rows_v >> 6u) & 1ull) st4(b6(live_p), lo4, w6_0);
                if ((live_rows_v >> 7u) & 1ull) st4(b7(live_p), lo4, w7_0);
                if ((live_rows_v >> 8u) & 1ull) st4(b8(live_p), lo4, w8_0);
                if ((live_rows_v >> 9u) & 1ull) st4(b9(live_p), lo4, w9_0);
                if ((live_rows_v >> 10u) & 1ull) st4(b10(live_p), lo4, w10_0);
                if ((live_rows_v >> 11u) & 1ull) st4(b11(live_p), lo4, w11_0);
                if ((live_rows_v >> 12u) & 1ull) st4(b12(live_p), lo4, w12_0);
                if ((live_rows_v >> 13u) & 1ull) st8(b13(live_p), lo8, w13_0);
            }
        }
        if (lane == 0) {
            *reinterpret_cast<uint64_t*>(live_p + 256ull + wi8) = alive_now;
            if ((live_pm_v >> 0u) & 1u) *reinterpret_cast<uint64_t*>(live_p + 126208ull + wi8) = mk0;
            if ((live_pm_v >> 1u) & 1u) *reinterpret_cast<uint64_t*>(live_p + 252160ull + wi8) = mk1;
            if ((live_pm_v >> 2u) & 1u) *reinterpret_cast<uint64_t*>(live_p + 378112ull + wi8) = mk2;
        }
    }
    }   // the wave's unit
    // ---- this workgroup's partial rows (blockIdx.z: member of a batch of identical checksum-only groups)
    __syncthreads();
    for (uint32_t r_ = wave; r_ < a.n_saves * 2u; r_ += 4u) {                  // one row per wave and trip: XOR over its 64 lanes
        const uint32_t sv = r_ / 2u;
        if (sv < o_first || sv >= o_last) continue;
        const ggrs_u64 v_ = wave_xor(s_lane[r_ * 64u + lane]);
        if (lane == 0) s_acc[sv * 3u + r_ % 2u] = v_;
    }
    __syncthreads();
    for (uint32_t i = tid; i < a.n_saves * 3u; i += 256u) {
        const uint32_t sv = i / 3u;
        if (sv >= o_first && sv < o_last) {
            const uint64_t at_ = ((uint64_t)blockIdx.z * a.n_saves * 3u + i) * a.part_stride + (uint64_t)tile * a.part_tstride;
            if (a.ff_self) {                                                  // self-fold: a 16-byte cell {value, tag} in ONE sc1 store, read by a fold workgroup of THIS launch
                const uint64_t v_ = s_acc[i], sq_ = (uint64_t)a.ff_seq;
                const ff_u32x4 q_ = {(uint32_t)v_, (uint32_t)(v_ >> 32), (uint32_t)sq_, (uint32_t)(sq_ >> 32)};
                asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(reinterpret_cast<ff_u32x4*>(a.parts) + at_), "v"(q_) : "memory");
            } else a.parts[at_] = s_acc[i];
        }
    }
}
