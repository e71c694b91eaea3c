// kernels.hpp -- gfx950 (CDNA4, wave64) device code of the rollback re-simulation engine.
//
// Everything here is HBM-bound element-wise integer/f32 work over SoA word columns: no MFMA
// (there is no contraction anywhere on this path).  Design rules applied:
//   * one workgroup (256 threads = 4 waves) owns one TILE of 1024 consecutive slots in EVERY
//     kernel, and tile t is always blockIdx t.  Workgroup b lands on XCD b % 8, so the same
//     XCD (and its private 4 MiB L2) touches the same slots in load -> advance -> save chains;
//   * word columns are stored TILE-MAJOR inside a state block, in LAYOUT TILES of 8192 slots (= 8 workgroup tiles): the
//     8192 slots of layout tile T of every registered word sit next to each other (tile_stride = bytes of all words of
//     8192 slots, 480 KiB for the particles world);
//     word w of slot e lives at  col_off[w] + (e >> 13) * tile_stride + (e & 8191) * word_bytes.
//     A workgroup's 1024-slot tile t is sub-tile t & 7 of layout tile t >> 3: its 4 KiB row of a 4-byte word starts at
//     col_off[w] + (t >> 3) * tile_stride + (t & 7) * 4096, and CONSECUTIVE ROWS OF ONE WORKGROUP ARE 32 KiB APART --
//     the period of the HBM channel interleave (128 channels x 256 B), so the rows a wave stores back to back land in
//     the same DRAM pages (scripts/ubench3.hip, layout 2 / G = 8: 100.6 us vs 108.5 us for 1024-slot layout tiles).
//     Live-only side columns keep the same formula with tile_stride = 8192 * word_bytes (a plain array);
//   * every column access is 16 B per lane (dwordx4), 1 KiB per wave instruction, all loads
//     of a tile issued before the first store;
//   * Rollback-entity liveness is a 1 bit/slot mask; despawn masks are built with wave64
//     __ballot and a scalar bit-interleave, live counts with popcount;
//   * f32 integration uses explicit __fmul_rn/__fadd_rn: Rust never contracts a*b+c, and the
//     checksum hashes the raw f32 bits, so results must be bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ggrs {

constexpr int TILE = 1024;     // slots per workgroup
// LT_SHIFT = 13 / LAYOUT_TILE = 8192 (slots of a LAYOUT tile = 8 workgroup tiles) are declared in device_prelude.hpp
constexpr int TPB = 256;       // threads per workgroup (4 waves of 64)
constexpr int MAX_ROWS = 128;  // 4 KiB copy rows per tile (a 4-byte column = 1 row, 8-byte = 2): 64 eight-byte words
constexpr int MAX_COMPS = GGRS_MAX_COMPONENTS;
constexpr int MAX_MASKS = MAX_COMPS + 1;  // alive + one presence mask per component
constexpr int MAX_UNITS = 32;

// SeaHash (seahash 4.1), the state-block Header, wave_xor and the box_game step: shared with the run-time generated kernels
#define GGRS_SHARED_CODE(...) __VA_ARGS__
#include "device_prelude.hpp"
#undef GGRS_SHARED_CODE

// byte offset (inside a state block) of word-column element `e`: see the layout note at the top
__host__ __device__ __forceinline__ uint64_t col_at(uint64_t col_off, uint32_t tile_stride, uint32_t word_bytes, uint64_t e) {
    return col_off + (e >> LT_SHIFT) * (uint64_t)tile_stride + (e & (uint64_t)(LAYOUT_TILE - 1)) * (uint64_t)word_bytes;
}
// byte offset (relative to the column's col_off) of the first element of workgroup tile t (1024 slots)
__host__ __device__ __forceinline__ uint64_t wtile_off(uint32_t t, uint32_t tile_stride, uint32_t word_bytes) {
    return (uint64_t)(t >> (LT_SHIFT - 10)) * tile_stride + (uint64_t)(t & ((LAYOUT_TILE / TILE) - 1u)) * (uint32_t)(TILE * word_bytes);
}

// ------------------------------------------------------------------ kernel argument blocks
struct RowDesc {          // one 4 KiB-per-tile copy row of the packed state block
    uint64_t col_off;     // byte offset of the column inside the state block
    uint32_t roff;        // byte offset of this row inside the column's tile (0 or 4096)
    uint32_t tile_stride; // bytes one tile of this column spans (TILE * word_bytes)
    uint32_t word_bytes;
    uint32_t bytes;       // bytes of this row per workgroup tile: 4096, or 1024 / 2048 for a 1- / 2-byte word
};
struct CopyPlan {
    uint32_t n_rows, n_masks, n_wide, pad;   // rows [0, n_wide) are 4096-byte rows (moved in straight-line batches), the rest are short
    uint64_t mask_off[MAX_MASKS];
    RowDesc row[MAX_ROWS];
};
struct StepArgs {         // fused GgrsSchedule step of the particles workload
    uint8_t* state;
    uint64_t off_alive, off_pT, off_pV, off_pL;
    uint64_t off_t[3], off_v[3], off_ttl;
    uint32_t dt_bits; float g[3];
    uint64_t* part_T; uint64_t* part_V; uint64_t* part_cnt;
    uint32_t ts; uint32_t pad;            // tile stride of the rollback word columns
};

struct UnitDesc { uint64_t off; uint32_t wb; uint32_t ts; };   // one hashed word: slot e's word of wb bytes at col_at(off, ts, wb, e)
struct CksArgs {          // generic component checksum
    const uint8_t* state;
    uint64_t off_alive;
    uint32_t n_cks; uint32_t part_stride;
    uint64_t off_present[MAX_COMPS];
    uint32_t n_units[MAX_COMPS];
    uint32_t unit_base[MAX_COMPS];
    uint64_t* parts;      // [n_cks][part_stride]
    uint64_t* part_cnt;   // [part_stride]
};

// ------------------------------------------------------------------ helpers
// bit i of x (16 bits) -> bit 4*i
__device__ __forceinline__ uint64_t spread4(uint64_t x) {
    x &= 0xFFFFULL;
    x = (x | (x << 24)) & 0x000000FF000000FFULL;
    x = (x | (x << 12)) & 0x000F000F000F000FULL;
    x = (x | (x << 6)) & 0x0303030303030303ULL;
    x = (x | (x << 3)) & 0x1111111111111111ULL;
    return x;
}

// ------------------------------------------------------------------ finalize (device function)
// Hash each component's XOR once more (component_checksum.rs:92-95), add the entity part
// (entity_checksum.rs:29-52), XOR-fold all parts (checksum.rs:88-99).  Runs inside workgroup 0
// of the SaveWorld copy kernel (the partials were completed by the previous kernel on the
// stream), so a SaveWorld is ONE launch.
struct FinalizeArgs {
    const uint64_t* parts;      // [n_cks][part_stride]
    const uint64_t* part_cnt;   // [part_stride]
    uint32_t n_cks, part_stride, n_parts, enabled;
    uint64_t total_len;
    uint64_t* out;              // {lo, hi} of Checksum(u128)
    Header* live_hdr;
};
constexpr int MAX_CKS = MAX_COMPS;

__device__ __forceinline__ void finalize_block(const FinalizeArgs& f, uint64_t* checksum_out) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint64_t fin_sx[MAX_CKS + 1];
    // one partial column per wave at a time (wave-uniform control flow, loads pipelined)
    for (uint32_t k = wave; k <= f.n_cks; k += 4) {
        const bool is_cnt = (k == f.n_cks);
        const uint64_t* __restrict__ p = is_cnt ? f.part_cnt : f.parts + (uint64_t)k * f.part_stride;
        uint64_t x = 0, sum = 0;
#pragma unroll 8
        for (uint32_t i = lane; i < f.n_parts; i += 64) { const uint64_t v = p[i]; x ^= v; sum += v; }
        x = wave_xor(x);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) fin_sx[k] = is_cnt ? sum : x;
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t total = 0;
        for (uint32_t k = 0; k < f.n_cks; ++k) total ^= sea_one(fin_sx[k]);
        const uint64_t active = fin_sx[f.n_cks];
        total ^= sea_pair(active, f.total_len);      // hash(active, total): same shape as pair()
        f.out[0] = total; f.out[1] = 0;              // `as u128` of a u64: upper half always 0
        f.live_hdr->active = active;
        f.live_hdr->checksum[0] = total; f.live_hdr->checksum[1] = 0;
        checksum_out[0] = total; checksum_out[1] = active;
    }
}

// ------------------------------------------------------------------ k_copy_state
// SaveWorld's Snapshot set (component_snapshot.rs:66-84, entity.rs:39-51, ring push
// mod.rs:147-181) and LoadWorld's Entity+Data sets (entity.rs:55-99,
// component_snapshot.rs:95-123, ring rollback mod.rs:210-226) both reduce to: copy every
// registered word column over [0, len) plus the liveness/presence masks between the live
// block and a ring slot.  Algorithmic traffic: 2 x (bytes per slot) per entity.
//
// Full tiles take a branch-free path: rows are moved in straight-line batches of 8/4/2/1
// (every load of a batch in flight before its first store -- a per-lane predicate around a
// load makes hipcc drain vmcnt after every single load).  Only the last, ragged tile uses the
// predicated path.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int B, bool NT>
__device__ __forceinline__ void copy_rows(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                          const CopyPlan& plan, uint32_t r0, uint32_t t, uint32_t tid) {
    u32x4 v[B];
    uint64_t pos[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const RowDesc rd = plan.row[r0 + j];
        pos[j] = rd.col_off + wtile_off(t, rd.tile_stride, rd.word_bytes) + rd.roff + (uint64_t)tid * 16;
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
        if (NT) v[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + pos[j]));
        else v[j] = *reinterpret_cast<const u32x4*>(src + pos[j]);
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
        if (NT) __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(dst + pos[j]));
        else *reinterpret_cast<u32x4*>(dst + pos[j]) = v[j];
    }
}

template <bool NT>
__global__ __launch_bounds__(TPB) void k_copy_state(const uint8_t* __restrict__ src,
                                                    uint8_t* __restrict__ dst, CopyPlan plan,
                                                    uint64_t len, Header hdr, FinalizeArgs fin) {
    const uint32_t t = blockIdx.x, tid = threadIdx.x;
    __shared__ uint64_t cks[2];
    if (t == 0 && fin.enabled) finalize_block(fin, cks);       // workgroup-uniform branch

    const uint32_t n_rows = plan.n_rows;
    if (((uint64_t)t + 1) * TILE <= len) {
        const uint32_t n_wide = plan.n_wide;
        uint32_t r = 0;
        for (; r + 8 <= n_wide; r += 8) copy_rows<8, NT>(src, dst, plan, r, t, tid);
        if (r + 4 <= n_wide) { copy_rows<4, NT>(src, dst, plan, r, t, tid); r += 4; }
        if (r + 2 <= n_wide) { copy_rows<2, NT>(src, dst, plan, r, t, tid); r += 2; }
        if (r < n_wide) copy_rows<1, NT>(src, dst, plan, r, t, tid);
        for (r = n_wide; r < n_rows; ++r) {                  // 1- / 2-byte words: 1 or 2 KiB per tile, whole waves (wave-uniform predicate)
            const RowDesc rd = plan.row[r];
            const uint64_t pos = rd.col_off + wtile_off(t, rd.tile_stride, rd.word_bytes) + (uint64_t)tid * 16;
            if (tid * 16u < rd.bytes) *reinterpret_cast<uint4*>(dst + pos) = *reinterpret_cast<const uint4*>(src + pos);
        }
    } else {
        for (uint32_t r = 0; r < n_rows; ++r) {
            const RowDesc rd = plan.row[r];
            const uint64_t pos = wtile_off(t, rd.tile_stride, rd.word_bytes) + rd.roff + (uint64_t)tid * 16;
            const uint64_t slot0 = (uint64_t)t * TILE + (rd.roff + tid * 16u) / rd.word_bytes;   // first slot of this lane's 16 bytes
            if (slot0 < len && tid * 16u < rd.bytes)
                *reinterpret_cast<uint4*>(dst + rd.col_off + pos) = *reinterpret_cast<const uint4*>(src + rd.col_off + pos);
        }
    }
    // masks: 16 u64 words per tile per mask (copied whole, so stale bits beyond the source's
    // len are cleared in the destination)
    for (uint32_t i = tid; i < 16u * plan.n_masks; i += blockDim.x) {      // (16 components + the liveness mask = 272 words: more than one trip of 256 threads)
        const uint32_t m = i >> 4, wi = i & 15u;
        const uint64_t o = plan.mask_off[m] + ((uint64_t)t * 16 + wi) * 8;
        *reinterpret_cast<uint64_t*>(dst + o) = *reinterpret_cast<const uint64_t*>(src + o);
    }
    if (t == 0) {
        if (fin.enabled) __syncthreads();
        if (tid == 0) {
            if (fin.enabled) { hdr.checksum[0] = cks[0]; hdr.checksum[1] = 0; hdr.active = cks[1]; }
            *reinterpret_cast<Header*>(dst) = hdr;
        }
    }
}

// ------------------------------------------------------------------ k_particles_step
// The GgrsSchedule of examples/stress_tests/particles.rs:233-240 as one pass:
//   update_particles  (particles.rs:272-280)  v += g*dt ; x += v*dt     (unfused mul, add)
//   despawn_particles (particles.rs:282-289)  ttl -= 1 ; ttl == 0 -> despawn
// and, because translation and velocity are already in registers, the per-entity part of
// ComponentChecksumPlugin::update (component_checksum.rs:77-90) for the NEXT SaveWorld:
// per-workgroup XOR partials + live count, folded later by finalize_block.
// Algorithmic traffic: 64 B per live entity (12+12+8 read, same written).
//
// All 4 mask loads and all 8 column loads of a lane are issued unconditionally and together
// (slots up to the padded capacity are always mapped); per-entity conditions are selects, and
// only wave-uniform conditions guard the stores.
template <bool UPD, bool TTL, bool CKS_T, bool CKS_V>
__global__ __launch_bounds__(TPB) void k_particles_step(StepArgs a) {
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t e0 = (uint64_t)t * TILE + (uint64_t)tid * 4;     // first of this lane's 4 slots
    const uint64_t tb4 = wtile_off(t, a.ts, 4) + (uint64_t)tid * 16;    // its 16 bytes inside a 4-byte column's tile row
    const uint64_t tb8 = wtile_off(t, a.ts, 8) + (uint64_t)tid * 32;    // its 32 bytes inside an 8-byte column's tile rows
    const uint64_t w0 = (uint64_t)t * 16 + wave * 4;                 // first mask word of this wave
    const uint32_t sh = (lane & 15u) * 4;
    const uint64_t wi = w0 + (lane >> 4);
    constexpr bool NEED_T = UPD || CKS_T, NEED_V = UPD || CKS_V;

    // ---- every load of the tile, back to back
    const uint64_t alive_w = *reinterpret_cast<const uint64_t*>(a.state + a.off_alive + wi * 8);
    uint64_t pT_w = 0, pV_w = 0, pL_w = 0;
    if (NEED_T) pT_w = *reinterpret_cast<const uint64_t*>(a.state + a.off_pT + wi * 8);
    if (NEED_V) pV_w = *reinterpret_cast<const uint64_t*>(a.state + a.off_pV + wi * 8);
    if (TTL) pL_w = *reinterpret_cast<const uint64_t*>(a.state + a.off_pL + wi * 8);
    float4 tx[3], vv[3];
    ulonglong2 tl[2];
    if (NEED_T) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tx[k] = *reinterpret_cast<const float4*>(a.state + a.off_t[k] + tb4);
    }
    if (NEED_V) {
#pragma unroll
        for (int k = 0; k < 3; ++k) vv[k] = *reinterpret_cast<const float4*>(a.state + a.off_v[k] + tb4);
    }
    if (TTL) {
        tl[0] = *reinterpret_cast<const ulonglong2*>(a.state + a.off_ttl + tb8);
        tl[1] = *reinterpret_cast<const ulonglong2*>(a.state + a.off_ttl + tb8 + 16);
    }

    const uint32_t n_alive = (uint32_t)(alive_w >> sh) & 0xFu;
    const uint32_t n_T = (uint32_t)(pT_w >> sh) & 0xFu, n_V = (uint32_t)(pV_w >> sh) & 0xFu,
                   n_L = (uint32_t)(pL_w >> sh) & 0xFu;
    const uint32_t m_upd = UPD ? (n_alive & n_T & n_V) : 0u;   // Query<(&mut Transform,&mut Velocity)>
    const uint32_t m_ttl = TTL ? (n_alive & n_L) : 0u;         // Query<(Entity,&mut Ttl)>

    // ---- update_particles (selects, no per-lane branches)
    if (UPD) {
        const float dt = __uint_as_float(a.dt_bits);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float gd = __fmul_rn(a.g[k], dt);           // gravity * time_step
            float* x = reinterpret_cast<float*>(&tx[k]);
            float* v = reinterpret_cast<float*>(&vv[k]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = (m_upd >> j) & 1u;
                const float nv = __fadd_rn(v[j], gd);                       // **velocity += ...
                const float nx = __fadd_rn(x[j], __fmul_rn(nv, dt));        // translation += **velocity * time_step
                v[j] = on ? nv : v[j];
                x[j] = on ? nx : x[j];
            }
        }
        if (__ballot(m_upd != 0) != 0) {                      // wave-uniform
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                *reinterpret_cast<float4*>(a.state + a.off_t[k] + tb4) = tx[k];
                *reinterpret_cast<float4*>(a.state + a.off_v[k] + tb4) = vv[k];
            }
        }
    }

    // ---- despawn_particles
    uint32_t kill = 0;
    if (TTL) {
        uint64_t* q = reinterpret_cast<uint64_t*>(&tl[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = (m_ttl >> j) & 1u;
            const uint64_t nq = q[j] - 1;                    // usize, wrapping
            q[j] = on ? nq : q[j];
            kill |= (on && nq == 0) ? (1u << j) : 0u;
        }
        if (__ballot(m_ttl != 0) != 0) {
            *reinterpret_cast<ulonglong2*>(a.state + a.off_ttl + tb8) = tl[0];
            *reinterpret_cast<ulonglong2*>(a.state + a.off_ttl + tb8 + 16) = tl[1];
        }
    }
    const uint32_t n_new = n_alive & ~kill;

    // ---- new liveness words for this wave's 256 slots: 4 ballots + scalar bit-interleave
    uint32_t cnt = 0;
    if (TTL) {
        const uint64_t b0 = __ballot((n_new >> 0) & 1u), b1 = __ballot((n_new >> 1) & 1u),
                       b2 = __ballot((n_new >> 2) & 1u), b3 = __ballot((n_new >> 3) & 1u);
        const bool any_kill = __ballot(kill != 0) != 0;
        uint64_t mine = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
            cnt += (uint32_t)__popcll(nw);
            if (lane == (uint32_t)w) mine = nw;
        }
        if (any_kill && lane < 4)
            *reinterpret_cast<uint64_t*>(a.state + a.off_alive + (w0 + lane) * 8) = mine;
    } else if (CKS_T || CKS_V) {
        cnt = (uint32_t)__popcll(__ballot(n_new & 1u)) + (uint32_t)__popcll(__ballot(n_new & 2u)) +
              (uint32_t)__popcll(__ballot(n_new & 4u)) + (uint32_t)__popcll(__ballot(n_new & 8u));
    }

    // ---- checksum partials of the post-step state
    if (CKS_T || CKS_V) {
        uint64_t hT = 0, hV = 0;
        const uint32_t c_T = CKS_T ? (n_new & n_T) : 0u, c_V = CKS_V ? (n_new & n_V) : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t order = e0 + j;                   // RollbackOrdered::order == slot
            if (CKS_T) {
                const uint64_t h = sea_pair(order, sea_inner3(__float_as_uint(reinterpret_cast<float*>(&tx[0])[j]),
                                                              __float_as_uint(reinterpret_cast<float*>(&tx[1])[j]),
                                                              __float_as_uint(reinterpret_cast<float*>(&tx[2])[j])));
                hT ^= ((c_T >> j) & 1u) ? h : 0ULL;
            }
            if (CKS_V) {
                const uint64_t h = sea_pair(order, sea_inner3(__float_as_uint(reinterpret_cast<float*>(&vv[0])[j]),
                                                              __float_as_uint(reinterpret_cast<float*>(&vv[1])[j]),
                                                              __float_as_uint(reinterpret_cast<float*>(&vv[2])[j])));
                hV ^= ((c_V >> j) & 1u) ? h : 0ULL;
            }
        }
        __shared__ uint64_t sT[4], sV[4];
        __shared__ uint32_t sC[4];
        if (CKS_T) hT = wave_xor(hT);
        if (CKS_V) hV = wave_xor(hV);
        if (lane == 0) { sT[wave] = hT; sV[wave] = hV; sC[wave] = cnt; }
        __syncthreads();
        if (tid == 0) {
            if (CKS_T) a.part_T[t] = sT[0] ^ sT[1] ^ sT[2] ^ sT[3];
            if (CKS_V) a.part_V[t] = sV[0] ^ sV[1] ^ sV[2] ^ sV[3];
            a.part_cnt[t] = (uint64_t)sC[0] + sC[1] + sC[2] + sC[3];
        }
    }
}

// ------------------------------------------------------------------ fused request groups
// handle_requests (schedule_systems.rs:170-289) receives the WHOLE request list of a ggrs tick at once, and every request of a
// kernel-backed world is slot-local, so a run of requests  [Load?] (Save | Advance)*  executes as ONE pass: the kernel the library
// writes for the world at seal (kernel_gen.hpp, `ggrs_jit_tick`).  The hand-written kernels of rounds 1-3 (k_tick .. k_tick3) are gone:
// round 4 measured the generated kernel ahead of k_tick3 in every mode it still served (all 15 rows stored by every Save:
// 88.5 vs 115.5 us per launch at 1 M, profiles/r04a).  What is left here are the limits of a group's argument block.
constexpr int MAX_TICK_OPS = 40, MAX_TICK_SAVES = 16, MAX_TICK_STEPS = 24;

constexpr int FIN_TPB = 1024;
constexpr int GEN_MAX_CKS = MAX_COMPS;

// ------------------------------------------------------------------ k_checksum (generic)
// ComponentChecksumPlugin::update (component_checksum.rs:67-108) for any registered spec,
// plus the live count EntityChecksumPlugin needs (entity_checksum.rs:40).  grid = (tiles, n_cks)
// (n_cks == 0 still launches y = 1 for the count).  One slot per lane per step, so one ballot
// is exactly one mask word.
__global__ __launch_bounds__(TPB) void k_checksum(CksArgs a, const UnitDesc* __restrict__ units) {
    const uint32_t t = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint64_t h = 0;
    uint32_t cnt = 0;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const uint64_t e = (uint64_t)t * TILE + (uint64_t)i * TPB + tid;
        const uint64_t wi = e >> 6;
        const uint64_t aw = *reinterpret_cast<const uint64_t*>(a.state + a.off_alive + wi * 8);
        if (k == 0 && lane == 0) cnt += (uint32_t)__popcll(aw);
        if (a.n_cks == 0) continue;
        const uint64_t pw = *reinterpret_cast<const uint64_t*>(a.state + a.off_present[k] + wi * 8);
        if (((aw & pw) >> (e & 63)) & 1ULL) {
            SeaStream s;
            const uint32_t n = a.n_units[k], ub = a.unit_base[k];
#pragma unroll 1
            for (uint32_t u = 0; u < n; ++u) {
                const UnitDesc ud = units[ub + u];
                const uint8_t* p = a.state + col_at(ud.off, ud.ts, ud.wb, e);
                const uint64_t v = ud.wb == 4 ? (uint64_t)*reinterpret_cast<const uint32_t*>(p) : (ud.wb == 8 ? *reinterpret_cast<const uint64_t*>(p)
                                 : (ud.wb == 2 ? (uint64_t)*reinterpret_cast<const uint16_t*>(p) : (uint64_t)*p));
                s.write(v, ud.wb);
            }
            h ^= sea_pair(e, s.finish());
        }
    }
    __shared__ uint64_t sH[4];
    __shared__ uint32_t sC[4];
    h = wave_xor(h);
    if (lane == 0) { sH[wave] = h; sC[wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
        if (a.n_cks) a.parts[(uint64_t)k * a.part_stride + t] = sH[0] ^ sH[1] ^ sH[2] ^ sH[3];
        if (k == 0) a.part_cnt[t] = (uint64_t)sC[0] + sC[1] + sC[2] + sC[3];
    }
}

// ------------------------------------------------------------------ generic systems
// benches/bench.rs:30-46 (increment_foos ...), tests/component_rollback.rs:24-28
__global__ __launch_bounds__(TPB) void k_add_u32(uint8_t* state, uint64_t off_alive, uint64_t off_present,
                                                 uint64_t off_col, uint32_t ts, uint32_t delta, uint64_t len) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= len) return;
    const uint64_t m = *reinterpret_cast<const uint64_t*>(state + off_alive + (e >> 6) * 8) &
                       *reinterpret_cast<const uint64_t*>(state + off_present + (e >> 6) * 8);
    if ((m >> (e & 63)) & 1ULL) {
        uint32_t* p = reinterpret_cast<uint32_t*>(state + col_at(off_col, ts, 4, e));
        *p = *p + delta;
    }
}
// tests/synctest.rs:37-44 decrease_health: saturating_sub then despawn at 0.  One slot per
// lane: the wave's ballot IS the new 64-bit liveness word.  With `defer` the despawn is
// commands.entity(e).despawn_rollback() on an unconfirmed frame (despawn.rs:114-143): the entity is
// disabled -- RollbackDespawned(frame) -- instead of freed.
struct DespawnMarks { uint64_t off_disabled, off_dframe; };   // live-only side state (outside every snapshot)
__global__ __launch_bounds__(TPB) void k_sat_sub_despawn(uint8_t* state, uint64_t off_alive, uint64_t off_present,
                                                         uint64_t off_col, uint32_t ts, uint32_t amount, uint64_t len_pad64,
                                                         int defer, int32_t frame, DespawnMarks dm) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= len_pad64) return;                       // whole waves only (len padded to 64)
    const uint64_t aw = *reinterpret_cast<const uint64_t*>(state + off_alive + (e >> 6) * 8);
    const uint64_t pw = *reinterpret_cast<const uint64_t*>(state + off_present + (e >> 6) * 8);
    bool alive = (aw >> (e & 63)) & 1ULL;
    bool killed = false;
    if (alive && ((pw >> (e & 63)) & 1ULL)) {
        uint32_t* p = reinterpret_cast<uint32_t*>(state + col_at(off_col, ts, 4, e));
        const uint32_t v = *p >= amount ? *p - amount : 0u;
        *p = v;
        if (v == 0) { alive = false; killed = true; }
    }
    const uint64_t nw = __ballot(alive);
    if ((threadIdx.x & 63u) == 0 && nw != aw) *reinterpret_cast<uint64_t*>(state + off_alive + (e >> 6) * 8) = nw;
    if (defer) {
        const uint64_t kw = __ballot(killed);
        if (killed) *reinterpret_cast<int32_t*>(state + dm.off_dframe + e * 4) = frame;
        if ((threadIdx.x & 63u) == 0 && kw) *reinterpret_cast<uint64_t*>(state + dm.off_disabled + (e >> 6) * 8) |= kw;
    }
}

// ------------------------------------------------------------------ box_game (examples/box_game/box_game.rs)
// move_cube_system (box_game.rs:154-206), one slot per lane.  Query<(&mut Transform, &mut Velocity, &Player),
// With<Rollback>>: the entity needs all three components.  Every operation is a single correctly rounded IEEE
// op in the reference's order (Rust never contracts; the build uses -ffp-contract=off and HIP's default
// correctly rounded fp32 divide / sqrt); glam's Vec3::clamp_length_max is
//   len_sq = x*x + y*y + z*z;  if len_sq > max*max { max * (v / sqrt(len_sq)) } else { v }
// and f32::clamp is two compares.  friction_pow = FRICTION.powf(dt), computed by the host's libm.
struct BoxMoveArgs {
    uint8_t* state;
    uint64_t off_alive, off_pT, off_pV, off_pP;
    uint64_t off_t[3], off_v[3], off_handle;
    uint32_t ts_t, ts_v, ts_handle, pad0;   // tile strides of the three components' columns
    uint64_t len;
    uint32_t dt_bits, friction_pow_bits;
    float accel, max_speed, half_width;
    uint32_t n_inputs;
    uint8_t inputs[16];
};
__global__ __launch_bounds__(TPB) void k_box_move(BoxMoveArgs a) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= a.len) return;
    const uint64_t wi8 = (e >> 6) * 8, b = e & 63;
    const uint64_t m = *reinterpret_cast<const uint64_t*>(a.state + a.off_alive + wi8) &
                       *reinterpret_cast<const uint64_t*>(a.state + a.off_pT + wi8) &
                       *reinterpret_cast<const uint64_t*>(a.state + a.off_pV + wi8) &
                       *reinterpret_cast<const uint64_t*>(a.state + a.off_pP + wi8);
    if (!((m >> b) & 1ULL)) return;
    const uint64_t handle = *reinterpret_cast<const uint64_t*>(a.state + col_at(a.off_handle, a.ts_handle, 8, e));
    if (handle >= a.n_inputs) return;                  // inputs[p.handle] would panic in the reference
    const uint8_t in = a.inputs[handle];
    const float dt = __uint_as_float(a.dt_bits), fp = __uint_as_float(a.friction_pow_bits);
    float* px = reinterpret_cast<float*>(a.state + col_at(a.off_t[0], a.ts_t, 4, e));
    float* pz = reinterpret_cast<float*>(a.state + col_at(a.off_t[2], a.ts_t, 4, e));
    float* py = reinterpret_cast<float*>(a.state + col_at(a.off_t[1], a.ts_t, 4, e));
    float* pvx = reinterpret_cast<float*>(a.state + col_at(a.off_v[0], a.ts_v, 4, e));
    float* pvy = reinterpret_cast<float*>(a.state + col_at(a.off_v[1], a.ts_v, 4, e));
    float* pvz = reinterpret_cast<float*>(a.state + col_at(a.off_v[2], a.ts_v, 4, e));
    float vx = *pvx, vy = *pvy, vz = *pvz, x = *px, y = *py, z = *pz;
    box_move_math(x, y, z, vx, vy, vz, in, dt, fp, a.accel, a.max_speed, a.half_width);
    *pvx = vx; *pvy = vy; *pvz = vz;
    *px = x; *py = y; *pz = z;
}

// ------------------------------------------------------------------ RollbackDespawned (snapshot/despawn.rs)
// Host-issued commands.entity(e).despawn_rollback() on an unconfirmed frame.
__global__ void k_mark_despawned(uint8_t* state, uint64_t off_alive, DespawnMarks dm, uint64_t slot, int32_t frame) {
    uint64_t* a = reinterpret_cast<uint64_t*>(state + off_alive + (slot >> 6) * 8);
    const uint64_t bitm = 1ULL << (slot & 63);
    if (!(*a & bitm)) return;
    *a &= ~bitm;
    *reinterpret_cast<uint64_t*>(state + dm.off_disabled + (slot >> 6) * 8) |= bitm;
    *reinterpret_cast<int32_t*>(state + dm.off_dframe + slot * 4) = frame;
}
// LoadWorldSystems::EntityResurrect (resurrect_entities, despawn.rs:69-87: markers > the loaded frame
// are removed) plus what EntitySnapshotPlugin::load's reconcile (entity.rs:55-99) means for
// NON-rollback components: an entity that survives the load (exists now AND is in the snapshot) or
// stays disabled keeps them; one that is freed, or re-created as a fresh Entity with the old
// RollbackId, does not have them any more.  Runs BEFORE the load's copy overwrites the live
// liveness mask (stream order).  One slot per lane: a wave's ballot is one mask word.
struct ReconcileArgs {
    uint8_t* live; const uint8_t* snap;
    uint64_t off_alive; DespawnMarks dm;
    uint32_t n_nr; int32_t frame;
    uint64_t n_slots_pad64;
    uint64_t nr_present_off[MAX_MASKS];
};
__global__ __launch_bounds__(TPB) void k_load_reconcile(ReconcileArgs a) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= a.n_slots_pad64) return;
    const uint64_t wi8 = (e >> 6) * 8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t dis = *reinterpret_cast<const uint64_t*>(a.live + a.dm.off_disabled + wi8);
    const bool d = (dis >> lane) & 1ULL;
    const int32_t m = d ? *reinterpret_cast<const int32_t*>(a.live + a.dm.off_dframe + e * 4) : 0;
    const uint64_t res = __ballot(d && m > a.frame);              // despawned_frame > rollback_frame
    if (lane == 0) {
        const uint64_t alive_live = *reinterpret_cast<const uint64_t*>(a.live + a.off_alive + wi8);
        const uint64_t s_alive = *reinterpret_cast<const uint64_t*>(a.snap + a.off_alive + wi8);   // zero beyond the snapshot's len
        const uint64_t dis_after = dis & ~res;
        if (res) *reinterpret_cast<uint64_t*>(a.live + a.dm.off_disabled + wi8) = dis_after;
        const uint64_t keep = ((alive_live | res) & s_alive) | dis_after;
        for (uint32_t k = 0; k < a.n_nr; ++k) {
            uint64_t* p = reinterpret_cast<uint64_t*>(a.live + a.nr_present_off[k] + wi8);
            const uint64_t v = *p;
            if (v & ~keep) *p = v & keep;
        }
    }
}
// AdvanceWorldSystems::DespawnConfirmed (despawn_confirmed_entities, despawn.rs:89-112): disabled
// entities whose marked frame is <= ConfirmedFrameCount are freed for good.
__global__ __launch_bounds__(TPB) void k_despawn_confirmed(uint8_t* live, DespawnMarks dm, int32_t confirmed, uint64_t n_slots_pad64) {
    const uint64_t e = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (e >= n_slots_pad64) return;
    const uint64_t wi8 = (e >> 6) * 8;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t dis = *reinterpret_cast<const uint64_t*>(live + dm.off_disabled + wi8);
    const bool d = (dis >> lane) & 1ULL;
    const int32_t m = d ? *reinterpret_cast<const int32_t*>(live + dm.off_dframe + e * 4) : 0;
    const uint64_t gone = __ballot(d && m <= confirmed);
    if (lane == 0 && gone) *reinterpret_cast<uint64_t*>(live + dm.off_disabled + wi8) = dis & ~gone;
}

// System-scope release + acquire on whatever CU / XCD the wave lands on: `buffer_wbl2 sc0 sc1` writes the XCD's dirty L2 lines back,
// `buffer_inv sc0 sc1` drops its clean ones.  2048 single-wave workgroups cover all 8 XCDs (workgroup b lands on XCD b % 8).
__global__ void k_flush_l2() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, ""); }

// ------------------------------------------------------------------ spawn / mask edits
// Set liveness + presence bits for slots [first, first+count) (Rollback on_add hook,
// rollback.rs:45-59) and clear the live-only masks a fresh entity does not carry (non-rollback
// components outside its bundle, a stale RollbackDespawned marker).  One mask word per thread; each
// word is owned by exactly one thread.
struct MaskOffs { uint64_t off[MAX_MASKS]; };
__global__ __launch_bounds__(TPB) void k_set_mask_range(uint8_t* state, uint64_t first, uint64_t count,
                                                        uint32_t n_masks, MaskOffs mask_off_set,
                                                        uint32_t n_clear, MaskOffs mask_off_clear) {
    const uint64_t w_first = first >> 6, w_last = (first + count - 1) >> 6;
    const uint64_t wi = w_first + (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (wi > w_last) return;
    const uint64_t lo = wi * 64, hi = lo + 64;
    const uint64_t a = first > lo ? first : lo, b = (first + count) < hi ? (first + count) : hi;
    const uint32_t nb = (uint32_t)(b - a);
    const uint64_t bits = (nb == 64 ? ~0ULL : ((1ULL << nb) - 1ULL)) << (a - lo);
    for (uint32_t m = 0; m < n_masks; ++m) {
        uint64_t* p = reinterpret_cast<uint64_t*>(state + mask_off_set.off[m] + wi * 8);
        *p |= bits;
    }
    for (uint32_t m = 0; m < n_clear; ++m) {
        uint64_t* p = reinterpret_cast<uint64_t*>(state + mask_off_clear.off[m] + wi * 8);
        *p &= ~bits;
    }
}
__global__ void k_edit_mask_bit(uint8_t* state, uint64_t mask_off, uint64_t slot, int value) {
    uint64_t* p = reinterpret_cast<uint64_t*>(state + mask_off + (slot >> 6) * 8);
    if (value) *p |= 1ULL << (slot & 63); else *p &= ~(1ULL << (slot & 63));
}
// Fill one column over [first, first+count) with a constant word (component defaults).
__global__ __launch_bounds__(TPB) void k_fill_col(uint8_t* state, uint64_t col_off, uint32_t ts, uint32_t word_bytes,
                                                  uint64_t first, uint64_t count, uint64_t value) {
    const uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= count) return;
    uint8_t* p = state + col_at(col_off, ts, word_bytes, first + i);
    if (word_bytes == 4) *reinterpret_cast<uint32_t*>(p) = (uint32_t)value;
    else if (word_bytes == 8) *reinterpret_cast<uint64_t*>(p) = value;
    else if (word_bytes == 2) *reinterpret_cast<uint16_t*>(p) = (uint16_t)value;
    else *p = (uint8_t)value;
}
// spawn_particles (particles.rs:258-270) payload: Velocity(vx, vy, 0.0), Ttl(ttl); Transform gets
// its default through k_fill_col.  Also emits checksum partials for the new rows so a fused
// step's pending partials stay complete.
struct SpawnArgs {
    uint8_t* state;
    uint64_t off_t[3], off_v[3], off_ttl;
    uint32_t ts;                          // tile stride of the rollback word columns
    uint32_t t_default[3];
    const float* vx; const float* vy;
    uint64_t first, count, ttl;
    uint64_t* part_T; uint64_t* part_V; uint64_t* part_cnt;   // slots [0, gridDim.x)
    int cks_T, cks_V;
};
__global__ __launch_bounds__(TPB) void k_spawn_particles(SpawnArgs a) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * TPB + tid;
    uint64_t hT = 0, hV = 0;
    uint32_t cnt = 0;
    if (i < a.count) {
        const uint64_t e = a.first + i;
        const float vx = a.vx[i], vy = a.vy[i];
        *reinterpret_cast<float*>(a.state + col_at(a.off_v[0], a.ts, 4, e)) = vx;
        *reinterpret_cast<float*>(a.state + col_at(a.off_v[1], a.ts, 4, e)) = vy;
        *reinterpret_cast<float*>(a.state + col_at(a.off_v[2], a.ts, 4, e)) = 0.0f;
        *reinterpret_cast<uint64_t*>(a.state + col_at(a.off_ttl, a.ts, 8, e)) = a.ttl;
        if (a.cks_T) hT = sea_pair(e, sea_inner3(a.t_default[0], a.t_default[1], a.t_default[2]));
        if (a.cks_V) hV = sea_pair(e, sea_inner3(__float_as_uint(vx), __float_as_uint(vy), 0u));
        cnt = 1;
    }
    __shared__ uint64_t sT[4], sV[4];
    __shared__ uint32_t sC[4];
    hT = wave_xor(hT); hV = wave_xor(hV);
    cnt = (uint32_t)__popcll(__ballot(cnt));
    if (lane == 0) { sT[wave] = hT; sV[wave] = hV; sC[wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
        a.part_T[blockIdx.x] = sT[0] ^ sT[1] ^ sT[2] ^ sT[3];
        a.part_V[blockIdx.x] = sV[0] ^ sV[1] ^ sV[2] ^ sV[3];
        a.part_cnt[blockIdx.x] = (uint64_t)sC[0] + sC[1] + sC[2] + sC[3];
    }
}

// ------------------------------------------------------------------ k_gen_finalize
// Fold of k_tick_gen's per-wave partials: one 1024-thread workgroup per Save; any number of checksummed components.
struct GenFinArgs {
    const uint64_t* parts; uint32_t part_stride, n_parts, n_cks, n_saves;     // grid = n_saves x members (batch of identical groups)
    uint64_t save_len[MAX_TICK_SAVES];                                        // RollbackOrdered::len at each Save of the group (a fused spawn grows it)
    uint64_t* out;
    uint64_t* done; uint64_t seq;                                             // completion tag per workgroup in pinned host memory (nullptr: none), see read_back
    uint64_t* out2;                                                           // a second copy of every Checksum(u128) in DEVICE memory (nullptr: none): what the fan-out's all-gather sends from (host_fanout.hpp)
    const uint8_t* mtab; uint32_t mstride, moff_save_len;                     // batch members with records (kernel_gen.hpp): member m's save_len[k] = *(u64*)(mtab + m * mstride + moff_save_len + 8 k)
    const uint64_t* dev_save_len;                                             // spawns decided on the device: RollbackOrdered::len at Save k as the launch left it (nullptr: save_len above)
};
__global__ __launch_bounds__(FIN_TPB) void k_gen_finalize(GenFinArgs f) {
    // rows = the n_cks component XORs + the live count; the 16 waves split over the rows so that every row's loads are in
    // flight together (one latency round, not one per row)
    const uint32_t k = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t NW = FIN_TPB / 64;
    __shared__ uint64_t acc[GEN_MAX_CKS + 1];
    const uint32_t nc = f.n_cks + 1u;
    if (tid < nc) acc[tid] = 0;
    __syncthreads();
    const uint32_t wpr = nc >= NW ? 1u : NW / nc;                     // waves per row
    for (uint32_t row = wave / wpr; row < nc; row += NW / wpr) {
        const uint64_t* p = f.parts + ((uint64_t)k * nc + row) * f.part_stride;
        const bool is_cnt = row == f.n_cks;
        uint64_t x = 0, sum = 0;
        // the rows sit in other XCDs' L2 / HBM: the fold is latency-bound unless every load of a lane is in flight at once --
        // 16 per trip (a 1 M-entity world: 3907 values per row over 5 waves = 13 per lane: ONE round trip)
        constexpr int INFL = 16;
        for (uint32_t i0 = (wave % wpr) * 64u + lane; i0 < f.n_parts; i0 += (uint32_t)INFL * wpr * 64u) {
            uint64_t v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; ++u) { const uint32_t i = i0 + (uint32_t)u * wpr * 64u; v[u] = i < f.n_parts ? p[i] : 0ULL; }
#pragma unroll
            for (int u = 0; u < INFL; ++u) { x ^= v[u]; sum += v[u]; }
        }
        x = wave_xor(x);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane == 0) {
            if (is_cnt) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[row]), (unsigned long long)sum);
            else atomicXor(reinterpret_cast<unsigned long long*>(&acc[row]), (unsigned long long)x);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t total = 0;
        for (uint32_t c = 0; c < f.n_cks; ++c) total ^= sea_one(acc[c]);      // component_checksum.rs:92-95
        const uint64_t len_k = f.dev_save_len ? f.dev_save_len[k % f.n_saves] : f.mtab ? *reinterpret_cast<const uint64_t*>(f.mtab + (uint64_t)(k / f.n_saves) * f.mstride + f.moff_save_len + 8u * (k % f.n_saves)) : f.save_len[k % f.n_saves];
        total ^= sea_pair(acc[f.n_cks], len_k);                               // entity_checksum.rs:29-52; XOR fold checksum.rs:88-99
        f.out[2 * (uint64_t)k] = total; f.out[2 * (uint64_t)k + 1] = 0;
        if (f.out2) { f.out2[2 * (uint64_t)k] = total; f.out2[2 * (uint64_t)k + 1] = 0; }
        // the result first, then the tag the waiting host polls (both in the same pinned allocation; release at system scope orders them)
        if (f.done) __hip_atomic_store(f.done + k, f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------ k_ff_fold
// Fold-forward without a following request-group launch (ff_flush): one 256-thread workgroup per chunk of a partial row of the LAST launch
// (device_prelude.hpp ff_fold_row) -- {value, tag} as one 16-byte cell into pinned host memory.  Workgroup b = row b / split, chunk b % split.
struct FfArgs { const uint64_t* rows; uint64_t* out; uint64_t seq; uint32_t nvals, g, stride, istride, nc1, split; };
__global__ __launch_bounds__(TPB) void k_ff_fold(FfArgs f) {
    const uint32_t b = blockIdx.x;
    if (b >= f.nvals * f.split) return;
    const uint32_t row = b / f.split, ck = b % f.split, per = (f.g + f.split - 1u) / f.split;
    ff_fold_row(f.rows + (uint64_t)row * f.stride, f.istride, ck * per, min(f.g, (ck + 1u) * per), (row % f.nc1) == f.nc1 - 1u, f.out + 2u * b, f.seq);
}

}  // namespace ggrs