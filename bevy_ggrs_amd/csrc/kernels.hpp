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
constexpr int MAX_ROWS = 96;   // 4 KiB copy rows per tile (a 4-byte column = 1 row, 8-byte = 2)
constexpr int MAX_MASKS = 17;  // alive + one presence mask per component
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
    uint32_t pad;
};
struct CopyPlan {
    uint32_t n_rows, n_masks;
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

struct UnitDesc { uint64_t off; uint32_t stride; uint32_t ts; };   // u32 unit e at col_at(off, ts, stride, e)
struct CksArgs {          // generic component checksum
    const uint8_t* state;
    uint64_t off_alive;
    uint32_t n_cks; uint32_t part_stride;
    uint64_t off_present[16];
    uint32_t n_units[16];
    uint32_t unit_base[16];
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
constexpr int MAX_CKS = 16;

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
        uint32_t r = 0;
        for (; r + 8 <= n_rows; r += 8) copy_rows<8, NT>(src, dst, plan, r, t, tid);
        if (r + 4 <= n_rows) { copy_rows<4, NT>(src, dst, plan, r, t, tid); r += 4; }
        if (r + 2 <= n_rows) { copy_rows<2, NT>(src, dst, plan, r, t, tid); r += 2; }
        if (r < n_rows) copy_rows<1, NT>(src, dst, plan, r, t, tid);
    } else {
        for (uint32_t r = 0; r < n_rows; ++r) {
            const RowDesc rd = plan.row[r];
            const uint64_t pos = wtile_off(t, rd.tile_stride, rd.word_bytes) + rd.roff + (uint64_t)tid * 16;
            const uint64_t slot0 = (uint64_t)t * TILE + (rd.roff + tid * 16u) / rd.word_bytes;   // first slot of this lane's 16 bytes
            if (slot0 < len)
                *reinterpret_cast<uint4*>(dst + rd.col_off + pos) = *reinterpret_cast<const uint4*>(src + rd.col_off + pos);
        }
    }
    // masks: 16 u64 words per tile per mask (copied whole, so stale bits beyond the source's
    // len are cleared in the destination)
    if (tid < 16u * plan.n_masks) {
        const uint32_t m = tid >> 4, wi = tid & 15u;
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

// ------------------------------------------------------------------ k_tick (fused request group)
// handle_requests (schedule_systems.rs:170-289) receives the WHOLE request list of a ggrs tick at
// once, and every request of the particles world is tile-local: LoadWorld, SaveWorld's snapshot
// copy, the per-entity half of the checksum and the GgrsSchedule step only ever touch a slot's own
// words.  So a run of requests  [Load?] (Save | Advance)*  executes as ONE pass over the tiles:
//   * the tile's state is read ONCE (from the ring slot being loaded, or from the live block),
//   * it stays in registers while the ops are replayed in request order -- a Save stores the
//     registers to its ring slot and emits that frame's checksum partials, an Advance runs
//     update_particles + despawn_particles on the registers,
//   * the live block is written ONCE at the end.
// Compulsory HBM traffic of a SyncTest tick at depth D drops from 1656 B/entity (one kernel per
// request: every Save re-reads the live block, every Advance re-reads and re-writes it) to
// 60 (read snapshot) + 60*D (write D snapshots) + 60 (write live) = 600 B at D = 8.  No work is
// skipped: every snapshot is written in full, every checksum is computed, every frame stepped.
//
// Aliasing: a Save's ring slot may be the slot the group was loaded from (ring depth 1); this is
// safe because every location is read by the lane that later writes it, and all reads of a
// location precede its first write (pointers are deliberately not __restrict__).
constexpr int MAX_TICK_OPS = 40, MAX_TICK_SAVES = 16, MAX_TICK_STEPS = 24;
// In-kernel checksum fold of a fused group (tick_fold below): one row of partials per workgroup, one arrival ticket,
// the last workgroup to arrive writes every Save's Checksum(u128).
struct FoldArgs {
    uint64_t* wg_parts;                    // [gridDim.x][n_saves * (n_comp + 1)]: per Save the XOR of each checksummed component, then the live count
    uint32_t* ticket;                      // arrival counter, zero between launches
    uint64_t* out;                         // {lo, hi} per Save (pinned, device-mapped host memory)
    uint32_t n_comp, comp_mask;            // component slots per Save; bit j: slot j is a registered checksum (contributes a part)
};
struct RowLite { uint64_t col_off; uint32_t roff; uint32_t tile_stride; uint32_t word_bytes; uint32_t pad; };
struct TickArgs {
    const uint8_t* src;                    // ring slot (group starts with LoadGameState) or live
    uint8_t* live;
    uint8_t* save_dst[MAX_TICK_SAVES];     // nullptr: ring depth 0, checksum only
    int32_t save_frame[MAX_TICK_SAVES];
    uint32_t dt_bits[MAX_TICK_STEPS];
    uint64_t op_bits;                      // bit i = 1: op i is an Advance, 0: a Save (request order)
    uint32_t n_ops, n_saves, n_steps, src_is_live;
    uint64_t len;
    uint32_t nt_load, skip_live;          // skip_live: the live block is overwritten before anyone reads it (a LoadGameState follows): do not write it
    uint64_t off_alive, off_pT, off_pV, off_pL, off_t[3], off_v[3], off_ttl;
    float g[3];
    uint32_t n_rest_rows, n_rest_masks, part_stride, ts;   // ts: tile stride of the rollback word columns
    uint32_t dp_s;                         // depth-parallel k_tick1: outputs (Saves, then the live world) per workgroup role
    uint64_t* parts;                       // [n_saves][3 = T,V,count][part_stride], one entry per WAVE (folded by k_tick_finalize)
    uint64_t rest_mask_off[MAX_MASKS];     // presence masks of components the schedule does not touch
    RowLite rest[MAX_ROWS];                // word rows the schedule does not touch
};

// Pins a wave-uniform pointer into an SGPR pair so that `sgpr_base(p) + lane_offset_u32` selects the
// saddr form of global_load/global_store (no 64-bit VALU address arithmetic per access).  The value
// comes back as an explicit global (address space 1) pointer: laundering a generic pointer through
// inline asm would otherwise make the compiler fall back to flat_* instructions.
static_assert(sizeof(TickArgs) <= 4096, "kernel argument segment limit");
#define GGRS_GLOBAL __attribute__((address_space(1)))
typedef GGRS_GLOBAL uint8_t g_u8;
__device__ __forceinline__ g_u8* sgpr_base(const uint8_t* p) {
    uint64_t x = reinterpret_cast<uint64_t>(p);
    asm volatile("" : "+s"(x));
    return (g_u8*)x;
}

// 16-byte store to `base + lo` (base wave-uniform, lo a 32-bit lane offset).  The non-temporal form is written
// as inline asm: __builtin_nontemporal_store on the same expression makes hipcc fall back to a 64-bit VGPR
// address that it recomputes into ONE register pair before every store, which serialises the whole store
// burst behind VALU address arithmetic (measured: NT 6 % slower than plain stores; saddr-form NT is faster).
template <bool NT, class V>
__device__ __forceinline__ void st16(g_u8* base, uint32_t lo, const V& v) {
    static_assert(sizeof(V) == 16, "16-byte register tuple");
    const u32x4 x = reinterpret_cast<const u32x4&>(v);
    if (NT) {
        const uint64_t b = reinterpret_cast<uint64_t>(base);
        asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(lo), "v"(x), "s"(b) : "memory");
    } else {
        *(GGRS_GLOBAL u32x4*)(base + lo) = x;
    }
}
__device__ __forceinline__ void st8(g_u8* p, uint64_t v) { *(GGRS_GLOBAL uint64_t*)p = v; }

template <int B, bool NT>
__device__ __forceinline__ void fan_rows(const TickArgs& a, uint32_t r0, uint32_t t, uint32_t tid) {
    u32x4 v[B];
    uint64_t pos[B];                          // wave-uniform part of the address (SGPRs)
    const uint32_t lo = tid * 16u;            // per-lane part
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const RowLite rd = a.rest[r0 + j];
        pos[j] = rd.col_off + wtile_off(t, rd.tile_stride, rd.word_bytes) + rd.roff;
    }
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = *reinterpret_cast<const u32x4*>(a.src + pos[j] + lo);
    // gfx9 counts loads AND stores in vmcnt: land the loads once here, or the compiler throttles
    // every store of the fan-out loop behind a conservative vmcnt(B-1)
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0) expcnt(7) lgkmcnt(15)
    for (uint32_t k = 0; k < a.n_saves; ++k) {
        uint8_t* dst = a.save_dst[k];
        if (!dst) continue;
#pragma unroll
        for (int j = 0; j < B; ++j) {
            st16<NT>(sgpr_base(dst + pos[j]), lo, v[j]);
        }
    }
    if (!a.src_is_live && !a.skip_live) {
#pragma unroll
        for (int j = 0; j < B; ++j) st16<false>(sgpr_base(a.live + pos[j]), lo, v[j]);
    }
}

// RESTL > 0: the (up to RESTL) word rows the schedule never touches stay in registers too and every Save stores
// its snapshot's full tile (schedule-owned rows + rest rows) together, instead of the up-front fan-out: fewer
// workgroups resident (4 * RESTL more VGPRs), each snapshot's tile written as one burst.  Measured 128-129 us vs
// 126-136 us (the plain variant is bimodal with the arena's placement, profiles/README.md); default when the
// world has at most RESTL such rows, GGRS_TICK_REST=0 selects the fan-out variant.
// WPB = waves per workgroup.  Every wave owns one 256-slot quarter of a tile and shares nothing with the other
// waves (no LDS, no barrier), so the same code runs as 4-wave workgroups (one tile each) for big worlds and as
// single-wave workgroups for small ones: four times as many workgroups to spread over the 256 CUs, still 16 bytes
// per lane per access.
template <bool CKS_T, bool CKS_V, bool NT, int RESTL = 0, int WPB = 4>
__global__ __launch_bounds__(WPB * 64) void k_tick(TickArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    // readfirstlane: the wave index is uniform, and the compiler must know it (uniform bases live in SGPRs)
    const uint32_t gw = blockIdx.x * WPB + (WPB == 1 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));   // global wave id == 256-slot quarter tile
    const uint32_t t = gw >> 2, wave = gw & 3u;                   // its tile, and which quarter of it
    const uint32_t tid = wave * 64u + lane;                       // lane index inside the tile
    const bool in_len = (uint64_t)gw * 256u < a.len;              // wave-uniform
    const uint64_t e0 = (uint64_t)t * TILE + (uint64_t)tid * 4;   // first of this lane's 4 slots
    // every access below is "uniform 64-bit base (block + column row + tile offset) + 32-bit lane
    // offset", i.e. the saddr form of global_load/store
    const uint64_t toff = wtile_off(t, a.ts, 4), toff8 = wtile_off(t, a.ts, 8);   // this tile's rows of a 4- / 8-byte column inside a block
    const uint32_t o4 = tid * 16u, o8 = tid * 32u;                // lane offsets inside a 4- / 8-byte column's tile rows
    const uint32_t w0 = t * 16u + wave * 4u;                      // first mask word of this wave
    const uint32_t sh = (lane & 15u) * 4;
    const uint32_t wi8 = (w0 + (lane >> 4)) * 8u;                 // byte offset of this lane's mask word

    // ---- every load of the schedule-owned state, back to back
    const uint64_t alive_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_alive + wi8);
    const uint64_t pT_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pT + wi8);
    const uint64_t pV_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pV + wi8);
    const uint64_t pL_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pL + wi8);
    float4 tx[3], vv[3];
    ulonglong2 tl[2];
    // the source block is read exactly once: a.nt_load (A/B knob GGRS_TICK_NTLOAD) marks the loads non-temporal
    auto ld16 = [&](const uint8_t* p) -> u32x4 {
        return a.nt_load ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
    };
    if (in_len) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_t[k] + toff + o4); tx[k] = reinterpret_cast<const float4&>(x); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_v[k] + toff + o4); vv[k] = reinterpret_cast<const float4&>(x); }
        { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + o8); tl[0] = reinterpret_cast<const ulonglong2&>(x); }
        { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + 16 + o8); tl[1] = reinterpret_cast<const ulonglong2&>(x); }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { tx[k] = make_float4(0, 0, 0, 0); vv[k] = make_float4(0, 0, 0, 0); }
        tl[0] = make_ulonglong2(0, 0); tl[1] = make_ulonglong2(0, 0);
    }

    // ---- state the schedule never touches: read once, fan out to every snapshot (+ live on load)
    if (lane < 4u * a.n_rest_masks) {                             // this wave's 4 words of every such mask
        const uint32_t m = lane >> 2, mw = lane & 3u;
        const uint64_t o = a.rest_mask_off[m] + ((uint64_t)gw * 4 + mw) * 8;
        const uint64_t v = *reinterpret_cast<const uint64_t*>(a.src + o);
        for (uint32_t k = 0; k < a.n_saves; ++k)
            if (a.save_dst[k]) *reinterpret_cast<uint64_t*>(a.save_dst[k] + o) = v;
        if (!a.src_is_live && !a.skip_live) *reinterpret_cast<uint64_t*>(a.live + o) = v;
    }
    u32x4 restv[RESTL > 0 ? RESTL : 1];
    uint64_t restpos[RESTL > 0 ? RESTL : 1];
    if (RESTL > 0) {
#pragma unroll
        for (int j = 0; j < RESTL; ++j) {
            restv[j] = u32x4{0, 0, 0, 0}; restpos[j] = 0;
            if ((uint32_t)j < a.n_rest_rows) {                   // wave-uniform
                const RowLite rd = a.rest[j];
                restpos[j] = rd.col_off + wtile_off(t, rd.tile_stride, rd.word_bytes) + rd.roff;
                if (in_len) restv[j] = ld16(a.src + restpos[j] + tid * 16u);
            }
        }
    } else if (in_len && (a.n_saves || !a.src_is_live)) {
        // batches of up to 8 rows: while this phase runs only the schedule-owned state is live in
        // registers (the hash temporaries come later), so 32 more VGPRs are free.  The particles world
        // has 7 such rows: ONE batch, one wait shared with the state loads above, then stores only.
        const uint32_t n_rows = a.n_rest_rows;
        for (uint32_t r = 0; r < n_rows; r += 8) {
            switch (n_rows - r) {
            case 1: fan_rows<1, NT>(a, r, t, tid); break;
            case 2: fan_rows<2, NT>(a, r, t, tid); break;
            case 3: fan_rows<3, NT>(a, r, t, tid); break;
            case 4: fan_rows<4, NT>(a, r, t, tid); break;
            case 5: fan_rows<5, NT>(a, r, t, tid); break;
            case 6: fan_rows<6, NT>(a, r, t, tid); break;
            case 7: fan_rows<7, NT>(a, r, t, tid); break;
            default: fan_rows<8, NT>(a, r, t, tid); break;
            }
        }
    }

    uint32_t alive4 = (uint32_t)(alive_w >> sh) & 0xFu;
    const uint32_t n_T = (uint32_t)(pT_w >> sh) & 0xFu, n_V = (uint32_t)(pV_w >> sh) & 0xFu,
                   n_L = (uint32_t)(pL_w >> sh) & 0xFu;

    // diffuse(K0 ^ order), order == slot (RollbackOrdered::order): shared by both components and all Saves
    uint64_t ordB[4];
    if (CKS_T || CKS_V) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ordB[j] = sea_order_lane(e0 + j);
    }

    // SaveWorld, per-entity half of the checksum (component_checksum.rs:77-90) of the registers as they
    // are now; one partial per wave.
    uint32_t si = 0, sj = 0;
    auto hash_save = [&](uint32_t cnt) {
        uint64_t hT = 0, hV = 0;
        if (CKS_T || CKS_V) {
            const uint32_t c_T = CKS_T ? (alive4 & n_T) : 0u, c_V = CKS_V ? (alive4 & n_V) : 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (CKS_T) {
                    const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&tx[0])[j]),
                                                                        __float_as_uint(reinterpret_cast<float*>(&tx[1])[j]),
                                                                        __float_as_uint(reinterpret_cast<float*>(&tx[2])[j])));
                    hT ^= ((c_T >> j) & 1u) ? h : 0ULL;
                }
                if (CKS_V) {
                    const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&vv[0])[j]),
                                                                        __float_as_uint(reinterpret_cast<float*>(&vv[1])[j]),
                                                                        __float_as_uint(reinterpret_cast<float*>(&vv[2])[j])));
                    hV ^= ((c_V >> j) & 1u) ? h : 0ULL;
                }
            }
            if (CKS_T) hT = wave_xor(hT);
            if (CKS_V) hV = wave_xor(hV);
        }
        if (lane == 0) {
            // plain per-wave partial stores: agent-scope atomics (tried: 64 accumulator copies + last-block
            // fold) cost ~1 ns EACH chip-wide on gfx950 -- 94k of them added 90 us to a 134 us kernel
            uint64_t* p = a.parts + (uint64_t)si * 3 * a.part_stride + gw;
            p[0] = hT; p[a.part_stride] = hV; p[2 * (uint64_t)a.part_stride] = cnt;
        }
    };
    // All waves of the chip start together and run the same op sequence, so without a stagger every
    // wave would be storing at the same time and hashing at the same time (memory idle while the ALUs
    // hash).  Odd waves hash a Save BEFORE storing it, even waves after: at any moment half the waves
    // of a SIMD feed the memory pipe while the other half multiply.
    const bool hash_first = (wave & 1u) != 0;                     // wave-uniform

    // every load issued above has to land before the first op anyway; saying so explicitly keeps the
    // compiler from guarding the op loop / the final live write with conservative vmcnt waits (gfx9
    // counts stores in vmcnt too: such a wait would drain every snapshot store in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0) expcnt(7) lgkmcnt(15)

    for (uint32_t i = 0; i < a.n_ops; ++i) {
        if (!((a.op_bits >> i) & 1ULL)) {
            // ---------------- SaveWorld: snapshot (component_snapshot.rs:66-84, entity.rs:39-51)
            uint8_t* dst = a.save_dst[si];
            const uint64_t b0 = __ballot((alive4 >> 0) & 1u), b1 = __ballot((alive4 >> 1) & 1u),
                           b2 = __ballot((alive4 >> 2) & 1u), b3 = __ballot((alive4 >> 3) & 1u);
            const uint32_t cnt = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            if (hash_first) hash_save(cnt);
            if (dst) {
                if (in_len) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        st16<NT>(sgpr_base(dst + a.off_t[k] + toff), o4, tx[k]);
                        st16<NT>(sgpr_base(dst + a.off_v[k] + toff), o4, vv[k]);
                    }
                    st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8, tl[0]);
                    st16<NT>(sgpr_base(dst + a.off_ttl + toff8 + 16), o8, tl[1]);
                    if (RESTL > 0) {
#pragma unroll
                        for (int j = 0; j < RESTL; ++j) if ((uint32_t)j < a.n_rest_rows) st16<NT>(sgpr_base(dst + restpos[j]), tid * 16u, restv[j]);
                    }
                }
                uint64_t mine = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                        (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
                    if (lane == (uint32_t)w) mine = nw;
                }
                if (lane < 4) st8(sgpr_base(dst + a.off_alive) + (w0 + lane) * 8u, mine);
                if ((lane & 15u) == 0) {
                    st8(sgpr_base(dst + a.off_pT) + wi8, pT_w);
                    st8(sgpr_base(dst + a.off_pV) + wi8, pV_w);
                    st8(sgpr_base(dst + a.off_pL) + wi8, pL_w);
                }
                if (gw == 0 && lane == 0) {
                    Header h; h.len = a.len; h.frame = a.save_frame[si]; h.pad0 = 0; h.active = 0;
                    h.checksum[0] = 0; h.checksum[1] = 0;
                    *reinterpret_cast<Header*>(dst) = h;
                }
            }
            if (!hash_first) hash_save(cnt);
            ++si;
        } else {
            // ---------------- AdvanceWorld: update_particles + despawn_particles (particles.rs:272-289)
            const float dt = __uint_as_float(a.dt_bits[sj]);
            ++sj;
            const uint32_t m_upd = alive4 & n_T & n_V;     // Query<(&mut Transform, &mut Velocity)>
            const uint32_t m_ttl = alive4 & n_L;           // Query<(Entity, &mut Ttl)>
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float gd = __fmul_rn(a.g[k], dt);    // gravity * time_step
                float* x = reinterpret_cast<float*>(&tx[k]);
                float* v = reinterpret_cast<float*>(&vv[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = (m_upd >> j) & 1u;
                    const float nv = __fadd_rn(v[j], gd);                     // **velocity += ...
                    const float nx = __fadd_rn(x[j], __fmul_rn(nv, dt));      // translation += **velocity * time_step
                    v[j] = on ? nv : v[j];
                    x[j] = on ? nx : x[j];
                }
            }
            uint32_t kill = 0;
            uint64_t* q = reinterpret_cast<uint64_t*>(&tl[0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = (m_ttl >> j) & 1u;
                const uint64_t nq = q[j] - 1;              // usize, wrapping
                q[j] = on ? nq : q[j];
                kill |= (on && nq == 0) ? (1u << j) : 0u;
            }
            alive4 &= ~kill;                               // despawn is deferred to the end of the frame
        }
    }

    // ---- the live block, written once
    if ((!a.src_is_live || a.n_steps) && !a.skip_live) {
        if (in_len) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                st16<false>(sgpr_base(a.live + a.off_t[k] + toff), o4, tx[k]);
                st16<false>(sgpr_base(a.live + a.off_v[k] + toff), o4, vv[k]);
            }
            st16<false>(sgpr_base(a.live + a.off_ttl + toff8), o8, tl[0]);
            st16<false>(sgpr_base(a.live + a.off_ttl + toff8 + 16), o8, tl[1]);
            if (RESTL > 0 && !a.src_is_live) {
#pragma unroll
                for (int j = 0; j < RESTL; ++j) if ((uint32_t)j < a.n_rest_rows) st16<false>(sgpr_base(a.live + restpos[j]), tid * 16u, restv[j]);
            }
        }
        const uint64_t b0 = __ballot((alive4 >> 0) & 1u), b1 = __ballot((alive4 >> 1) & 1u),
                       b2 = __ballot((alive4 >> 2) & 1u), b3 = __ballot((alive4 >> 3) & 1u);
        uint64_t mine = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
            if (lane == (uint32_t)w) mine = nw;
        }
        if (lane < 4) st8(sgpr_base(a.live + a.off_alive) + (w0 + lane) * 8u, mine);
        if (!a.src_is_live && (lane & 15u) == 0) {
            st8(sgpr_base(a.live + a.off_pT) + wi8, pT_w);
            st8(sgpr_base(a.live + a.off_pV) + wi8, pV_w);
            st8(sgpr_base(a.live + a.off_pL) + wi8, pL_w);
        }
    }

}

// GGRS_LDS_BARRIER_DEFINED
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for every
// snapshot store still in flight (1-2 us per Save); nobody in the workgroup reads those stores back.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);      // vmcnt(63) expcnt(7) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------ k_tick2 (persistent fused request group, round 2)
// The same request-group fusion as k_tick, restructured after the round-2 store-path study (scripts/ubench3.hip,
// profiles/r02a): on MI355X the HBM write path runs closest to its ceiling when FEW waves stream stores continuously
// (a linear fill from 256 workgroups reaches 6.2-6.4 TB/s, the same fill from 1024 workgroups 4.7-5.4 TB/s), and
// k_tick's Save was a burst of 15 stores followed by a block of ~660 VALU instructions (80 u64 multiplies), so store
// issue and hashing only overlapped across waves.  Here:
//   * PERSISTENT grid: `gridDim.x` = CUs x workgroups-per-CU (host-chosen, <= the tiles); a wave walks over 256-slot
//     units u = wave id, += waves of the grid.  No second "wave" of workgroups starting its loads while the rest of
//     the chip is storing, a bounded number of store streams in flight;
//   * inside a Save the eight independent SeaHash chains (4 slots x {Transform, Velocity}) are INTERLEAVED with the
//     tile's 15 row stores -- two stores, one chain, pinned with sched_barrier -- so the store queue drains while the
//     VALU multiplies instead of after it;
//   * snapshot stores may be non-temporal (`nt`): with the order of stores now reaching DRAM in a dense sweep that is a
//     gain (ubench3: 100-105 us vs 113-115 us for the same traffic);
//   * NO finalize launch: a wave folds its partials into LDS, the workgroup publishes one 48-value row with
//     write-through (sc0 sc1) stores, takes ONE agent-scope ticket, and the last workgroup to arrive folds the rows
//     (component_checksum.rs:92-95, entity_checksum.rs:29-52, checksum.rs:88-99) and writes every Save's Checksum
//     straight to pinned host memory.  256-768 tickets per launch, not one atomic per partial.
// Register-resident rows: the 8 schedule-owned rows (translation, velocity, ttl) + up to RESTL untouched rows, which
// the layout keeps back to back behind them (4-byte words only: rest row j of tile t at rest_off + wtile_off(t) + j * 32 KiB).
constexpr uint32_t REST_ROW_STRIDE = LAYOUT_TILE * 4u;     // k_tick2: the untouched words are 4-byte words laid out back to back
struct Tick2Args {
    const uint8_t* src; uint8_t* live;
    uint8_t* save_dst[MAX_TICK_SAVES];     // nullptr: ring depth 0, checksum only
    int32_t save_frame[MAX_TICK_SAVES];
    uint32_t dt_bits[MAX_TICK_STEPS];
    uint64_t op_bits;                      // bit i = 1: op i is an Advance, 0: a Save (request order)
    uint32_t n_ops, n_saves, n_steps, src_is_live;
    uint64_t len;                          // slots of the source state == RollbackOrdered::len at every Save (a spawn ends the group)
    uint32_t n_units, ts;                  // 256-slot units to walk (covers every dirty mask word); tile stride
    uint64_t off_alive, off_pT, off_pV, off_pL, off_t[3], off_v[3], off_ttl, rest_off;
    float g[3];
    uint32_t n_rest_rows, n_rest_masks;
    uint32_t skip_live, pad_sl;            // skip_live: see TickArgs
    uint64_t rest_mask_off[MAX_MASKS];
    FoldArgs fold;
};
static_assert(sizeof(Tick2Args) <= 1024, "keep the kernel argument block small: it is re-sent every tick");

// Cross-workgroup hand-off of the partial rows: relaxed agent-scope 8-byte atomics on both sides (lowered to
// `global_store / global_load ... sc1`: write-through stores, L1-bypassing loads -- MI355X_MICROARCH.md, "Valid forms"),
// an explicit vmcnt(0) between a workgroup's row and its ticket, one agent-scope acquire in the workgroup that folds.
__device__ __forceinline__ void st8_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld8_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte store whose cache policy the instruction scheduler can see (a compiler-generated store, unlike st16<true>'s
// inline asm): used by the ILV variant, where sched_group_barrier spaces the stores out between the hash multiplies.
// The checksum fold shared by k_tick2 / k_tick3: one row of partials per workgroup (relaxed agent-scope stores), one
// agent-scope ticket, and the LAST workgroup to arrive folds every row (component_checksum.rs:92-95,
// entity_checksum.rs:29-52, checksum.rs:88-99) and writes each Save's Checksum(u128) to pinned host memory.
template <int NTHREADS>
__device__ __forceinline__ void tick_fold(const FoldArgs& f, uint32_t n_saves, uint64_t total_len, uint64_t* acc, uint32_t* s_last) {
    if (n_saves == 0) return;
    __syncthreads();                                              // the LDS atomics of every wave have landed
    const uint32_t nv = f.n_comp + 1u;                            // values per Save
    const uint32_t n_vals = n_saves * nv;
    for (uint32_t i = threadIdx.x; i < n_vals; i += NTHREADS) st8_agent(f.wg_parts + (uint64_t)blockIdx.x * n_vals + i, acc[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the row is in memory before the ticket is taken
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = (ticket == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!*s_last) return;                                         // workgroup-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (uint32_t i = threadIdx.x; i < n_vals; i += NTHREADS) acc[i] = 0;
    __syncthreads();
    {
        // rows are [gridDim.x][n_vals] u64 (compact): flat index i -> value i % n_vals.  24 loads in flight per lane and trip.
        const uint32_t n_flat = gridDim.x * n_vals;
        constexpr int INFL = 24;
        for (uint32_t i0 = threadIdx.x; i0 < n_flat; i0 += (uint32_t)INFL * NTHREADS) {
            uint64_t v[INFL];
#pragma unroll
            for (int u = 0; u < INFL; ++u) {
                const uint32_t i = i0 + (uint32_t)u * NTHREADS;
                v[u] = i < n_flat ? ld8_agent(f.wg_parts + i) : 0ULL;
            }
#pragma unroll
            for (int u = 0; u < INFL; ++u) {
                const uint32_t i = i0 + (uint32_t)u * NTHREADS;
                if (i >= n_flat) continue;
                const uint32_t c = i % n_vals;
                if ((c % nv) == f.n_comp) atomicAdd(reinterpret_cast<unsigned long long*>(&acc[c]), (unsigned long long)v[u]);
                else atomicXor(reinterpret_cast<unsigned long long*>(&acc[c]), (unsigned long long)v[u]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < n_saves) {
        const uint32_t k = threadIdx.x;
        uint64_t total = 0;
        for (uint32_t j = 0; j < f.n_comp; ++j)
            if ((f.comp_mask >> j) & 1u) total ^= sea_one(acc[k * nv + j]);     // component_checksum.rs:92-95
        total ^= sea_pair(acc[k * nv + f.n_comp], total_len);                  // entity_checksum.rs:29-52; XOR fold checksum.rs:88-99
        f.out[2 * (uint64_t)k] = total; f.out[2 * (uint64_t)k + 1] = 0;
    }
    if (threadIdx.x == 0) *f.ticket = 0;                          // ready for the next launch on this stream
}

__device__ __forceinline__ g_u8* sgpr_base_nv(const uint8_t* p) {      // sgpr_base without `volatile`: the statement may move
    uint64_t x = reinterpret_cast<uint64_t>(p);
    asm("" : "+s"(x));
    return (g_u8*)x;
}
template <bool NT>
__device__ __forceinline__ void st16v(g_u8* base, uint32_t lo, const u32x4& v) {
    if (NT) __builtin_nontemporal_store(v, (GGRS_GLOBAL u32x4*)(base + lo));
    else *(GGRS_GLOBAL u32x4*)(base + lo) = v;
}

// RESTL: EXACT number of untouched rows (the stress_test world: 7) -- a Save's body is then one straight-line block.
// ILV 0: a Save = one burst of row stores + the eight hash chains scheduled by the compiler, odd waves hashing first;
// ILV 1: the same instructions with one store placed after every ~1/15th of the hash VALU work (sched_group_barrier).
template <bool CKS_T, bool CKS_V, bool NT, int RESTL, int ILV>
__global__ __launch_bounds__(TPB) void k_tick2(Tick2Args a) {
    __shared__ uint64_t acc[MAX_TICK_SAVES * 3];      // this workgroup's partials: [save][T, V, count]
    __shared__ uint32_t s_last;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < MAX_TICK_SAVES * 3) acc[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t sh = (lane & 15u) * 4;
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t gw = blockIdx.x * 4u + wave_in_wg; gw < a.n_units; gw += n_waves) {       // gw: 256-slot unit
        const uint32_t t = gw >> 2, wave = gw & 3u;                   // its tile, and which quarter of it
        const uint32_t tid = wave * 64u + lane;                       // lane index inside the tile
        const bool in_len = (uint64_t)gw * 256u < a.len;              // wave-uniform
        const uint64_t e0 = (uint64_t)t * TILE + (uint64_t)tid * 4;   // first of this lane's 4 slots
        const uint64_t toff = wtile_off(t, a.ts, 4), toff8 = wtile_off(t, a.ts, 8);   // this tile's rows of a 4- / 8-byte column inside a block
        const uint32_t o4 = tid * 16u;                                // lane offset inside a 4-byte column's tile row
        // The 8-byte Ttl column: TWO fully contiguous 1 KiB accesses per wave (lane l moves bytes [16 l, 16 l + 16) of each
        // half of the wave's 2 KiB), not two 16-byte pieces at a 32-byte lane stride -- a streaming (nt) store of half-filled
        // lines leaves the L2 before its other half arrives.  So lane l owns the Ttl of unit slots {2l, 2l+1, 128+2l, 129+2l}
        // ("L slots") while it owns translation / velocity of unit slots {4l .. 4l+3}; liveness crosses over by ballot.
        const uint32_t o8a = wave * 2048u + lane * 16u, o8b = o8a + 1024u;
        const uint32_t w0 = t * 16u + wave * 4u;                      // first mask word of this wave
        const uint32_t wi8 = (w0 + (lane >> 4)) * 8u;                 // byte offset of this lane's mask word

        // ---- every load of the unit, back to back
        const uint64_t alive_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_alive + wi8);
        const uint64_t pT_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pT + wi8);
        const uint64_t pV_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pV + wi8);
        const uint64_t pL_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pL + wi8);
        float4 tx[3], vv[3];
        ulonglong2 tl[2];
        u32x4 restv[RESTL > 0 ? RESTL : 1];
        auto ld16 = [&](const uint8_t* p) -> u32x4 { return *reinterpret_cast<const u32x4*>(p); };
        if (in_len) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_t[k] + toff + o4); tx[k] = reinterpret_cast<const float4&>(x); }
#pragma unroll
            for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_v[k] + toff + o4); vv[k] = reinterpret_cast<const float4&>(x); }
            { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + o8a); tl[0] = reinterpret_cast<const ulonglong2&>(x); }
            { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + o8b); tl[1] = reinterpret_cast<const ulonglong2&>(x); }
#pragma unroll
            for (int j = 0; j < RESTL; ++j) restv[j] = ld16(a.src + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE + o4);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) { tx[k] = make_float4(0, 0, 0, 0); vv[k] = make_float4(0, 0, 0, 0); }
            tl[0] = make_ulonglong2(0, 0); tl[1] = make_ulonglong2(0, 0);
#pragma unroll
            for (int j = 0; j < RESTL; ++j) restv[j] = u32x4{0, 0, 0, 0};
        }
        // presence masks of components the schedule never touches: read once, fanned out to every snapshot (+ live on load)
        if (lane < 4u * a.n_rest_masks) {
            const uint32_t m = lane >> 2, mw = lane & 3u;
            const uint64_t o = a.rest_mask_off[m] + ((uint64_t)gw * 4 + mw) * 8;
            const uint64_t v = *reinterpret_cast<const uint64_t*>(a.src + o);
            for (uint32_t k = 0; k < a.n_saves; ++k)
                if (a.save_dst[k]) *reinterpret_cast<uint64_t*>(a.save_dst[k] + o) = v;
            if (!a.src_is_live && !a.skip_live) *reinterpret_cast<uint64_t*>(a.live + o) = v;
        }

        uint32_t alive4 = (uint32_t)(alive_w >> sh) & 0xFu;
        const uint32_t n_T = (uint32_t)(pT_w >> sh) & 0xFu, n_V = (uint32_t)(pV_w >> sh) & 0xFu;
        // liveness / Ttl presence of this lane's four L slots: words l >> 5 and 2 + (l >> 5) of the wave's four mask words
        // (lanes 0, 16, 32, 48 hold them), bit pair 2 (l & 31)
        auto word_of = [&](uint64_t v, int k) -> uint64_t {
            return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * k);
        };
        const uint32_t shL = 2u * (lane & 31u);
        auto l_bits = [&](uint64_t v) -> uint32_t {
            const uint64_t lo = lane < 32 ? word_of(v, 0) : word_of(v, 1), hi = lane < 32 ? word_of(v, 2) : word_of(v, 3);
            return ((uint32_t)(lo >> shL) & 3u) | (((uint32_t)(hi >> shL) & 3u) << 2);
        };
        uint32_t aliveL = l_bits(alive_w);
        const uint32_t presL = l_bits(pL_w);

        // diffuse(K0 ^ order), order == slot (RollbackOrdered::order): shared by both components and all Saves
        uint64_t ordB[4];
        if (CKS_T || CKS_V) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ordB[j] = sea_order_lane(e0 + j);
        }
        // one SeaHash chain: component_checksum.rs:81-90 for slot j of this lane (5 diffuses; the 8 chains of a Save are independent)
        auto chainT = [&](int j) -> uint64_t {
            const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&tx[0])[j]),
                                                                __float_as_uint(reinterpret_cast<float*>(&tx[1])[j]),
                                                                __float_as_uint(reinterpret_cast<float*>(&tx[2])[j])));
            return (((alive4 & n_T) >> j) & 1u) ? h : 0ULL;
        };
        auto chainV = [&](int j) -> uint64_t {
            const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&vv[0])[j]),
                                                                __float_as_uint(reinterpret_cast<float*>(&vv[1])[j]),
                                                                __float_as_uint(reinterpret_cast<float*>(&vv[2])[j])));
            return (((alive4 & n_V) >> j) & 1u) ? h : 0ULL;
        };
        const bool hash_first = (wave & 1u) != 0;                     // wave-uniform (see k_tick)

        // every load issued above has to land before the first op anyway; saying so explicitly keeps the op loop free of
        // conservative vmcnt waits (gfx9 counts stores in vmcnt too: such a wait would drain the snapshot stores in flight)
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0) expcnt(7) lgkmcnt(15)

        uint32_t si = 0, sj = 0;
        for (uint32_t i = 0; i < a.n_ops; ++i) {
            if (!((a.op_bits >> i) & 1ULL)) {
                // ---------------- SaveWorld: snapshot (component_snapshot.rs:66-84, entity.rs:39-51) + per-entity checksum half
                uint8_t* dst = a.save_dst[si];
                const uint64_t b0 = __ballot((alive4 >> 0) & 1u), b1 = __ballot((alive4 >> 1) & 1u),
                               b2 = __ballot((alive4 >> 2) & 1u), b3 = __ballot((alive4 >> 3) & 1u);
                const uint32_t cnt = (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
                uint64_t hT = 0, hV = 0;
                auto hash_all = [&]() {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (CKS_T) hT ^= chainT(j); if (CKS_V) hV ^= chainV(j); }
                };
                if (dst && in_len) {
                    if (ILV == 0) {
                        if (hash_first) hash_all();
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            st16<NT>(sgpr_base(dst + a.off_t[k] + toff), o4, tx[k]);
                            st16<NT>(sgpr_base(dst + a.off_v[k] + toff), o4, vv[k]);
                        }
                        st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8a, tl[0]);
                        st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8b, tl[1]);
#pragma unroll
                        for (int j = 0; j < RESTL; ++j) st16<NT>(sgpr_base(dst + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                        if (!hash_first) hash_all();
                    } else {
                        // four pieces: ~4 row stores, then the two chains of one slot (T_j, V_j: enough independent multiplies
                        // to keep the VALU issue-bound), pinned in this order
                        st16<NT>(sgpr_base(dst + a.off_t[0] + toff), o4, tx[0]); st16<NT>(sgpr_base(dst + a.off_t[1] + toff), o4, tx[1]);
                        st16<NT>(sgpr_base(dst + a.off_t[2] + toff), o4, tx[2]); st16<NT>(sgpr_base(dst + a.off_v[0] + toff), o4, vv[0]);
                        if (CKS_T) hT ^= chainT(0);
                        if (CKS_V) hV ^= chainV(0);
                        __builtin_amdgcn_sched_barrier(0);
                        st16<NT>(sgpr_base(dst + a.off_v[1] + toff), o4, vv[1]); st16<NT>(sgpr_base(dst + a.off_v[2] + toff), o4, vv[2]);
                        st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8a, tl[0]); st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8b, tl[1]);
                        if (CKS_T) hT ^= chainT(1);
                        if (CKS_V) hV ^= chainV(1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < (RESTL + 1) / 2; ++j) st16<NT>(sgpr_base(dst + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                        if (CKS_T) hT ^= chainT(2);
                        if (CKS_V) hV ^= chainV(2);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = (RESTL + 1) / 2; j < RESTL; ++j) st16<NT>(sgpr_base(dst + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                        if (CKS_T) hT ^= chainT(3);
                        if (CKS_V) hV ^= chainV(3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    hash_all();
                }
                if (dst) {
                    uint64_t mine = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                            (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
                        if (lane == (uint32_t)w) mine = nw;
                    }
                    if (lane < 4) st8(sgpr_base(dst + a.off_alive) + (w0 + lane) * 8u, mine);
                    if ((lane & 15u) == 0) {
                        st8(sgpr_base(dst + a.off_pT) + wi8, pT_w);
                        st8(sgpr_base(dst + a.off_pV) + wi8, pV_w);
                        st8(sgpr_base(dst + a.off_pL) + wi8, pL_w);
                    }
                    if (gw == 0 && lane == 0) {
                        Header h; h.len = a.len; h.frame = a.save_frame[si]; h.pad0 = 0; h.active = 0;
                        h.checksum[0] = 0; h.checksum[1] = 0;
                        *reinterpret_cast<Header*>(dst) = h;
                    }
                }
                // this wave's partials of the Save -> the workgroup's LDS accumulators (fire-and-forget LDS atomics)
                if (CKS_T) hT = wave_xor(hT);
                if (CKS_V) hV = wave_xor(hV);
                if (lane == 0) {
                    if (CKS_T) atomicXor(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 0]), (unsigned long long)hT);
                    if (CKS_V) atomicXor(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 1]), (unsigned long long)hV);
                    atomicAdd(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 2]), (unsigned long long)cnt);
                }
                ++si;
            } else {
                // ---------------- AdvanceWorld: update_particles + despawn_particles (particles.rs:272-289)
                const float dt = __uint_as_float(a.dt_bits[sj]);
                ++sj;
                const uint32_t m_upd = alive4 & n_T & n_V;     // Query<(&mut Transform, &mut Velocity)>
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gd = __fmul_rn(a.g[k], dt);    // gravity * time_step
                    float* x = reinterpret_cast<float*>(&tx[k]);
                    float* v = reinterpret_cast<float*>(&vv[k]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = (m_upd >> j) & 1u;
                        const float nv = __fadd_rn(v[j], gd);                     // **velocity += ...
                        const float nx = __fadd_rn(x[j], __fmul_rn(nv, dt));      // translation += **velocity * time_step
                        v[j] = on ? nv : v[j];
                        x[j] = on ? nx : x[j];
                    }
                }
                const uint32_t m_ttl = aliveL & presL;         // Query<(Entity, &mut Ttl)>, over this lane's L slots
                uint32_t killL = 0;
                uint64_t* q = reinterpret_cast<uint64_t*>(&tl[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = (m_ttl >> j) & 1u;
                    const uint64_t nq = q[j] - 1;              // usize, wrapping
                    q[j] = on ? nq : q[j];
                    killL |= (on && nq == 0) ? (1u << j) : 0u;
                }
                aliveL &= ~killL;                              // despawn is deferred to the end of the frame
                // the kills, seen from the lanes that own the same slots' translation / velocity: unit slot 4m + i lives in
                // L lane (2m + (i >> 1)) & 63, L index (i & 1) for m < 32 and 2 + (i & 1) for m >= 32
                const uint64_t k0 = __ballot((killL >> 0) & 1u), k1 = __ballot((killL >> 1) & 1u),
                               k2 = __ballot((killL >> 2) & 1u), k3 = __ballot((killL >> 3) & 1u);
                if ((k0 | k1 | k2 | k3) != 0) {                // wave-uniform
                    const uint64_t ka = lane < 32 ? k0 : k2, kb = lane < 32 ? k1 : k3;
                    const uint32_t pa = (uint32_t)(ka >> shL) & 3u, pb = (uint32_t)(kb >> shL) & 3u;
                    const uint32_t kill4 = (pa & 1u) | ((pb & 1u) << 1) | ((pa >> 1) << 2) | ((pb >> 1) << 3);
                    alive4 &= ~kill4;
                }
            }
        }

        // ---- the live block, written once
        if ((!a.src_is_live || a.n_steps) && !a.skip_live) {
            if (in_len) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    st16<false>(sgpr_base(a.live + a.off_t[k] + toff), o4, tx[k]);
                    st16<false>(sgpr_base(a.live + a.off_v[k] + toff), o4, vv[k]);
                }
                st16<false>(sgpr_base(a.live + a.off_ttl + toff8), o8a, tl[0]);
                st16<false>(sgpr_base(a.live + a.off_ttl + toff8), o8b, tl[1]);
                if (!a.src_is_live) {
#pragma unroll
                    for (int j = 0; j < RESTL; ++j) st16<false>(sgpr_base(a.live + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                }
            }
            const uint64_t b0 = __ballot((alive4 >> 0) & 1u), b1 = __ballot((alive4 >> 1) & 1u),
                           b2 = __ballot((alive4 >> 2) & 1u), b3 = __ballot((alive4 >> 3) & 1u);
            uint64_t mine = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                    (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
                if (lane == (uint32_t)w) mine = nw;
            }
            if (lane < 4) st8(sgpr_base(a.live + a.off_alive) + (w0 + lane) * 8u, mine);
            if (!a.src_is_live && (lane & 15u) == 0) {
                st8(sgpr_base(a.live + a.off_pT) + wi8, pT_w);
                st8(sgpr_base(a.live + a.off_pV) + wi8, pV_w);
                st8(sgpr_base(a.live + a.off_pL) + wi8, pL_w);
            }
        }
    }

    tick_fold<TPB>(a.fold, a.n_saves, a.len, acc, &s_last);
}

// ------------------------------------------------------------------ k_tick3 (wave-specialised fused request group)
// k_tick2's remaining cost over the memory system's floor for this traffic (scripts/ubench3.hip: ~95-99 us of stores
// with no ALU at all vs 112-118 us) is ISSUE COUPLING: a wave that is blocked issuing a store (the TA queue is full
// 55 % of the time, SQ_WAIT_INST_ANY) cannot hash, and a wave that hashes does not feed the store queue.  So the two
// jobs get their own waves.  A 512-thread workgroup owns one 1024-slot tile:
//   * waves 0-3, COMPUTE: keep the 8 schedule-owned rows of their 256-slot quarter in registers, replay the ops, hash;
//     at every Save they drop the 8 rows (+ the 4 rebuilt liveness words) into LDS and go on hashing / stepping;
//   * waves 4-7, STORE: keep the 7 untouched rows of the same quarter in registers; per Save they pick the 8 rows up
//     from LDS and stream all 15 rows (nt) + the mask words to the ring slot, blocking on the store queue as long as
//     it takes -- nobody is waiting for them but the next barrier.
// One LDS-only barrier per Save, LDS double-buffered (2 x 32 KiB): compute runs at most one Save ahead.  Two
// workgroups per CU (64.5 KiB LDS, <= 128 VGPRs): while one workgroup's store waves wait at a barrier the other's keep
// the queue fed.  ubench3 `fan_spec` (same structure, hash stand-in): 98.7-102.5 us vs 108.9-110.4 us for the uniform
// kernel at the engine's occupancy.  Checksum fold: tick_fold (as k_tick2).
// PAIRSYNC 0: one workgroup-wide LDS barrier per hand-off.  PAIRSYNC 1: every compute / store pair has its own two-slot
// ring guarded by LDS flags -- no coupling between the four pairs of a workgroup, and a store wave frees its slot as
// soon as the rows are in its registers, i.e. BEFORE it starts issuing the (blocking) global stores.
// RESTL / EXACT: the store waves keep up to RESTL untouched 4-byte rows in registers; EXACT: the world has exactly RESTL of
// them (the stress_test: 7) and the row loops are straight-line code, else every row is guarded by `j < n_rest_rows`.
template <bool CKS_T, bool CKS_V, bool NT, int RESTL, int PAIRSYNC, bool EXACT = true>
__global__ __launch_bounds__(512, 4) void k_tick3(Tick2Args a) {
    __shared__ __attribute__((aligned(16))) u32x4 rowbuf[2][4][8][64];   // [parity][quarter][row][lane]: 64 KiB
    __shared__ uint64_t maskbuf[2][4][4];                                  // the quarter's 4 liveness words per Save
    __shared__ uint32_t full[4][2];                                        // PAIRSYNC: slot state of pair q (0 free, 1 filled)
    __shared__ uint64_t acc[MAX_TICK_SAVES * 3];
    __shared__ uint32_t s_last;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave8 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool store_role = wave8 >= 4u;                              // wave-uniform
    const uint32_t wave = wave8 & 3u;                                 // the quarter both waves of a pair serve
    if (threadIdx.x < MAX_TICK_SAVES * 3) acc[threadIdx.x] = 0;
    if (threadIdx.x < 8) full[threadIdx.x >> 1][threadIdx.x & 1u] = 0;
    __syncthreads();
    // LDS-only flag traffic of a pair (LDS operations of one wave execute in issue order; the waits are lgkmcnt-only so
    // that a store wave's global stores stay in flight)
    auto flag_wait = [&](uint32_t q, uint32_t p, uint32_t want) {
        for (;;) {
            const uint32_t v = __hip_atomic_load(&full[q][p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == want) break;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto flag_set = [&](uint32_t q, uint32_t p, uint32_t v) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                           // lgkmcnt(0): this wave's LDS reads / writes of the slot are done
        __hip_atomic_store(&full[q][p], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    const uint32_t sh = (lane & 15u) * 4;
    const uint32_t n_tiles = (a.n_units + 3u) >> 2;
    uint32_t par = 0;                                                 // LDS buffer parity, advances with every hand-off
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t gw = t * 4u + wave;                            // 256-slot unit
        const uint32_t tid = wave * 64u + lane;                       // lane index inside the tile
        const bool in_len = (uint64_t)gw * 256u < a.len;              // wave-uniform
        const uint64_t e0 = (uint64_t)t * TILE + (uint64_t)tid * 4;
        const uint64_t toff = wtile_off(t, a.ts, 4), toff8 = wtile_off(t, a.ts, 8);
        const uint32_t o4 = tid * 16u;
        const uint32_t o8a = wave * 2048u + lane * 16u, o8b = o8a + 1024u;   // contiguous Ttl halves (see k_tick2)
        const uint32_t w0 = t * 16u + wave * 4u;
        const uint32_t wi8 = (w0 + (lane >> 4)) * 8u;

        // both roles need the presence words (the store waves write them into every snapshot)
        const uint64_t pT_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pT + wi8);
        const uint64_t pV_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pV + wi8);
        const uint64_t pL_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pL + wi8);

        if (store_role) {
            // ================================================= STORE waves
            u32x4 restv[RESTL > 0 ? RESTL : 1];
#pragma unroll
            for (int j = 0; j < RESTL; ++j) {
                restv[j] = u32x4{0, 0, 0, 0};
                if (in_len && (EXACT || (uint32_t)j < a.n_rest_rows)) restv[j] = *reinterpret_cast<const u32x4*>(a.src + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE + o4);
            }
            if (lane < 4u * a.n_rest_masks) {                         // untouched presence masks: fan out (+ live on load)
                const uint32_t m = lane >> 2, mw = lane & 3u;
                const uint64_t o = a.rest_mask_off[m] + ((uint64_t)gw * 4 + mw) * 8;
                const uint64_t v = *reinterpret_cast<const uint64_t*>(a.src + o);
                for (uint32_t k = 0; k < a.n_saves; ++k)
                    if (a.save_dst[k]) *reinterpret_cast<uint64_t*>(a.save_dst[k] + o) = v;
                if (!a.src_is_live && !a.skip_live) *reinterpret_cast<uint64_t*>(a.live + o) = v;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);                       // the loads have landed: the op loop stays free of vmcnt waits
            auto put = [&](uint8_t* dst, bool snapshot, bool with_rest, bool with_presence, int32_t frame) {
                // rows of this quarter: 8 from LDS, RESTL from registers
                if (PAIRSYNC) flag_wait(wave, par, 1u);
                u32x4 h[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) h[r] = rowbuf[par][wave][r][lane];
                const uint64_t mw = lane < 4 ? maskbuf[par][wave][lane] : 0ULL;
                if (PAIRSYNC) flag_set(wave, par, 0u);                 // rows are in registers: the compute wave may refill the slot
                if (in_len) {
                    if (snapshot) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { st16<NT>(sgpr_base(dst + a.off_t[k] + toff), o4, h[k]); st16<NT>(sgpr_base(dst + a.off_v[k] + toff), o4, h[3 + k]); }
                        st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8a, h[6]);
                        st16<NT>(sgpr_base(dst + a.off_ttl + toff8), o8b, h[7]);
#pragma unroll
                        for (int j = 0; j < RESTL; ++j) if (EXACT || (uint32_t)j < a.n_rest_rows) st16<NT>(sgpr_base(dst + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 3; ++k) { st16<false>(sgpr_base(dst + a.off_t[k] + toff), o4, h[k]); st16<false>(sgpr_base(dst + a.off_v[k] + toff), o4, h[3 + k]); }
                        st16<false>(sgpr_base(dst + a.off_ttl + toff8), o8a, h[6]);
                        st16<false>(sgpr_base(dst + a.off_ttl + toff8), o8b, h[7]);
                        if (with_rest) {
#pragma unroll
                            for (int j = 0; j < RESTL; ++j) if (EXACT || (uint32_t)j < a.n_rest_rows) st16<false>(sgpr_base(dst + a.rest_off + toff + (uint32_t)j * REST_ROW_STRIDE), o4, restv[j]);
                        }
                    }
                }
                if (lane < 4) st8(sgpr_base(dst + a.off_alive) + (w0 + lane) * 8u, mw);
                if (with_presence && (lane & 15u) == 0) {
                    st8(sgpr_base(dst + a.off_pT) + wi8, pT_w);
                    st8(sgpr_base(dst + a.off_pV) + wi8, pV_w);
                    st8(sgpr_base(dst + a.off_pL) + wi8, pL_w);
                }
                if (snapshot && gw == 0 && lane == 0) {
                    Header hd; hd.len = a.len; hd.frame = frame; hd.pad0 = 0; hd.active = 0; hd.checksum[0] = 0; hd.checksum[1] = 0;
                    *reinterpret_cast<Header*>(dst) = hd;
                }
            };
            uint32_t si = 0;
            for (uint32_t i = 0; i < a.n_ops; ++i) {
                if ((a.op_bits >> i) & 1ULL) continue;                // Advance: nothing to store
                uint8_t* dst = a.save_dst[si];
                if (dst) { if (!PAIRSYNC) lds_barrier(); put(dst, true, true, true, a.save_frame[si]); par ^= 1u; }
                ++si;
            }
            if ((!a.src_is_live || a.n_steps) && !a.skip_live) { if (!PAIRSYNC) lds_barrier(); put(a.live, false, !a.src_is_live, !a.src_is_live, 0); par ^= 1u; }
        } else {
            // ================================================= COMPUTE waves
            const uint64_t alive_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_alive + wi8);
            float4 tx[3], vv[3];
            ulonglong2 tl[2];
            auto ld16 = [&](const uint8_t* p) -> u32x4 { return *reinterpret_cast<const u32x4*>(p); };
            if (in_len) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_t[k] + toff + o4); tx[k] = reinterpret_cast<const float4&>(x); }
#pragma unroll
                for (int k = 0; k < 3; ++k) { const u32x4 x = ld16(a.src + a.off_v[k] + toff + o4); vv[k] = reinterpret_cast<const float4&>(x); }
                { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + o8a); tl[0] = reinterpret_cast<const ulonglong2&>(x); }
                { const u32x4 x = ld16(a.src + a.off_ttl + toff8 + o8b); tl[1] = reinterpret_cast<const ulonglong2&>(x); }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { tx[k] = make_float4(0, 0, 0, 0); vv[k] = make_float4(0, 0, 0, 0); }
                tl[0] = make_ulonglong2(0, 0); tl[1] = make_ulonglong2(0, 0);
            }
            uint32_t alive4 = (uint32_t)(alive_w >> sh) & 0xFu;
            const uint32_t n_T = (uint32_t)(pT_w >> sh) & 0xFu, n_V = (uint32_t)(pV_w >> sh) & 0xFu;
            auto word_of = [&](uint64_t v, int k) -> uint64_t {
                return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 16 * k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 16 * k);
            };
            const uint32_t shL = 2u * (lane & 31u);
            auto l_bits = [&](uint64_t v) -> uint32_t {
                const uint64_t lo = lane < 32 ? word_of(v, 0) : word_of(v, 1), hi = lane < 32 ? word_of(v, 2) : word_of(v, 3);
                return ((uint32_t)(lo >> shL) & 3u) | (((uint32_t)(hi >> shL) & 3u) << 2);
            };
            uint32_t aliveL = l_bits(alive_w);
            const uint32_t presL = l_bits(pL_w);
            uint64_t ordB[4];
            if (CKS_T || CKS_V) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ordB[j] = sea_order_lane(e0 + j);
            }
            auto chainT = [&](int j) -> uint64_t {
                const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&tx[0])[j]),
                                                                    __float_as_uint(reinterpret_cast<float*>(&tx[1])[j]),
                                                                    __float_as_uint(reinterpret_cast<float*>(&tx[2])[j])));
                return (((alive4 & n_T) >> j) & 1u) ? h : 0ULL;
            };
            auto chainV = [&](int j) -> uint64_t {
                const uint64_t h = sea_pair_pre(ordB[j], sea_inner3(__float_as_uint(reinterpret_cast<float*>(&vv[0])[j]),
                                                                    __float_as_uint(reinterpret_cast<float*>(&vv[1])[j]),
                                                                    __float_as_uint(reinterpret_cast<float*>(&vv[2])[j])));
                return (((alive4 & n_V) >> j) & 1u) ? h : 0ULL;
            };
            // the quarter's rows + rebuilt liveness words -> LDS[par]; the barrier hands them to the paired store wave
            auto hand_off = [&]() {
                if (PAIRSYNC) flag_wait(wave, par, 0u);
#pragma unroll
                for (int k = 0; k < 3; ++k) { rowbuf[par][wave][k][lane] = reinterpret_cast<const u32x4&>(tx[k]); rowbuf[par][wave][3 + k][lane] = reinterpret_cast<const u32x4&>(vv[k]); }
                rowbuf[par][wave][6][lane] = reinterpret_cast<const u32x4&>(tl[0]);
                rowbuf[par][wave][7][lane] = reinterpret_cast<const u32x4&>(tl[1]);
                const uint64_t b0 = __ballot((alive4 >> 0) & 1u), b1 = __ballot((alive4 >> 1) & 1u),
                               b2 = __ballot((alive4 >> 2) & 1u), b3 = __ballot((alive4 >> 3) & 1u);
                uint64_t mine = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint64_t nw = spread4(b0 >> (16 * w)) | (spread4(b1 >> (16 * w)) << 1) |
                                        (spread4(b2 >> (16 * w)) << 2) | (spread4(b3 >> (16 * w)) << 3);
                    if (lane == (uint32_t)w) mine = nw;
                }
                if (lane < 4) maskbuf[par][wave][lane] = mine;
                if (PAIRSYNC) flag_set(wave, par, 1u); else lds_barrier();
                par ^= 1u;
                return (uint32_t)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);

            uint32_t si = 0, sj = 0;
            for (uint32_t i = 0; i < a.n_ops; ++i) {
                if (!((a.op_bits >> i) & 1ULL)) {
                    // ---------------- SaveWorld: hand the rows to the store wave, then hash them
                    uint32_t cnt;
                    if (a.save_dst[si]) cnt = hand_off();
                    else cnt = (uint32_t)(__popcll(__ballot((alive4 >> 0) & 1u)) + __popcll(__ballot((alive4 >> 1) & 1u)) +
                                          __popcll(__ballot((alive4 >> 2) & 1u)) + __popcll(__ballot((alive4 >> 3) & 1u)));
                    uint64_t hT = 0, hV = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { if (CKS_T) hT ^= chainT(j); if (CKS_V) hV ^= chainV(j); }
                    if (CKS_T) hT = wave_xor(hT);
                    if (CKS_V) hV = wave_xor(hV);
                    if (lane == 0) {
                        if (CKS_T) atomicXor(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 0]), (unsigned long long)hT);
                        if (CKS_V) atomicXor(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 1]), (unsigned long long)hV);
                        atomicAdd(reinterpret_cast<unsigned long long*>(&acc[si * 3 + 2]), (unsigned long long)cnt);
                    }
                    ++si;
                } else {
                    // ---------------- AdvanceWorld: update_particles + despawn_particles (particles.rs:272-289)
                    const float dt = __uint_as_float(a.dt_bits[sj]);
                    ++sj;
                    const uint32_t m_upd = alive4 & n_T & n_V;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float gd = __fmul_rn(a.g[k], dt);
                        float* x = reinterpret_cast<float*>(&tx[k]);
                        float* v = reinterpret_cast<float*>(&vv[k]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool on = (m_upd >> j) & 1u;
                            const float nv = __fadd_rn(v[j], gd);
                            const float nx = __fadd_rn(x[j], __fmul_rn(nv, dt));
                            v[j] = on ? nv : v[j];
                            x[j] = on ? nx : x[j];
                        }
                    }
                    const uint32_t m_ttl = aliveL & presL;
                    uint32_t killL = 0;
                    uint64_t* q = reinterpret_cast<uint64_t*>(&tl[0]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = (m_ttl >> j) & 1u;
                        const uint64_t nq = q[j] - 1;
                        q[j] = on ? nq : q[j];
                        killL |= (on && nq == 0) ? (1u << j) : 0u;
                    }
                    aliveL &= ~killL;
                    const uint64_t k0 = __ballot((killL >> 0) & 1u), k1 = __ballot((killL >> 1) & 1u),
                                   k2 = __ballot((killL >> 2) & 1u), k3 = __ballot((killL >> 3) & 1u);
                    if ((k0 | k1 | k2 | k3) != 0) {
                        const uint64_t ka = lane < 32 ? k0 : k2, kb = lane < 32 ? k1 : k3;
                        const uint32_t pa = (uint32_t)(ka >> shL) & 3u, pb = (uint32_t)(kb >> shL) & 3u;
                        alive4 &= ~((pa & 1u) | ((pb & 1u) << 1) | ((pa >> 1) << 2) | ((pb >> 1) << 3));
                    }
                }
            }
            if ((!a.src_is_live || a.n_steps) && !a.skip_live) (void)hand_off();        // the live block, written once (by the store wave)
        }
    }
    tick_fold<512>(a.fold, a.n_saves, a.len, acc, &s_last);
}

// ------------------------------------------------------------------ k_tick1 (one slot per lane)
// The same fused request group with a 256-slot tile: one slot per lane, one wave == one 64-bit mask
// word (the wave's __ballot IS the liveness word).  Four times as many, four times lighter
// workgroups: small worlds (10k..100k entities: BASELINE configs 2, 4, 5) fill the 256 CUs instead
// of leaving most of them idle behind a few long-running 1024-slot tiles, ~60 VGPRs give 8 waves/SIMD,
// and with more workgroups than residency slots the read phase of late tiles overlaps the store phase
// of early ones.  Accesses are 4 B (8 B for u64 columns) per lane: 256 B per wave instruction.
//
// DP ("depth-parallel", grid.z = roles): a tick of a small world is ONE dependent chain per lane -- load, 9 steps, 8 hashes,
// 8 store bursts, ~2500 instructions at one wave per SIMD, ~12 us however few entities there are.  The steps are a handful
// of flops; the hashes and stores are the chain.  The group's outputs are its Saves plus the live world; workgroup (t, z)
// REPLAYS the steps and produces only outputs [z*dp_s, (z+1)*dp_s): dp_s = 1 gives nine times the workgroups, each a sixth of
// the chain.  Same operations in the same order per slot, so the same bits.  Every role reads the source block and writes
// different blocks: valid only when the source is none of the destinations (the host checks; a group that starts with
// LoadWorld -- every SyncTest tick, every rollback -- qualifies).
constexpr int TILE1 = 256;
template <bool CKS_T, bool CKS_V, bool NT, bool DP = false>
__global__ __launch_bounds__(TPB) void k_tick1(TickArgs a) {
    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
    const uint32_t o_first = DP ? blockIdx.z * a.dp_s : 0u;         // DP: this workgroup's outputs; output n_saves is the live world
    const uint32_t o_last = DP ? min(o_first + a.dp_s, a.n_saves + 1u) : a.n_saves + 1u;
    if (DP && o_first == a.n_saves && !writes_live) return;         // a role with nothing to write
    // Small worlds are latency-bound, and every dynamically indexed kernel argument (save_dst[si], dt_bits[sj], rest[r]
    // ...) is a dependent scalar load that misses the scalar cache on its first touch.  One vector load per lane
    // stages the whole argument block in LDS; the op loop then reads it with ds_read (an order of magnitude closer).
    __shared__ __attribute__((aligned(16))) TickArgs sa;
    static_assert(sizeof(TickArgs) % 16 == 0 && sizeof(TickArgs) <= TPB * 16, "one 16-byte piece per lane");
    if (tid < sizeof(TickArgs) / 16) reinterpret_cast<u32x4*>(&sa)[tid] = reinterpret_cast<const u32x4*>(&a)[tid];
    __syncthreads();
    auto uni64 = [](uint64_t v) -> uint64_t {                      // a uniform value read through LDS, back into SGPRs
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const bool in_len = (uint64_t)t * TILE1 < a.len;              // workgroup-uniform
    const uint32_t e = t * TILE1 + tid;                           // this lane's slot
    const uint64_t toff = wtile_off(t >> 2, a.ts, 4), toff8 = wtile_off(t >> 2, a.ts, 8);   // its 1024-slot tile's rows of a 4- / 8-byte column
    const uint32_t ti = (t & 3u) * TILE1 + tid;                   // its index inside that tile
    const uint32_t o4 = ti * 4u, o8 = ti * 8u;
    const uint32_t wi8 = (t * 4u + wave) * 8u;                    // this wave's mask word

    const uint64_t alive_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_alive + wi8);
    const uint64_t pT_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pT + wi8);
    const uint64_t pV_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pV + wi8);
    const uint64_t pL_w = *reinterpret_cast<const uint64_t*>(a.src + a.off_pL + wi8);
    float tx[3] = {0, 0, 0}, vv[3] = {0, 0, 0};
    uint64_t ttl = 0;
    if (in_len) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tx[k] = *reinterpret_cast<const float*>(a.src + a.off_t[k] + toff + o4);
#pragma unroll
        for (int k = 0; k < 3; ++k) vv[k] = *reinterpret_cast<const float*>(a.src + a.off_v[k] + toff + o4);
        ttl = *reinterpret_cast<const uint64_t*>(a.src + a.off_ttl + toff8 + o8);
    }

    // ---- state the schedule never touches: read once, fan out to every snapshot (+ live on load)
    if (tid < 4u * a.n_rest_masks) {
        const uint32_t m = tid >> 2, mw = tid & 3u;
        const uint64_t o = sa.rest_mask_off[m] + ((uint64_t)t * 4 + mw) * 8;
        const uint64_t v = *reinterpret_cast<const uint64_t*>(a.src + o);
        for (uint32_t k = 0; k < a.n_saves; ++k)
            if (k >= o_first && k < o_last && a.save_dst[k]) *reinterpret_cast<uint64_t*>(a.save_dst[k] + o) = v;
        if (o_last == a.n_saves + 1u && !a.src_is_live && !a.skip_live) *reinterpret_cast<uint64_t*>(a.live + o) = v;
    }
    if (in_len && (a.n_saves || !a.src_is_live)) {
        // rest[] lists 4 KiB rows of 1024-slot tiles; a column is its row with roff == 0.
        // Up to 8 columns per batch, one wait per batch.
        const uint32_t n_rows = a.n_rest_rows;
        uint32_t r = 0;
        while (r < n_rows) {
            uint64_t v[8]; uint64_t off[8]; uint32_t wb[8];
            int nb = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = 0; off[j] = 0; wb[j] = 0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                while (r < n_rows && sa.rest[r].roff != 0) ++r;
                if (r < n_rows) {
                    const RowLite rd = sa.rest[r]; ++r;
                    wb[j] = rd.word_bytes; off[j] = rd.col_off; nb = j + 1;
                    if (wb[j] == 8) v[j] = *reinterpret_cast<const uint64_t*>(a.src + off[j] + toff8 + o8);
                    else v[j] = *reinterpret_cast<const uint32_t*>(a.src + off[j] + toff + o4);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): land the loads once
            for (uint32_t k = o_first; k < o_last; ++k) {
                uint8_t* dst = k < a.n_saves ? a.save_dst[k] : ((a.src_is_live || a.skip_live) ? nullptr : a.live);
                if (!dst) continue;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j < nb) {
                        if (wb[j] == 8) *reinterpret_cast<uint64_t*>(dst + off[j] + toff8 + o8) = v[j];
                        else *reinterpret_cast<uint32_t*>(dst + off[j] + toff + o4) = (uint32_t)v[j];
                    }
                }
            }
        }
    }

    bool alive = (alive_w >> lane) & 1ULL;
    const bool has_T = (pT_w >> lane) & 1ULL, has_V = (pV_w >> lane) & 1ULL, has_L = (pL_w >> lane) & 1ULL;
    uint64_t ordB = 0;
    if (CKS_T || CKS_V) ordB = sea_order_lane(e);

    uint32_t si = 0, sj = 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // see k_tick: keep the op loop wait-free
    for (uint32_t i = 0; i < a.n_ops; ++i) {
        if (!((a.op_bits >> i) & 1ULL)) {
            // ---------------- SaveWorld
            if (DP && si < o_first) { ++si; continue; }           // another workgroup's snapshot
            uint8_t* dst = reinterpret_cast<uint8_t*>(uni64(reinterpret_cast<uint64_t>(sa.save_dst[si])));
            const uint64_t alive_now = __ballot(alive);           // == this wave's liveness word
            if (dst) {
                if (in_len) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        *reinterpret_cast<float*>(dst + a.off_t[k] + toff + o4) = tx[k];
                        *reinterpret_cast<float*>(dst + a.off_v[k] + toff + o4) = vv[k];
                    }
                    *reinterpret_cast<uint64_t*>(dst + a.off_ttl + toff8 + o8) = ttl;
                }
                if (lane == 0) {
                    *reinterpret_cast<uint64_t*>(dst + a.off_alive + wi8) = alive_now;
                    *reinterpret_cast<uint64_t*>(dst + a.off_pT + wi8) = pT_w;
                    *reinterpret_cast<uint64_t*>(dst + a.off_pV + wi8) = pV_w;
                    *reinterpret_cast<uint64_t*>(dst + a.off_pL + wi8) = pL_w;
                }
                if (t == 0 && tid == 0) {
                    Header h; h.len = a.len; h.frame = sa.save_frame[si]; h.pad0 = 0; h.active = 0;
                    h.checksum[0] = 0; h.checksum[1] = 0;
                    *reinterpret_cast<Header*>(dst) = h;
                }
            }
            uint64_t hT = 0, hV = 0;
            if (CKS_T) {
                const uint64_t h = sea_pair_pre(ordB, sea_inner3(__float_as_uint(tx[0]), __float_as_uint(tx[1]), __float_as_uint(tx[2])));
                hT = wave_xor((alive && has_T) ? h : 0ULL);
            }
            if (CKS_V) {
                const uint64_t h = sea_pair_pre(ordB, sea_inner3(__float_as_uint(vv[0]), __float_as_uint(vv[1]), __float_as_uint(vv[2])));
                hV = wave_xor((alive && has_V) ? h : 0ULL);
            }
            if (lane == 0) {
                // per-wave partials, folded by k_tick_finalize.  In-kernel folds LOSE at these sizes: one last workgroup
                // (tick_fold) has no long store drain to hide its ticket + acquire + gather behind (10 k: 25.5 vs 23.5 us,
                // 300 k: 59 vs 47 us per tick), and a ticket per depth-parallel role serialises hundreds of short workgroups on
                // one agent-scope atomic (profiles/r02dp/ab3.txt: 30 k 23.1 vs 17.4 us, 100 k 32.5 vs 25.3 us).
                // blockIdx.y: member of a batch of identical checksum-only groups (speculative branches off one snapshot)
                uint64_t* p = a.parts + ((uint64_t)blockIdx.y * a.n_saves + si) * 3 * a.part_stride + (uint64_t)t * 4 + wave;
                p[0] = hT; p[a.part_stride] = hV; p[2 * (uint64_t)a.part_stride] = (uint64_t)__popcll(alive_now);
            }
            ++si;
            if (DP && si == o_last) return;                       // this workgroup's snapshots are out (the live world is another role's)
        } else {
            // ---------------- AdvanceWorld: update_particles + despawn_particles (particles.rs:272-289)
            const float dt = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)sa.dt_bits[sj]));
            ++sj;
            const bool upd = alive && has_T && has_V, tt = alive && has_L;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float gd = __fmul_rn(a.g[k], dt);
                const float nv = __fadd_rn(vv[k], gd);
                const float nx = __fadd_rn(tx[k], __fmul_rn(nv, dt));
                vv[k] = upd ? nv : vv[k];
                tx[k] = upd ? nx : tx[k];
            }
            const uint64_t nq = ttl - 1;                          // usize, wrapping
            ttl = tt ? nq : ttl;
            alive = alive && !(tt && nq == 0);                    // despawn is deferred to the end of the frame
        }
    }

    if ((!a.src_is_live || a.n_steps) && !a.skip_live) {
        if (in_len) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                *reinterpret_cast<float*>(a.live + a.off_t[k] + toff + o4) = tx[k];
                *reinterpret_cast<float*>(a.live + a.off_v[k] + toff + o4) = vv[k];
            }
            *reinterpret_cast<uint64_t*>(a.live + a.off_ttl + toff8 + o8) = ttl;
        }
        const uint64_t alive_now = __ballot(alive);
        if (lane == 0) {
            *reinterpret_cast<uint64_t*>(a.live + a.off_alive + wi8) = alive_now;
            if (!a.src_is_live) {
                *reinterpret_cast<uint64_t*>(a.live + a.off_pT + wi8) = pT_w;
                *reinterpret_cast<uint64_t*>(a.live + a.off_pV + wi8) = pV_w;
                *reinterpret_cast<uint64_t*>(a.live + a.off_pL + wi8) = pL_w;
            }
        }
    }
}

// Folds the per-wave partials of every Save of a fused group: one 1024-thread workgroup per Save.
// component_checksum.rs:92-95 (hash the XOR once more), entity_checksum.rs:29-52,
// checksum.rs:88-99 (XOR of all parts; upper 64 bits of the u128 are always 0).
struct TickFinArgs {
    const uint64_t* parts; uint32_t part_stride, n_parts, cks_T, cks_V;
    uint64_t total_len;
    uint64_t* out;                         // {lo, hi} per Save (pinned, device-mapped host memory)
};
constexpr int FIN_TPB = 1024;
__global__ __launch_bounds__(FIN_TPB) void k_tick_finalize(TickFinArgs f) {
    const uint32_t k = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t* pT = f.parts + (uint64_t)k * 3 * f.part_stride;
    const uint64_t* pV = pT + f.part_stride;
    const uint64_t* pC = pV + f.part_stride;
    uint64_t xT = 0, xV = 0, sum = 0;
    // 4 independent strided loads per array per trip: the partials sit in other XCDs' L2 / HBM, so the
    // fold is latency-bound unless every load of a thread is in flight at once
    for (uint32_t i0 = tid; i0 < f.n_parts; i0 += 4 * FIN_TPB) {
        uint64_t t[4], v[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * FIN_TPB;
            const bool in = i < f.n_parts;
            const uint32_t ii = in ? i : i0;
            t[u] = pT[ii]; v[u] = pV[ii]; c[u] = pC[ii];
            if (!in) { t[u] = 0; v[u] = 0; c[u] = 0; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { xT ^= t[u]; xV ^= v[u]; sum += c[u]; }
    }
    xT = wave_xor(xT); xV = wave_xor(xV);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __shared__ uint64_t s[3][FIN_TPB / 64];
    if (lane == 0) { s[0][wave] = xT; s[1][wave] = xV; s[2][wave] = sum; }
    __syncthreads();
    if (tid == 0) {
        uint64_t aT = 0, aV = 0, active = 0;
        for (int w = 0; w < FIN_TPB / 64; ++w) { aT ^= s[0][w]; aV ^= s[1][w]; active += s[2][w]; }
        uint64_t total = 0;
        if (f.cks_T) total ^= sea_one(aT);
        if (f.cks_V) total ^= sea_one(aV);
        total ^= sea_pair(active, f.total_len);
        f.out[2 * (uint64_t)k] = total; f.out[2 * (uint64_t)k + 1] = 0;
    }
}

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
                s.unit(*reinterpret_cast<const uint32_t*>(a.state + col_at(ud.off, ud.ts, ud.stride, e)));
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
    if (word_bytes == 4) *reinterpret_cast<uint32_t*>(state + col_at(col_off, ts, 4, first + i)) = (uint32_t)value;
    else *reinterpret_cast<uint64_t*>(state + col_at(col_off, ts, 8, first + i)) = value;
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

// ------------------------------------------------------------------ k_tick_gen (generic fused request group)
// The same request-group fusion as k_tick for ANY mix of the kernel-backed systems (box_game, add_u32, the Health
// scenario of tests/synctest.rs, particles with extra checksum specs ...): a workgroup stages the `sub` slots it
// owns -- every registered word and every mask -- in LDS (tile-major columns make that `n_words` contiguous global
// spans), replays the ops of the group on the LDS image (Save = LDS -> ring slot + generic checksum partials,
// Advance = each registered system in order over the LDS columns) and writes the live block once at the end.
// One launch per group instead of one per request: an 18-request tick of a small world is 2 launches.
// despawn_rollback systems are covered too: the live-only RollbackDespawned markers of the slots are staged in LDS and
// DespawnConfirmed runs in-kernel before each step.  Not covered (one launch per request): worlds whose words do not
// fit 64 KiB of LDS at 256 slots per workgroup.
constexpr int GEN_MAX_SYS = 16, GEN_MAX_CKS = 16;
// LDS image of a workgroup: the registered word columns in tile order, each `sub` slots long -- a word whose
// preceding words take `pso` bytes per slot starts at byte pso * sub -- then the masks ([n_masks][sub / 64] u64).
struct GenWord { uint32_t tcol, wb, pso, pad; };      // offset inside a tile, word bytes, bytes per slot before it
struct GenUnit { uint32_t pso, add, stride, pad; };   // u32 checksum unit of slot i: pso * sub + add + i * stride
struct GenSys {
    uint32_t kind;
    uint32_t pso[6];          // per-slot byte offsets (GenWord::pso) of the words the system touches (t.x t.y t.z v.x v.y v.z | word)
    uint32_t pmask[3];        // mask index of each component's presence mask; ~0u: live-only component
    uint32_t pso_h, side_ts;  // BOX_MOVE: Player.handle in LDS (rollback component) or in the live block (live-only)
    uint32_t pad;
    uint64_t side_off;        // live-only column the system reads from the live block (BOX_MOVE: Player.handle)
    uint64_t side_pmask_off;  // and its presence mask
    int64_t iparam[2];
    float fparam[4];
};
struct GenArgs {
    const uint8_t* src; uint8_t* live;
    uint8_t* save_dst[MAX_TICK_SAVES]; int32_t save_frame[MAX_TICK_SAVES];
    uint32_t dt_bits[MAX_TICK_STEPS]; uint32_t aux_bits[MAX_TICK_STEPS];      // aux: FRICTION.powf(dt) of BOX_MOVE (host libm)
    uint8_t inputs[MAX_TICK_STEPS][16]; uint8_t n_inputs[MAX_TICK_STEPS];
    // RollbackDespawned markers (snapshot/despawn.rs), staged only for worlds with a despawn_rollback system:
    int32_t step_frame[MAX_TICK_STEPS], step_confirmed[MAX_TICK_STEPS];
    uint8_t step_flags[MAX_TICK_STEPS];            // bit 0: DespawnConfirmed runs before this step; bit 1: its frame is unconfirmed (despawns are deferred)
    uint32_t marks, pad_m; DespawnMarks dm;
    uint64_t op_bits; uint32_t n_ops, n_saves, n_steps, src_is_live;
    uint32_t skip_live, dp_s;                          // skip_live: see TickArgs; dp_s: depth-parallel roles (k_tick1's DP), 0 = off
    uint64_t len, cols_base;
    uint32_t ts, sub, n_words, n_masks, n_units, n_sys, n_cks, part_stride;
    uint64_t mask_off[MAX_MASKS];
    const GenWord* words; const GenUnit* units;
    uint32_t cks_pmask[GEN_MAX_CKS], cks_unit_base[GEN_MAX_CKS], cks_n_units[GEN_MAX_CKS];   // pmask: mask index
    uint64_t* parts;                                   // [n_saves][n_cks + 1][part_stride], one entry per wave; last = live count
    GenSys sys[GEN_MAX_SYS];
};
static_assert(sizeof(GenArgs) <= 4096, "kernel argument segment limit");


// 512-thread workgroups, wave-specialised like k_tick3: waves 0-3 (COMPUTE) hash and step the LDS image, waves 4-7 (STORE)
// own every transfer LDS image -> global.  At a Save: barrier A (the image is stable) -> the store waves pull the whole image
// (<= 64 KiB = 16 chunks of 16 B per lane) and the masks into registers while the compute waves hash it -> barrier B (the
// image may change again) -> the store waves stream it to the ring slot, blocking on the store queue for as long as it
// takes, while the compute waves run the next Advance.
constexpr int GEN_TPB = 512;
__global__ __launch_bounds__(GEN_TPB, 4) void k_tick_gen(GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const bool store_role = threadIdx.x >= (uint32_t)TPB;             // wave-uniform
    const uint32_t tid = threadIdx.x & (TPB - 1u), lane = tid & 63u, wave = tid >> 6;   // index inside the role
    const uint32_t sub = a.sub, mw = sub >> 6;                        // slots / mask words per workgroup
    const uint64_t s0 = (uint64_t)blockIdx.x * sub;                   // first slot of this workgroup
    const uint64_t tbase = a.cols_base + (s0 >> LT_SHIFT) * a.ts;     // its layout tile inside a block
    const uint32_t in_tile = (uint32_t)(s0 & (uint64_t)(LAYOUT_TILE - 1));
    const bool in_len = s0 < a.len;                                   // workgroup-uniform
    // depth-parallel roles (see k_tick1): workgroup (x, y) replays the steps and produces only outputs [y*dp_s, (y+1)*dp_s)
    // of the group (its Saves in order, then the live world).  dp_s == 0: one workgroup produces everything.
    const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
    const uint32_t o_first = a.dp_s ? blockIdx.y * a.dp_s : 0u;
    const uint32_t o_last = a.dp_s ? min(o_first + a.dp_s, a.n_saves + 1u) : a.n_saves + 1u;
    const bool my_live = o_last == a.n_saves + 1u;
    if (a.dp_s && o_first == a.n_saves && !writes_live) return;        // a role with nothing to write
    const uint32_t n_rows = a.ts >> (LT_SHIFT + 2);                   // 4-byte row units per slot (8-byte words = 2)
    const uint32_t img = n_rows * sub * 4u;                           // bytes of the word image
    uint64_t* lmask = reinterpret_cast<uint64_t*>(lds + img);         // [n_masks][mw] behind the words; mask 0 = liveness
    // small tables staged once: global offset of every row unit of this workgroup, the checksum units
    uint32_t* grow = reinterpret_cast<uint32_t*>(lds + img + a.n_masks * mw * 8u);   // [n_rows] byte offset inside the tile
    GenUnit* lunits = reinterpret_cast<GenUnit*>(grow + ((n_rows + 3u) & ~3u));      // [n_units]
    for (uint32_t w = threadIdx.x; w < a.n_words; w += GEN_TPB) {
        const GenWord gw = a.words[w];
        const uint32_t r0 = gw.pso >> 2;
        grow[r0] = gw.tcol + in_tile * gw.wb;
        if (gw.wb == 8) grow[r0 + 1] = gw.tcol + in_tile * 8u + sub * 4u;      // second half of the contiguous sub x 8 bytes
    }
    for (uint32_t u = threadIdx.x; u < a.n_units; u += GEN_TPB) lunits[u] = a.units[u];
    // live-only RollbackDespawned markers of these slots: disabled bits + the frame each was despawned on.  Not part of
    // any snapshot; LoadWorld's resurrect pass (k_load_reconcile) has already run on the live copy (stream order).
    uint64_t* ldis = reinterpret_cast<uint64_t*>(lunits + a.n_units);                 // [mw]
    int32_t* ldf = reinterpret_cast<int32_t*>(ldis + mw);                             // [sub]
    if (a.marks) {
        for (uint32_t m = threadIdx.x; m < mw; m += GEN_TPB) ldis[m] = *reinterpret_cast<const uint64_t*>(a.live + a.dm.off_disabled + ((s0 >> 6) + m) * 8);
        for (uint32_t i = threadIdx.x; i < sub; i += GEN_TPB) ldf[i] = *reinterpret_cast<const int32_t*>(a.live + a.dm.off_dframe + (s0 + i) * 4);
    }
    __syncthreads();

    // LDS image <-> one state block: 16-byte chunks, chunk c lives at LDS byte c * 16 and belongs to row c >> row_shift
    // (sub is 256 / 512 / 1024: a row is 64 / 128 / 256 chunks -- shifts, not the divisions of round 1), 8 chunks in flight
    // per lane.  Snapshot stores are non-temporal like k_tick3's: the ring is written once and read a whole tick later.
    const uint32_t n_chunks = img >> 4;
    const uint32_t row_shift = sub == 1024u ? 8u : (sub == 512u ? 7u : 6u), row_mask = (1u << row_shift) - 1u;
    // global -> LDS image, all 512 threads, 8 loads in flight per lane
    auto stage_in = [&](const uint8_t* block) {
        g_u8* gb = sgpr_base(block + tbase);
        for (uint32_t c0 = threadIdx.x; c0 < n_chunks; c0 += 8 * GEN_TPB) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t c = c0 + j * GEN_TPB;
                v[j] = u32x4{0, 0, 0, 0};
                if (c < n_chunks) v[j] = *(const GGRS_GLOBAL u32x4*)(gb + grow[c >> row_shift] + (c & row_mask) * 16u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t c = c0 + j * GEN_TPB;
                if (c < n_chunks) *reinterpret_cast<u32x4*>(lds + c * 16u) = v[j];
            }
        }
        for (uint32_t m = threadIdx.x; m < a.n_masks * mw; m += GEN_TPB)
            lmask[m] = *reinterpret_cast<const uint64_t*>(block + a.mask_off[m / mw] + ((s0 >> 6) + m % mw) * 8);
    };
    // STORE waves: the image (<= 4096 chunks: 16 per lane) and the masks (<= 17 x 16 words: 2 per lane) in registers
    u32x4 ireg[16]; uint64_t mreg[2];
    auto image_pull = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) { const uint32_t c = tid + (uint32_t)j * TPB; ireg[j] = c < n_chunks ? *reinterpret_cast<const u32x4*>(lds + c * 16u) : u32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const uint32_t m = tid + (uint32_t)j * TPB; mreg[j] = m < a.n_masks * mw ? lmask[m] : 0ULL; }
    };
    auto image_push = [&](uint8_t* block, bool nt, bool words) {
        g_u8* gb = sgpr_base(block + tbase);
        if (words) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t c = tid + (uint32_t)j * TPB;
                if (c < n_chunks) {
                    const uint32_t go = grow[c >> row_shift] + (c & row_mask) * 16u;
                    if (nt) st16<true>(gb, go, ireg[j]); else st16<false>(gb, go, ireg[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t m = tid + (uint32_t)j * TPB;
            if (m < a.n_masks * mw) *reinterpret_cast<uint64_t*>(block + a.mask_off[m / mw] + ((s0 >> 6) + m % mw) * 8) = mreg[j];
        }
    };
    // ---- stage the workgroup's slots
    stage_in(a.src);        // (slots beyond len hold whatever the block holds there: their mask bits are zero)
    __syncthreads();

    if (store_role) {
        // ================================================= STORE waves: one hand-off per Save with a ring slot + the live write
        uint32_t si = 0;
        for (uint32_t op = 0; op < a.n_ops; ++op) {
            if ((a.op_bits >> op) & 1ULL) continue;
            if (si < o_first) { ++si; continue; }                     // another role's snapshot
            if (si >= o_last) break;
            uint8_t* dst = a.save_dst[si];
            if (dst) {
                lds_barrier();                                        // A: the image is stable
                image_pull();
                lds_barrier();                                        // B: (pull has landed: lds_barrier waits lgkmcnt(0)) the image may change
                image_push(dst, true, in_len);
                if (blockIdx.x == 0 && tid == 0) {
                    Header h; h.len = a.len; h.frame = a.save_frame[si]; h.pad0 = 0; h.active = 0; h.checksum[0] = 0; h.checksum[1] = 0;
                    *reinterpret_cast<Header*>(dst) = h;
                }
            }
            ++si;
        }
        if (my_live && writes_live) {
            lds_barrier();
            image_pull();
            lds_barrier();
            image_push(a.live, false, in_len);
        }
    } else {
    // ===================================================== COMPUTE waves
    // diffuse(K0 ^ order) of the (up to 4) slots this lane hashes: it depends on the slot only, so one value serves every
    // checksummed component and every Save of the group (as in k_tick)
    uint64_t ordB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ordB[q] = sea_order_lane(s0 + tid + (uint32_t)q * TPB);

    uint32_t si = 0, sj = 0;
    for (uint32_t op = 0; op < a.n_ops; ++op) {
        if (!((a.op_bits >> op) & 1ULL)) {
            // ---------------- SaveWorld: snapshot (store waves) + per-entity half of every component checksum (here).
            // Barrier A hands the stable image to the store waves; the hash below only READS it, so it overlaps their pull;
            // barrier B (after the hash) lets the next Advance change it.
            if (si < o_first) { ++si; continue; }                     // another role's snapshot
            if (si >= o_last) break;                                  // this role's Saves are out (a later role owns the rest)
            const bool hand_off = a.save_dst[si] != nullptr;
            if (hand_off) lds_barrier();
            uint64_t* prow = a.parts + (uint64_t)si * (a.n_cks + 1) * a.part_stride + (uint64_t)blockIdx.x * 4 + wave;
            for (uint32_t k = 0; k < a.n_cks; ++k) {
                const uint64_t* pm = lmask + a.cks_pmask[k] * mw;
                const uint32_t nu = a.cks_n_units[k], ub = a.cks_unit_base[k];
                uint64_t h = 0;
                if (nu <= 4u) {
                    // the common specs (1-4 four-byte units, e.g. translation.xyz): unit addresses hoisted out of the slot
                    // loop, the (up to 4) slots of a lane hashed as independent chains, dead slots selected away
                    const uint8_t* ubase[4]; uint32_t ustride[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const GenUnit gu = lunits[ub + ((uint32_t)u < nu ? (uint32_t)u : 0u)];
                        ubase[u] = lds + gu.pso * sub + gu.add; ustride[u] = gu.stride;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t i = tid + (uint32_t)q * TPB;
                        if (i < sub) {                                     // workgroup-uniform (sub is 256, 512 or 1024)
                            const bool on = ((lmask[i >> 6] & pm[i >> 6]) >> (i & 63u)) & 1ULL;
                            uint32_t x[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const uint32_t*>(ubase[u] + i * ustride[u]);
                            SeaStream st;
                            st.unit(x[0]);
                            if (nu > 1u) st.unit(x[1]);
                            if (nu > 2u) st.unit(x[2]);
                            if (nu > 3u) st.unit(x[3]);
                            const uint64_t e = sea_pair_pre(ordB[q], st.finish());
                            h ^= on ? e : 0ULL;
                        }
                    }
                } else {
                    for (uint32_t i = tid; i < sub; i += TPB) {
                        if (((lmask[i >> 6] & pm[i >> 6]) >> (i & 63u)) & 1ULL) {
                            SeaStream st;
                            for (uint32_t u = 0; u < nu; ++u) {
                                const GenUnit gu = lunits[ub + u];
                                st.unit(*reinterpret_cast<const uint32_t*>(lds + gu.pso * sub + gu.add + i * gu.stride));
                            }
                            h ^= sea_pair(s0 + i, st.finish());          // order == slot
                        }
                    }
                }
                h = wave_xor(h);
                if (lane == 0) prow[(uint64_t)k * a.part_stride] = h;
            }
            uint32_t cnt = 0;
            for (uint32_t wi = wave; wi < mw; wi += 4) cnt += (uint32_t)__popcll(lmask[wi]);
            if (lane == 0) prow[(uint64_t)a.n_cks * a.part_stride] = cnt;   // folded by k_gen_finalize (an in-kernel tick_fold measured no gain here: 178 vs 174 us per 1 M tick)
            if (hand_off) lds_barrier();
            ++si;
            if (si >= o_last) break;                                  // (no point stepping an image nobody will read)
        } else {
            // ---------------- AdvanceWorld: the registered systems, in order, on the LDS image
            const float dt = __uint_as_float(a.dt_bits[sj]);
            const uint32_t sflags = a.step_flags[sj];
            if (a.marks && (sflags & 1u)) {
                // AdvanceWorldSystems::DespawnConfirmed (despawn.rs:89-112): marks <= ConfirmedFrameCount are freed for good
                const int32_t confirmed = a.step_confirmed[sj];
                for (uint32_t i = tid; i < sub; i += TPB) {
                    const uint32_t wi = i >> 6;
                    const uint64_t dw = ldis[wi];
                    const uint64_t gone = __ballot(((dw >> (i & 63u)) & 1ULL) && ldf[i] <= confirmed);
                    if (gone && lane == 0) ldis[wi] = dw & ~gone;
                }
            }
            for (uint32_t s = 0; s < a.n_sys; ++s) {
                const GenSys& y = a.sys[s];
                const uint64_t* p0 = y.pmask[0] != ~0u ? lmask + y.pmask[0] * mw : lmask;
                const uint64_t* p1 = y.pmask[1] != ~0u ? lmask + y.pmask[1] * mw : lmask;
                for (uint32_t i = tid; i < sub; i += TPB) {               // a wave's 64 lanes == one mask word
                    const uint32_t wi = i >> 6;
                    const uint64_t alive_w = lmask[wi];
                    bool kill = false;
                    switch (y.kind) {
                    case 1u: {   // GGRS_SYS_PARTICLES_UPDATE (particles.rs:272-280)
                        if (((alive_w & p0[wi] & p1[wi]) >> (i & 63u)) & 1ULL) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                float* x = reinterpret_cast<float*>(lds + y.pso[k] * sub + i * 4u);
                                float* v = reinterpret_cast<float*>(lds + y.pso[3 + k] * sub + i * 4u);
                                const float nv = __fadd_rn(*v, __fmul_rn(y.fparam[k], dt));
                                *v = nv; *x = __fadd_rn(*x, __fmul_rn(nv, dt));
                            }
                        }
                    } break;
                    case 2u: {   // GGRS_SYS_TTL_DESPAWN (particles.rs:282-289)
                        if (((alive_w & p0[wi]) >> (i & 63u)) & 1ULL) {
                            uint64_t* q = reinterpret_cast<uint64_t*>(lds + y.pso[0] * sub + i * 8u);
                            const uint64_t nq = *q - 1; *q = nq; kill = nq == 0;
                        }
                    } break;
                    case 4u: {   // GGRS_SYS_ADD_U32 (benches/bench.rs:30-46)
                        if (((alive_w & p0[wi]) >> (i & 63u)) & 1ULL) {
                            uint32_t* q = reinterpret_cast<uint32_t*>(lds + y.pso[0] * sub + i * 4u);
                            *q = *q + (uint32_t)y.iparam[0];
                        }
                    } break;
                    case 5u: {   // GGRS_SYS_SAT_SUB_DESPAWN (tests/synctest.rs:37-44): despawn() or despawn_rollback()
                        if (((alive_w & p0[wi]) >> (i & 63u)) & 1ULL) {
                            uint32_t* q = reinterpret_cast<uint32_t*>(lds + y.pso[0] * sub + i * 4u);
                            const uint32_t amount = (uint32_t)y.iparam[0];
                            const uint32_t v = *q >= amount ? *q - amount : 0u;
                            *q = v; kill = v == 0;
                        }
                        // despawn_rollback() on an unconfirmed frame (despawn.rs:129-137): disabled, marked with the frame
                        const bool defer = a.marks && y.iparam[1] == 1 && (sflags & 2u);
                        const uint64_t marked = __ballot(kill && defer);
                        if (kill && defer) ldf[i] = a.step_frame[sj];
                        if (marked && lane == 0) ldis[wi] |= marked;
                    } break;
                    case 6u: {   // GGRS_SYS_BOX_MOVE (box_game.rs:154-206), arithmetic shared with k_box_move
                        const uint64_t e = s0 + i;
                        uint64_t m = alive_w & p0[wi] & p1[wi];
                        const bool h_lds = y.pmask[2] != ~0u;                // Player registered for rollback: staged in LDS
                        if (h_lds) m &= lmask[y.pmask[2] * mw + wi];
                        else m &= *reinterpret_cast<const uint64_t*>(a.live + y.side_pmask_off + (e >> 6) * 8);
                        if ((m >> (i & 63u)) & 1ULL) {
                            const uint64_t handle = h_lds ? *reinterpret_cast<const uint64_t*>(lds + y.pso_h * sub + i * 8u)
                                                          : *reinterpret_cast<const uint64_t*>(a.live + col_at(y.side_off, y.side_ts, 8, e));
                            if (handle < a.n_inputs[sj]) {
                                float* px = reinterpret_cast<float*>(lds + y.pso[0] * sub + i * 4u);
                                float* py = reinterpret_cast<float*>(lds + y.pso[1] * sub + i * 4u);
                                float* pz = reinterpret_cast<float*>(lds + y.pso[2] * sub + i * 4u);
                                float* pvx = reinterpret_cast<float*>(lds + y.pso[3] * sub + i * 4u);
                                float* pvy = reinterpret_cast<float*>(lds + y.pso[4] * sub + i * 4u);
                                float* pvz = reinterpret_cast<float*>(lds + y.pso[5] * sub + i * 4u);
                                float x = *px, yy = *py, z = *pz, vx = *pvx, vy = *pvy, vz = *pvz;
                                box_move_math(x, yy, z, vx, vy, vz, a.inputs[sj][handle], dt, __uint_as_float(a.aux_bits[sj]),
                                              y.fparam[0], y.fparam[1], y.fparam[3]);
                                *px = x; *py = yy; *pz = z; *pvx = vx; *pvy = vy; *pvz = vz;
                            }
                        }
                    } break;
                    default: break;
                    }
                    const uint64_t kills = __ballot(kill);
                    if (kills && lane == 0) lmask[wi] = alive_w & ~kills;   // word wi is only ever read by this wave: no barrier
                }
            }
            ++sj;
        }
    }
    // ---- the live block, written once (by the store waves)
    if (my_live && writes_live) { lds_barrier(); lds_barrier(); }
    if (my_live && a.marks && a.n_steps) {
        for (uint32_t m = tid; m < mw; m += TPB) *reinterpret_cast<uint64_t*>(a.live + a.dm.off_disabled + ((s0 >> 6) + m) * 8) = ldis[m];
        for (uint32_t i = tid; i < sub; i += TPB) *reinterpret_cast<int32_t*>(a.live + a.dm.off_dframe + (s0 + i) * 4) = ldf[i];
    }
    }   // COMPUTE waves
}

// Fold of k_tick_gen's per-wave partials: one 1024-thread workgroup per Save; any number of checksummed components.
struct GenFinArgs {
    const uint64_t* parts; uint32_t part_stride, n_parts, n_cks, pad;
    uint64_t total_len;
    uint64_t* out;
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
        for (uint32_t i0 = (wave % wpr) * 64u + lane; i0 < f.n_parts; i0 += 4u * wpr * 64u) {
            uint64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t i = i0 + (uint32_t)u * wpr * 64u; v[u] = i < f.n_parts ? p[i] : 0ULL; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { x ^= v[u]; sum += v[u]; }
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
        total ^= sea_pair(acc[f.n_cks], f.total_len);                        // entity_checksum.rs:29-52; XOR fold checksum.rs:88-99
        f.out[2 * (uint64_t)k] = total; f.out[2 * (uint64_t)k + 1] = 0;
    }
}

}  // namespace ggrs
