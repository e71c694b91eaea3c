// ubench2.hip -- which store pattern gets the k_tick traffic shape (read 1 block, write 8 ring blocks + live)
// closest to what hipMemset reaches (6.5 TB/s write-only)?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

constexpr int ROWS = 15;                 // 4 KiB rows per tile (60 KiB)
constexpr size_t TILE_V = ROWS * 256;    // u32x4 per tile

// ---- pure fills
// A: one tile per WG (grid = tiles), lane-interleaved 16 B  (the engine's pattern)
template <bool NT> __global__ __launch_bounds__(256) void fill_tile(u32x4* dst) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    u32x4* p = dst + (size_t)blockIdx.x * TILE_V + threadIdx.x;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) st<NT>(p + r * 256, v);
}
// B: persistent grid-stride over tiles
template <bool NT> __global__ __launch_bounds__(256) void fill_persist(u32x4* dst, int tiles) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        u32x4* p = dst + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) st<NT>(p + r * 256, v);
    }
}
// C: persistent, each WG owns a CONTIGUOUS chunk of tiles
template <bool NT> __global__ __launch_bounds__(256) void fill_chunk(u32x4* dst, int tiles) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    const int per = (tiles + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(tiles, t0 + per);
    for (int t = t0; t < t1; ++t) {
        u32x4* p = dst + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) st<NT>(p + r * 256, v);
    }
}
// D: each lane writes 64 contiguous bytes (4 x 16 B)
template <bool NT> __global__ __launch_bounds__(256) void fill_lane64(u32x4* dst) {
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    u32x4* p = dst + (size_t)blockIdx.x * TILE_V + threadIdx.x * 4;
    for (int r = 0; r + 4 <= ROWS + 1; r += 4) {          // 16 rows' worth; last quarter clipped below
#pragma unroll
        for (int j = 0; j < 4; ++j) if ((size_t)(r * 256 + threadIdx.x * 4 + j) < TILE_V) st<NT>(p + r * 256 + j, v);
    }
}

// ---- fan-out: read tile from src block, write to D ring slots + live
// layout 0: slot-major blocks (slot d at d * bs);  layout 1: interleaved (tile t of slot d at (t * D + d) * TILE_V)
template <bool NT, int LAYOUT, bool ROWMAJOR_ORDER>
__global__ __launch_bounds__(256) void fan(const u32x4* src, u32x4* ring, u32x4* live, size_t bs, int D, int tiles) {
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        u32x4 v[ROWS];
        const u32x4* s = src + (LAYOUT ? (size_t)t * D * TILE_V : (size_t)t * TILE_V) + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) v[j] = s[j * 256];
        if (!ROWMAJOR_ORDER) {
            for (int d = 0; d < D; ++d) {
                u32x4* p = ring + (LAYOUT ? ((size_t)t * D + d) * TILE_V : (size_t)d * bs + (size_t)t * TILE_V) + threadIdx.x;
#pragma unroll
                for (int j = 0; j < ROWS; ++j) { v[j].x += d; st<NT>(p + j * 256, v[j]); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                for (int d = 0; d < D; ++d) {
                    u32x4* p = ring + (LAYOUT ? ((size_t)t * D + d) * TILE_V : (size_t)d * bs + (size_t)t * TILE_V) + threadIdx.x;
                    st<NT>(p + j * 256, v[j]);
                }
        }
        u32x4* l = live + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) st<false>(l + j * 256, v[j]);
    }
}


// fan + the small stores k_tick issues per Save: MASKS: 4 x 32 B per wave into 4 mask regions of the destination
// block; PARTS: 3 x 8 B per wave into three partial arrays
template <bool NT, bool MASKS, bool PARTS, bool MASKLINE>
__global__ __launch_bounds__(256) void fan_small(const u32x4* src, u32x4* ring, u32x4* live, size_t bs, int D, int tiles,
                                                 uint64_t* masks, size_t mask_stride, uint64_t* parts, size_t part_stride) {
    const int t = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 v[ROWS];
    const u32x4* s = src + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) v[j] = s[j * 256];
    for (int d = 0; d < D; ++d) {
        u32x4* p = ring + (size_t)d * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { v[j].x += d; st<NT>(p + j * 256, v[j]); }
        if (MASKS) {
            uint64_t* m = masks + (size_t)d * 4 * mask_stride;
            if (MASKLINE) {   // one full 128 B line per wave: 4 masks x 32 B interleaved per 256 slots
                if (lane < 16) m[((size_t)t * 4 + wave) * 16 + lane] = v[0].y + lane;
            } else {
                if (lane < 4) m[(size_t)t * 16 + wave * 4 + lane] = v[0].y;
                if ((lane & 15) == 0) {
                    m[mask_stride + (size_t)t * 16 + wave * 4 + (lane >> 4)] = v[1].y;
                    m[2 * mask_stride + (size_t)t * 16 + wave * 4 + (lane >> 4)] = v[2].y;
                    m[3 * mask_stride + (size_t)t * 16 + wave * 4 + (lane >> 4)] = v[3].y;
                }
            }
        }
        if (PARTS && lane == 0) {
            uint64_t* q = parts + (size_t)d * 3 * part_stride + (size_t)t * 4 + wave;
            q[0] = v[0].z; q[part_stride] = v[1].z; q[2 * part_stride] = v[2].z;
        }
    }
    u32x4* l = live + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) st<false>(l + j * 256, v[j]);
}

// the engine's real rotation: iteration i reads slot (i % 9) -- written 8 iterations ago, like the snapshot a
// SyncTest tick loads -- and writes the other 8 slots + live, so the source is never Infinity-Cache resident by reuse
template <bool NT>
__global__ __launch_bounds__(256) void fan_rot(u32x4* ring, u32x4* live, size_t bs, int src_slot, int tiles) {
    const int t = blockIdx.x;
    u32x4 v[ROWS];
    const u32x4* s = ring + (size_t)src_slot * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) v[j] = s[j * 256];
    for (int k = 1; k <= 8; ++k) {
        const int d = (src_slot + k) % 9;
        u32x4* p = ring + (size_t)d * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { v[j].x += d; st<NT>(p + j * 256, v[j]); }
    }
    u32x4* l = live + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) st<false>(l + j * 256, v[j]);
}

// same rotation, but the block the NEXT launch will read (src + 1) is stored LAST instead of first: is a block
// that was just written still cache-resident (L2 / Infinity Cache) when the next launch reads it?
template <bool NT>
__global__ __launch_bounds__(256) void fan_rot_last(u32x4* ring, u32x4* live, size_t bs, int src_slot, int tiles) {
    const int t = blockIdx.x;
    u32x4 v[ROWS];
    const u32x4* s = ring + (size_t)src_slot * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) v[j] = s[j * 256];
    for (int k = 2; k <= 9; ++k) {
        const int d = (src_slot + (k == 9 ? 1 : k)) % 9;
        u32x4* p = ring + (size_t)d * bs + (size_t)t * TILE_V + threadIdx.x;
        if (k == 9) {      // the next source: after the live block
            u32x4* l = live + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) st<false>(l + j * 256, v[j]);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { v[j].x += d; st<NT>(p + j * 256, v[j]); }
    }
}

// the engine's rotation WITH the SeaHash work k_tick does per Save (40 diffuse = 80 u64 multiplies per lane): does the
// ALU burst between store bursts change what non-temporal stores buy?
__device__ __forceinline__ uint64_t diffuse(uint64_t x) {
    x *= 0x6eed0e9da4d94a4fULL; x ^= (x >> 32) >> (x >> 60); x *= 0x6eed0e9da4d94a4fULL; return x;
}
template <bool NT>
__global__ __launch_bounds__(256) void fan_rot_hash(u32x4* ring, u32x4* live, size_t bs, int src_slot, int tiles, uint64_t* sink) {
    const int t = blockIdx.x;
    u32x4 v[ROWS];
    const u32x4* s = ring + (size_t)src_slot * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) v[j] = s[j * 256];
    uint64_t acc = 0;
    for (int k = 1; k <= 8; ++k) {
        const int d = (src_slot + k) % 9;
        u32x4* p = ring + (size_t)d * bs + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { v[j].x += d; st<NT>(p + j * 256, v[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                       // 8 rows x 4 lanes-words... 40 diffuses in chains of 5
            uint64_t h = ((uint64_t)v[j].x << 32) | v[j].y;
            h = diffuse(h ^ 0x16f11fe89b0d677cULL); h = diffuse(h ^ v[j].z); h = diffuse(h ^ 0xb480a793d8e6c86cULL);
            h = diffuse(h ^ v[j].w); h = diffuse(h ^ (uint64_t)k);
            acc ^= h;
        }
    }
    u32x4* l = live + (size_t)t * TILE_V + threadIdx.x;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) st<false>(l + j * 256, v[j]);
    if (acc == 0x1234567ULL) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int tiles = 977;
    const size_t bs = (size_t)tiles * TILE_V;            // u32x4 per block
    const size_t block = bs * 16;
    const int D = 8;
    u32x4 *ring, *live; 
    CK(hipMalloc(&ring, block * (D + 1))); CK(hipMalloc(&live, block));
    CK(hipMemset(ring, 1, block * (D + 1))); CK(hipMemset(live, 2, block));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-64s %8.1f us  %7.1f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    printf("ring=%p live=%p block=%zu\n", (void*)ring, (void*)live, block);
    const double W8 = 8.0 * block;
    timeit("hipMemsetAsync 8 blocks", W8, [&](int i) { (void)hipMemsetAsync(ring, i, block * 8, 0); });
    timeit("fill 8 blocks: tile/WG", W8, [&](int) { hipLaunchKernelGGL(fill_tile<false>, tiles * 8, 256, 0, 0, ring); });
    timeit("fill 8 blocks: tile/WG nt", W8, [&](int) { hipLaunchKernelGGL(fill_tile<true>, tiles * 8, 256, 0, 0, ring); });
    for (int g : {256, 512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, 96, "fill 8 blocks: persistent grid-stride, %d WGs", g);
        timeit(nm, W8, [&](int) { hipLaunchKernelGGL(fill_persist<false>, g, 256, 0, 0, ring, tiles * 8); });
        snprintf(nm, 96, "fill 8 blocks: persistent grid-stride nt, %d WGs", g);
        timeit(nm, W8, [&](int) { hipLaunchKernelGGL(fill_persist<true>, g, 256, 0, 0, ring, tiles * 8); });
        snprintf(nm, 96, "fill 8 blocks: persistent contiguous chunks, %d WGs", g);
        timeit(nm, W8, [&](int) { hipLaunchKernelGGL(fill_chunk<false>, g, 256, 0, 0, ring, tiles * 8); });
    }
    timeit("fill 8 blocks: 64 B per lane", W8, [&](int) { hipLaunchKernelGGL(fill_lane64<false>, tiles * 8, 256, 0, 0, ring); });
    const double F = 10.0 * block;
    const u32x4* src0 = ring + 8 * bs;                    // slot-major: read slot 8
    timeit("fan ROTATING src (engine-like), default stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    timeit("fan ROTATING src (engine-like), nt stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot<true>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    timeit("fan ROTATING src (engine-like), default stores again", F, [&](int i) { hipLaunchKernelGGL(fan_rot<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    timeit("fan ROTATING, next source stored LAST, default stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot_last<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    timeit("fan ROTATING, next source stored LAST, nt stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot_last<true>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    timeit("fan ROTATING src (engine-like), default stores #3", F, [&](int i) { hipLaunchKernelGGL(fan_rot<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles); });
    uint64_t* sinkp; CK(hipMalloc(&sinkp, 64));
    timeit("fan ROTATING + 40 diffuse per Save, default stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot_hash<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles, sinkp); });
    timeit("fan ROTATING + 40 diffuse per Save, nt stores", F, [&](int i) { hipLaunchKernelGGL(fan_rot_hash<true>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles, sinkp); });
    timeit("fan ROTATING + 40 diffuse per Save, default stores #2", F, [&](int i) { hipLaunchKernelGGL(fan_rot_hash<false>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles, sinkp); });
    timeit("fan ROTATING + 40 diffuse per Save, nt stores #2", F, [&](int i) { hipLaunchKernelGGL(fan_rot_hash<true>, tiles, 256, 0, 0, ring, live, bs, i % 9, tiles, sinkp); });
    timeit("fan slot-major", F, [&](int) { hipLaunchKernelGGL((fan<false, 0, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles); });
    timeit("fan slot-major nt", F, [&](int) { hipLaunchKernelGGL((fan<true, 0, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles); });
    timeit("fan slot-major row-order", F, [&](int) { hipLaunchKernelGGL((fan<false, 0, true>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles); });
    timeit("fan interleaved", F, [&](int) { hipLaunchKernelGGL((fan<false, 1, false>), tiles, 256, 0, 0, ring, ring, live, bs, D, tiles); });
    timeit("fan interleaved nt", F, [&](int) { hipLaunchKernelGGL((fan<true, 1, false>), tiles, 256, 0, 0, ring, ring, live, bs, D, tiles); });
    timeit("fan interleaved row-order", F, [&](int) { hipLaunchKernelGGL((fan<false, 1, true>), tiles, 256, 0, 0, ring, ring, live, bs, D, tiles); });
    {
        uint64_t *masks, *parts;
        const size_t mask_stride = (size_t)tiles * 16 + 64, part_stride = (size_t)tiles * 4 + 64;
        CK(hipMalloc(&masks, 9 * 4 * mask_stride * 8)); CK(hipMalloc(&parts, 9 * 3 * part_stride * 8));
        timeit("fan_small: no small stores", F, [&](int) { hipLaunchKernelGGL((fan_small<false, false, false, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small: + masks (4 x 32 B per wave per save)", F, [&](int) { hipLaunchKernelGGL((fan_small<false, true, false, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small: + masks as one 128 B line per wave", F, [&](int) { hipLaunchKernelGGL((fan_small<false, true, false, true>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small: + parts (3 x 8 B per wave per save)", F, [&](int) { hipLaunchKernelGGL((fan_small<false, false, true, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small: + masks + parts", F, [&](int) { hipLaunchKernelGGL((fan_small<false, true, true, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small nt: no small stores", F, [&](int) { hipLaunchKernelGGL((fan_small<true, false, false, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small nt: + masks + parts", F, [&](int) { hipLaunchKernelGGL((fan_small<true, true, true, false>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
        timeit("fan_small nt: + masks as 128 B lines + parts", F, [&](int) { hipLaunchKernelGGL((fan_small<true, true, true, true>), tiles, 256, 0, 0, src0, ring, live, bs, D, tiles, masks, mask_stride, parts, part_stride); });
    }
    for (int g : {256, 512}) {
        char nm[96];
        snprintf(nm, 96, "fan slot-major persistent %d WGs", g);
        timeit(nm, F, [&](int) { hipLaunchKernelGGL((fan<false, 0, false>), g, 256, 0, 0, src0, ring, live, bs, D, tiles); });
        snprintf(nm, 96, "fan slot-major persistent nt %d WGs", g);
        timeit(nm, F, [&](int) { hipLaunchKernelGGL((fan<true, 0, false>), g, 256, 0, 0, src0, ring, live, bs, D, tiles); });
        snprintf(nm, 96, "fan interleaved persistent %d WGs", g);
        timeit(nm, F, [&](int) { hipLaunchKernelGGL((fan<false, 1, false>), g, 256, 0, 0, ring, ring, live, bs, D, tiles); });
        snprintf(nm, 96, "fan interleaved persistent nt %d WGs", g);
        timeit(nm, F, [&](int) { hipLaunchKernelGGL((fan<true, 1, false>), g, 256, 0, 0, ring, ring, live, bs, D, tiles); });
    }
    return 0;
}
