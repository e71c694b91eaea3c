// ubench3.hip -- store-path study for k_tick's traffic shape on MI355X (round 2).
//
// Question: hipMemset writes at ~6.5 TB/s, a fill kernel that gives every workgroup its own contiguous 60 KiB tile at
// 5.1-5.5 TB/s, and k_tick (read 1 block, write 8 ring blocks + the live block) at 4.7-5.0 TB/s.  Which property of the
// store stream costs the difference: the number of concurrent write fronts, how dense a front is (lane-interleaved
// sweep vs one tile per workgroup), the cache policy of the stores (plain / nt / sc1), the number of resident
// workgroups, or the ALU work between store bursts?  Every kernel below moves the same bytes; only the pattern differs.
//
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench3.hip -o scripts/ubench3        Run: ./scripts/ubench3 [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 15;                         // 4 KiB rows per 1024-slot tile (60 B / slot)
constexpr size_t ROW_V = 256;                    // u32x4 per row
constexpr size_t TILE_V = ROWS * ROW_V;          // u32x4 per tile (60 KiB)

enum { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, NTSC1 = 4 };
template <int K> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (K == PLAIN) *p = v;
    else if (K == NT) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else if (K == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if (K == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
}
template <int K> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if (K == NT) return __builtin_nontemporal_load(p);
    return *p;
}

// ------------------------------------------------------------------ write-only fills over F "fronts" (blocks)
// LIN: the hipMemset pattern -- the whole grid sweeps one dense window (grid x 4 KiB) through the block
template <int K> __global__ __launch_bounds__(256) void fill_lin(u32x4* base, size_t n_vec, int fronts, size_t stride_v, int fronts_inner) {
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    const size_t step = (size_t)gridDim.x * 256;
    if (fronts_inner) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += step)
            for (int f = 0; f < fronts; ++f) st<K>(base + f * stride_v + i, v);
    } else {
        for (int f = 0; f < fronts; ++f)
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += step) st<K>(base + f * stride_v + i, v);
    }
}
// TILE: every workgroup owns whole 60 KiB tiles (the engine's tile-major pattern); order 0: block-major (one tile of
// block 0, then the same tile of block 1 ...), order 1: row-major (row r of every block, then row r + 1)
template <int K> __global__ __launch_bounds__(256) void fill_tile(u32x4* base, int tiles, int fronts, size_t stride_v, int order) {
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        u32x4* p = base + (size_t)t * TILE_V + threadIdx.x;
        if (order == 0) {
            for (int f = 0; f < fronts; ++f)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) st<K>(p + f * stride_v + r * ROW_V, v);
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                for (int f = 0; f < fronts; ++f) st<K>(p + f * stride_v + r * ROW_V, v);
        }
    }
}

// ------------------------------------------------------------------ the engine's shape: read 1 block, write 8 + live
// Layouts of a block (T tiles):
//   0 tile-major     row r of tile t at  t * TILE_V + r * ROW_V                         (current engine)
//   1 column-major   row r of tile t at  r * (T * ROW_V) + t * ROW_V                    (round-1 first layout)
//   2 super-tiles    groups of G tiles, rows interleaved inside the group:
//                    (t / G) * G * TILE_V + r * (G * ROW_V) + (t % G) * ROW_V
struct Fan {
    u32x4* ring; u32x4* live; size_t bs_v;       // ring of 9 blocks at bs_v strides (vec units)
    int src_slot, tiles, layout, G, alu, stagger, rest_first;
};
__device__ __forceinline__ size_t row_off(const Fan& a, int t, int r) {
    if (a.layout == 0) return (size_t)t * TILE_V + (size_t)r * ROW_V;
    if (a.layout == 1) return (size_t)r * ((size_t)a.tiles * ROW_V) + (size_t)t * ROW_V;
    return (size_t)(t / a.G) * a.G * TILE_V + (size_t)r * ((size_t)a.G * ROW_V) + (size_t)(t % a.G) * ROW_V;
}
__device__ __forceinline__ uint64_t churn(uint64_t x, int n) {      // n dependent 64-bit multiplies (SeaHash stand-in)
    for (int i = 0; i < n; ++i) { x *= 0x6eed0e9da4d94a4fULL; x ^= x >> 29; }
    return x;
}
template <int K, int KL, int MINW>
__global__ __launch_bounds__(256, MINW) void fan(Fan a) {
    extern __shared__ uint8_t lds_dummy[];
    const uint32_t wave = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < a.tiles; t += gridDim.x) {
        u32x4 v[ROWS];
        const u32x4* s = a.ring + (size_t)a.src_slot * a.bs_v + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) v[j] = ld<KL>(s + row_off(a, t, j));
        uint64_t h = v[0].x;
        for (int k = 1; k <= 8; ++k) {
            const int d = (a.src_slot + k) % 9;
            u32x4* p = a.ring + (size_t)d * a.bs_v + threadIdx.x;
            const bool hash_first = a.stagger && (wave & 1u);
            if (a.alu && hash_first) h = churn(h + k, a.alu);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { if (j < 8) v[j].x += d; st<K>(p + row_off(a, t, j), v[j]); }
            if (a.alu && !hash_first) h = churn(h + k, a.alu);
        }
        v[0].y ^= (uint32_t)h;
        u32x4* l = a.live + threadIdx.x;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) st<PLAIN>(l + row_off(a, t, j), v[j]);
    }
}

// Wave specialisation: a 512-thread workgroup, waves 0-3 "compute" (hash stand-in: 8 independent chains of alu/8
// dependent u64 multiplies per Save, then the 8 schedule-owned rows of the tile go to LDS), waves 4-7 "store" (7
// untouched rows stay in their registers; per Save: 8 rows from LDS + 7 from registers -> the ring slot).  One
// s_barrier per Save, double-buffered LDS: the compute waves never wait for a global store to be accepted.
template <int K>
__global__ __launch_bounds__(512) void fan_spec(Fan a) {
    __shared__ u32x4 buf[2][8][256];                 // 2 x 32 KiB
    const uint32_t tid = threadIdx.x & 255u;
    const bool store_role = threadIdx.x >= 256;
    for (int t = blockIdx.x; t < a.tiles; t += gridDim.x) {
        u32x4 v[8]; u32x4 r[7];
        const u32x4* s = a.ring + (size_t)a.src_slot * a.bs_v + tid;
        if (!store_role) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s[row_off(a, t, j)];
        } else {
#pragma unroll
            for (int j = 0; j < 7; ++j) r[j] = s[row_off(a, t, 8 + j)];
        }
        uint64_t h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = c + 1;
        for (int k = 1; k <= 9; ++k) {               // 8 snapshots + the live block
            if (!store_role) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j].x += k; buf[k & 1][j][tid] = v[j]; }
            }
            __syncthreads();
            if (store_role) {
                const bool live = k == 9;
                u32x4* p = (live ? a.live : a.ring + (size_t)((a.src_slot + k) % 9) * a.bs_v) + tid;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const u32x4 x = buf[k & 1][j][tid]; if (live) st<PLAIN>(p + row_off(a, t, j), x); else st<K>(p + row_off(a, t, j), x); }
#pragma unroll
                for (int j = 0; j < 7; ++j) { if (live) st<PLAIN>(p + row_off(a, t, 8 + j), r[j]); else st<K>(p + row_off(a, t, 8 + j), r[j]); }
            } else if (k < 9 && a.alu) {
#pragma unroll
                for (int c = 0; c < 8; ++c) h[c] = churn(h[c] + v[c].y, a.alu / 8);
#pragma unroll
                for (int c = 0; c < 8; ++c) { v[c].z ^= (uint32_t)h[c]; if (a.rest_first) { v[c].x ^= (uint32_t)(h[c] >> 7); v[c].y ^= (uint32_t)(h[c] >> 13); v[c].w ^= (uint32_t)(h[c] >> 32); } }
            }
        }
        __syncthreads();
    }
}

__global__ void init_random(u32x4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ULL + 12345; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 29;
        uint64_t y = x * 0x94D049BB133111EBULL; y ^= y >> 32;
        p[i] = u32x4{(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32)};
    }
}
struct Dev { u32x4* ring; u32x4* live; size_t bs_v; int tiles; };

template <class F> float time_us(int iters, F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f(i + 3);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters * 1e3f;
}

static const char* kname(int k) { static const char* n[] = {"plain", "nt", "sc1", "sc0sc1", "nt+sc1"}; return n[k]; }
static std::string g_filter;
static bool want(const char* name) { return g_filter.empty() || strstr(name, g_filter.c_str()); }

template <int K> void run_fills(const Dev& d) {
    const size_t n_vec = (size_t)d.tiles * TILE_V;
    const double mb = n_vec * 16.0 / 1e6;
    char name[128];
    for (int fronts : {1, 2, 4, 9}) {
        for (int grid : {256, 512, 1024, 2048, 4096}) {
            for (int inner : {0, 1}) {
                if (fronts == 1 && inner) continue;
                snprintf(name, sizeof name, "fill_lin   %-6s fronts=%d grid=%-5d %s", kname(K), fronts, grid, inner ? "fronts-inner" : "front-by-front");
                if (!want(name)) continue;
                float us = time_us(10, [&](int) { hipLaunchKernelGGL(fill_lin<K>, dim3(grid), dim3(256), 0, 0, d.ring, n_vec, fronts, d.bs_v, inner); });
                printf("%-64s %8.1f us  %6.2f TB/s\n", name, us, fronts * mb / us);
            }
        }
        for (int grid : {256, 768, 1024, d.tiles}) {
            for (int order : {0, 1}) {
                if (fronts == 1 && order) continue;
                snprintf(name, sizeof name, "fill_tile  %-6s fronts=%d grid=%-5d %s", kname(K), fronts, grid, order ? "row-major" : "block-major");
                if (!want(name)) continue;
                float us = time_us(10, [&](int) { hipLaunchKernelGGL(fill_tile<K>, dim3(grid), dim3(256), 0, 0, d.ring, d.tiles, fronts, d.bs_v, order); });
                printf("%-64s %8.1f us  %6.2f TB/s\n", name, us, fronts * mb / us);
            }
        }
    }
}

template <int K, int KL, int MINW> void run_fan(const Dev& d, int layout, int G, int grid, int alu, int stagger, int lds) {
    char name[160];
    snprintf(name, sizeof name, "fan %-6s ld=%-5s layout=%d G=%-4d grid=%-5d minw=%d lds=%-6d alu=%-3d stagger=%d", kname(K), kname(KL), layout, G, grid, MINW, lds, alu, stagger);
    if (!want(name)) return;
    Fan a; a.ring = d.ring; a.live = d.live; a.bs_v = d.bs_v; a.tiles = d.tiles; a.layout = layout; a.G = G; a.alu = alu; a.stagger = stagger; a.rest_first = 0;
    float us = time_us(18, [&](int i) { Fan b = a; b.src_slot = i % 9; hipLaunchKernelGGL((fan<K, KL, MINW>), dim3(grid), dim3(256), lds, 0, b); });
    const double mb = (double)d.tiles * TILE_V * 16.0 * 10 / 1e6;
    printf("%-100s %8.1f us  %6.2f TB/s\n", name, us, mb / us);
}

template <int K> void run_fan_spec(const Dev& d, int layout, int G, int grid, int alu, int scramble = 0) {
    char name[160];
    snprintf(name, sizeof name, "fan_spec %-6s layout=%d G=%-4d grid=%-5d alu=%-3d scramble=%d (4 compute + 4 store waves)", kname(K), layout, G, grid, alu, scramble);
    if (!want(name)) return;
    Fan a; a.ring = d.ring; a.live = d.live; a.bs_v = d.bs_v; a.tiles = d.tiles; a.layout = layout; a.G = G; a.alu = alu; a.stagger = 0; a.rest_first = scramble;
    float us = time_us(18, [&](int i) { Fan b = a; b.src_slot = i % 9; hipLaunchKernelGGL((fan_spec<K>), dim3(grid), dim3(512), 0, 0, b); });
    const double mb = (double)d.tiles * TILE_V * 16.0 * 10 / 1e6;
    printf("%-100s %8.1f us  %6.2f TB/s\n", name, us, mb / us);
}

int main(int argc, char** argv) {
    if (argc > 1) g_filter = argv[1];
    const int tiles = 977;
    // the engine's block stride for 1 M entities: header + 4 masks (4 KiB aligned) + 977 tiles x 60 KiB
    const size_t bs = 503808 + (size_t)tiles * 61440;
    Dev d; d.tiles = tiles; d.bs_v = bs / 16;
    if (getenv("UB_CONTIG") && atoi(getenv("UB_CONTIG"))) CK(hipExtMallocWithFlags((void**)&d.ring, bs * 10 + (64 << 20), hipDeviceMallocContiguous));
    else CK(hipMalloc((void**)&d.ring, bs * 10 + (64 << 20)));   // slack: layout 2 rounds the tile count up to a multiple of G
    printf("ring at %p (%s)\n", (void*)d.ring, getenv("UB_CONTIG") ? "contiguous" : "hipMalloc");
    d.live = d.ring + 9 * d.bs_v;
    CK(hipMemset(d.ring, 1, bs * 10));
    if (getenv("UB_RANDOM") && atoi(getenv("UB_RANDOM"))) { hipLaunchKernelGGL(init_random, dim3(4096), dim3(256), 0, 0, d.ring, (bs * 10) / 16); CK(hipDeviceSynchronize()); printf("ring initialised with random data\n"); }
    // reference points: hipMemset / hipMemcpy of one block
    {
        float us = time_us(10, [&](int) { CK(hipMemsetAsync(d.ring, 0, bs, 0)); });
        printf("%-64s %8.1f us  %6.2f TB/s\n", "hipMemsetAsync one block", us, bs / 1e6 / us);
        us = time_us(10, [&](int) { CK(hipMemsetAsync(d.ring, 0, bs * 9, 0)); });
        printf("%-64s %8.1f us  %6.2f TB/s\n", "hipMemsetAsync nine blocks", us, bs * 9 / 1e6 / us);
        us = time_us(10, [&](int i) { CK(hipMemcpyAsync(d.ring + (size_t)((i % 8) + 1) * d.bs_v, d.ring, bs, hipMemcpyDeviceToDevice, 0)); });
        printf("%-64s %8.1f us  %6.2f TB/s (read+write)\n", "hipMemcpyAsync D2D one block", us, 2 * bs / 1e6 / us);
    }
    run_fills<PLAIN>(d);
    run_fills<NT>(d);
    run_fills<SC1>(d);

    // ---- the engine's shape
    const int T = tiles;
    for (int layout : {0, 1}) {
        run_fan<PLAIN, PLAIN, 1>(d, layout, 0, T, 0, 0, 0);
        run_fan<NT, PLAIN, 1>(d, layout, 0, T, 0, 0, 0);
        run_fan<SC1, PLAIN, 1>(d, layout, 0, T, 0, 0, 0);
        run_fan<NTSC1, PLAIN, 1>(d, layout, 0, T, 0, 0, 0);
        run_fan<NT, NT, 1>(d, layout, 0, T, 0, 0, 0);
    }
    for (int G : {2, 4, 8, 16, 32, 128, 256}) {
        run_fan<PLAIN, PLAIN, 1>(d, 2, G, T, 0, 0, 0);
        run_fan<NT, PLAIN, 1>(d, 2, G, T, 0, 0, 0);
    }
    // resident workgroups per CU through dynamic LDS (160 KiB / n): 1, 2, 3, 4, 6 per CU
    for (int lds : {150 * 1024, 76 * 1024, 52 * 1024, 38 * 1024, 25 * 1024}) {
        run_fan<PLAIN, PLAIN, 1>(d, 0, 0, T, 0, 0, lds);
        run_fan<NT, PLAIN, 1>(d, 0, 0, T, 0, 0, lds);
    }
    // persistent grids (every workgroup resident from the start, loops over tiles)
    for (int grid : {256, 512, 768, 1024}) {
        run_fan<PLAIN, PLAIN, 1>(d, 0, 0, grid, 0, 0, 0);
        run_fan<NT, PLAIN, 1>(d, 0, 0, grid, 0, 0, 0);
        run_fan<NT, PLAIN, 1>(d, 1, 0, grid, 0, 0, 0);
    }
    // ALU between the store bursts (80 dependent u64 multiplies per Save ~ the SeaHash work of 4 slots x 2 components)
    for (int alu : {40, 80, 160}) {
        for (int stagger : {0, 1}) {
            run_fan<PLAIN, PLAIN, 1>(d, 0, 0, T, alu, stagger, 52 * 1024);
            run_fan<NT, PLAIN, 1>(d, 0, 0, T, alu, stagger, 52 * 1024);
            run_fan<NT, PLAIN, 1>(d, 0, 0, T, alu, stagger, 38 * 1024);
        }
    }
    // wave specialisation vs the uniform kernel at the same ALU load, on the engine's layout (2, G = 8)
    for (int alu : {0, 80, 160}) {
        for (int grid : {256, 512, 977}) {
            run_fan_spec<NT>(d, 2, 8, grid, alu);
            run_fan_spec<NT>(d, 2, 8, grid, alu, 1);
            run_fan_spec<PLAIN>(d, 2, 8, grid, alu);
        }
        run_fan<NT, PLAIN, 1>(d, 2, 8, T, alu, 1, 52 * 1024);
        run_fan<NT, PLAIN, 1>(d, 2, 8, T, alu, 1, 0);
        run_fan<NT, PLAIN, 1>(d, 2, 8, 768, alu, 1, 0);
    }
    CK(hipFree(d.ring));
    return 0;
}
