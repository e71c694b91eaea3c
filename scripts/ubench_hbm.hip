// ubench_hbm.hip -- what can the MI355X memory system sustain for the traffic SHAPES of this engine?
// (write-only snapshot streams, 1-read -> D-write fan-out, plain copy).  Build: hipcc --offload-arch=gfx950 -O3
// Output feeds DESIGN.md section 6 (the "achievable" ceiling next to the 8 TB/s spec peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// each WG owns `rows` consecutive 4 KiB rows (256 lanes x 16 B); grid = bytes / (rows * 4096)
template <bool NT> __global__ __launch_bounds__(256) void k_fill(u32x4* dst, int rows) {
    const size_t base = ((size_t)blockIdx.x * rows) * 256 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int r = 0; r < rows; ++r) st<NT>(dst + base + (size_t)r * 256, v);
}
__global__ __launch_bounds__(256) void k_read(const u32x4* src, int rows, uint32_t* sink) {
    const size_t base = ((size_t)blockIdx.x * rows) * 256 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rows; ++r) { u32x4 v = src[base + (size_t)r * 256]; acc ^= v; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
template <bool NT> __global__ __launch_bounds__(256) void k_copy(const u32x4* src, u32x4* dst, int rows) {
    const size_t base = ((size_t)blockIdx.x * rows) * 256 + threadIdx.x;
    for (int r0 = 0; r0 + 5 <= rows; r0 += 5) {        // rows must be a multiple of 5
        u32x4 v[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) v[j] = src[base + (size_t)(r0 + j) * 256];
#pragma unroll
        for (int j = 0; j < 5; ++j) st<NT>(dst + base + (size_t)(r0 + j) * 256, v[j]);
    }
}
// the k_tick shape: read `rows` rows once, write them to D destination blocks (stride dst_stride u32x4)
template <bool NT> __global__ __launch_bounds__(256) void k_fan(const u32x4* src, u32x4* dst, size_t dst_stride, int D, int rows) {
    const size_t base = ((size_t)blockIdx.x * rows) * 256 + threadIdx.x;
    u32x4 v[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) v[j] = src[base + (size_t)j * 256];
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int j = 0; j < 15; ++j) { v[j].x += d; st<NT>(dst + (size_t)d * dst_stride + base + (size_t)j * 256, v[j]); }
    }
}

// the same fan-out over a COLUMN-MAJOR block: row j of tile t lives at j * col_stride + t * 256 (u32x4 units),
// i.e. 15 separate 4 MB column arrays per block -- the engine's round-1 layout
template <bool NT> __global__ __launch_bounds__(256) void k_fan_cols(const u32x4* src, u32x4* dst, size_t dst_stride, int D, size_t col_stride) {
    const size_t base = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32x4 v[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) v[j] = src[base + (size_t)j * col_stride];
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int j = 0; j < 15; ++j) { v[j].x += d; st<NT>(dst + (size_t)d * dst_stride + base + (size_t)j * col_stride, v[j]); }
    }
}

int main() {
    const size_t block = 60ull * 1000 * 1024;          // ~one 1M-entity state block (15 rows x 4 KiB x 1000 tiles)
    const int D = 9;
    u32x4 *ring, *live; uint32_t* sink;
    CK(hipMalloc(&ring, block * D)); CK(hipMalloc(&live, block)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(ring, 1, block * D)); CK(hipMemset(live, 2, block));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-46s %8.1f us  %7.1f GB/s\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
    };
    const int rows = 15;                                // rows per WG (one tile of the particles world)
    const int grid = (int)(block / (rows * 4096));
    const size_t bs = block / 16;                       // block size in u32x4
    for (int nt = 0; nt < 2; ++nt) {
        printf("---- %s stores\n", nt ? "non-temporal" : "default");
        timeit("fill 1 block (rotating over the 9-block ring)", (double)block, [&](int i) { if (nt) hipLaunchKernelGGL(k_fill<true>, grid, 256, 0, 0, ring + (size_t)(i % D) * bs, rows); else hipLaunchKernelGGL(k_fill<false>, grid, 256, 0, 0, ring + (size_t)(i % D) * bs, rows); });
        timeit("fill 8 blocks in one launch", (double)block * 8, [&](int i) { if (nt) hipLaunchKernelGGL(k_fill<true>, grid * 8, 256, 0, 0, ring, rows); else hipLaunchKernelGGL(k_fill<false>, grid * 8, 256, 0, 0, ring, rows); });
        timeit("copy live -> ring slot (rotating)", 2.0 * block, [&](int i) { if (nt) hipLaunchKernelGGL(k_copy<true>, grid, 256, 0, 0, live, ring + (size_t)(i % D) * bs, rows); else hipLaunchKernelGGL(k_copy<false>, grid, 256, 0, 0, live, ring + (size_t)(i % D) * bs, rows); });
        timeit("copy ring slot -> live (rotating, cold reads)", 2.0 * block, [&](int i) { if (nt) hipLaunchKernelGGL(k_copy<true>, grid, 256, 0, 0, ring + (size_t)(i % D) * bs, live, rows); else hipLaunchKernelGGL(k_copy<false>, grid, 256, 0, 0, ring + (size_t)(i % D) * bs, live, rows); });
        timeit("fan-out: read 1 block, write 8 ring blocks", 9.0 * block, [&](int i) { if (nt) hipLaunchKernelGGL(k_fan<true>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 8, rows); else hipLaunchKernelGGL(k_fan<false>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 8, rows); });
        timeit("fan-out COLUMN-MAJOR: read 1, write 8 + 1", 10.0 * block, [&](int i) { if (nt) hipLaunchKernelGGL(k_fan_cols<true>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 9, bs / 15); else hipLaunchKernelGGL(k_fan_cols<false>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 9, bs / 15); });
        timeit("fan-out: read 1, write 8 ring + 1 live (k_tick)", 10.0 * block, [&](int i) { if (nt) hipLaunchKernelGGL(k_fan<true>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 9, rows); else hipLaunchKernelGGL(k_fan<false>, grid, 256, 0, 0, ring + (size_t)8 * bs, ring, bs, 9, rows); });
    }
    timeit("read 1 block (rotating, cold)", (double)block, [&](int i) { hipLaunchKernelGGL(k_read, grid, 256, 0, 0, ring + (size_t)(i % D) * bs, rows, sink); });
    timeit("read 8 blocks in one launch", (double)block * 8, [&](int i) { hipLaunchKernelGGL(k_read, grid * 8, 256, 0, 0, ring, rows, sink); });
    timeit("hipMemsetAsync 8 blocks", (double)block * 8, [&](int i) { (void)hipMemsetAsync(ring, i, block * 8, 0); });
    return 0;
}
