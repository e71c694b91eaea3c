// ubench_alu.hip -- the chip's ceiling for the checksum arithmetic: SeaHash `diffuse` (two 64-bit multiplies by a constant +
// a variable shift-xor) per second over all CUs, and the instruction mix behind it (v_mul_lo_u32 / v_mad_u64_u32 rates).
// Used to state a roofline for the checksum-only paths (BASELINE config 5 after dead-snapshot elimination) and to judge how
// far the hash ALU of a fused tick is from its floor.   build: hipcc --offload-arch=gfx950 -O3 ubench_alu.hip -o ubench_alu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr uint64_t P = 0x6eed0e9da4d94a4fULL;
// the kernels' own spelling (bevy_ggrs_amd/csrc/device_prelude.hpp sea_diffuse): the variable shift in 32-bit terms -- a ceiling measured with the 64-bit form
// (v_lshrrev_b64 per diffuse: rounds 4-5) was one the kernels could exceed (config 5 read 1.02 of it in profiles/r06final)
__device__ __forceinline__ uint64_t diffuse(uint64_t x) { x *= P; const uint32_t hi = (uint32_t)(x >> 32); x ^= (uint64_t)(hi >> (hi >> 28)); x *= P; return x; }

template <int CH>
__global__ __launch_bounds__(256) void k_diffuse(uint64_t* out, int iters) {
    uint64_t x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = (uint64_t)(blockIdx.x * 256 + threadIdx.x) * 0x9e3779b97f4a7c15ull + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = diffuse(x[c] ^ (uint64_t)i);
    }
    uint64_t r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r ^= x[c];
    if (r == 0x1234567) out[0] = r;          // never true: keeps the chains alive
}
template <int CH>
__global__ __launch_bounds__(256) void k_mul32(uint32_t* out, int iters) {           // v_mul_lo_u32 chains
    uint32_t x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = x[c] * 0xa4d94a4fu + (uint32_t)i;
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r ^= x[c];
    if (r == 0x1234567) out[0] = r;
}
template <int CH>
__global__ __launch_bounds__(256) void k_fma32(float* out, int iters) {              // v_fma_f32 chains (the full-rate reference)
    float x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = __builtin_fmaf(x[c], 1.0000001f, 0.5f);
    }
    float r = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) r += x[c];
    if (r == 0.1234f) out[0] = r;
}

template <class F>
double time_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main() {
    int n_cu = 0; hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    void* out; hipMalloc(&out, 4096);
    const int iters = 4096;
    printf("CUs %d, clock attr %d kHz\n", n_cu, clk);
    printf("%-28s %8s %12s %14s %14s\n", "kernel", "waves/SIMD", "ms", "G ops/s", "cyc/op/SIMD@2.4GHz");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = n_cu * wps;          // 256-thread blocks = 4 waves = 1 per SIMD
        auto report = [&](const char* name, double ms, double ops_per_thread) {
            const double ops = (double)blocks * 256 * ops_per_thread;     // lane-ops
            const double wave_ops_per_simd = ops / 64 / (n_cu * 4);
            printf("%-28s %8d %12.4f %14.1f %14.2f\n", name, wps, ms, ops / ms / 1e6, ms * 1e-3 * 2.4e9 / wave_ops_per_simd);
        };
        report("diffuse x1 chain", time_ms([&] { hipLaunchKernelGGL(k_diffuse<1>, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters); }), iters * 1.0);
        report("diffuse x4 chains", time_ms([&] { hipLaunchKernelGGL(k_diffuse<4>, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters); }), iters * 4.0);
        report("diffuse x8 chains", time_ms([&] { hipLaunchKernelGGL(k_diffuse<8>, dim3(blocks), dim3(256), 0, 0, (uint64_t*)out, iters); }), iters * 8.0);
        report("v_mul_lo_u32 x8 chains", time_ms([&] { hipLaunchKernelGGL(k_mul32<8>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)out, iters); }), iters * 8.0);
        report("v_fma_f32 x8 chains", time_ms([&] { hipLaunchKernelGGL(k_fma32<8>, dim3(blocks), dim3(256), 0, 0, (float*)out, iters); }), iters * 8.0);
    }
    return 0;
}
