// kernels of scripts/ubench_launch: one trivial kernel per kernarg size (bytes), built as a code object (hipcc --genco)
#include <hip/hip_runtime.h>
template <int N> struct Args { unsigned long long* out; unsigned int n, pad; unsigned char fill[N - 16]; };
#define K(N) extern "C" __global__ __launch_bounds__(256) void k_args_##N(Args<N> a) { if (a.n == 0xFFFFFFFFu && threadIdx.x == 0) a.out[blockIdx.x] = a.fill[0]; }
K(64) K(256) K(512) K(1024) K(2112) K(4096)
// a kernel that runs for roughly `n` x 100 ns (s_sleep), to emulate a 5-6 us tick kernel
extern "C" __global__ __launch_bounds__(256) void k_busy(Args<64> a) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)a.n) __builtin_amdgcn_s_sleep(2);
    if (a.pad == 0xFFFFFFFFu && threadIdx.x == 0) a.out[blockIdx.x] = t0;
}
