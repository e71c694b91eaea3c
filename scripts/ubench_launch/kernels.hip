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

// RESIDENT-KERNEL MAILBOX (DESIGN "open leads": would a kernel that stays on the device and is fed ticks through a mailbox beat one launch per tick for a small
// world?).  40 workgroups stay resident; per tick, workgroup 0 polls `cmd` (pinned host memory, written by the host) until it holds the tick's number and releases the
// others through a device-memory flag; every workgroup then "works" for n x 100 ns (the tick kernel's duration) and counts itself done; the last one writes `done`
// (pinned) for the host to poll.  Bounded: leaves after `ticks` ticks or 3 s, whichever comes first.
struct MailArgs { volatile unsigned long long* cmd; volatile unsigned long long* done; unsigned long long* dev; unsigned int n, ticks; };
extern "C" __global__ __launch_bounds__(256) void k_mailbox(MailArgs a) {
    const unsigned long long t_begin = wall_clock64();
    __shared__ unsigned long long s_go;
    for (unsigned long long k = 1; k <= a.ticks; ++k) {
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) {
                while (__hip_atomic_load(a.cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < k) { if (wall_clock64() - t_begin > 300000000ull) break; }
                __hip_atomic_store(a.dev, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);                       // release the other workgroups
            } else {
                while (__hip_atomic_load(a.dev, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < k) { if (wall_clock64() - t_begin > 300000000ull) break; }
            }
            s_go = k;
        }
        __syncthreads();
        if (wall_clock64() - t_begin > 300000000ull) return;
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)a.n) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long arrived = __hip_atomic_fetch_add(a.dev + 1, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
            if (arrived == k * gridDim.x) __hip_atomic_store(a.done, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // the tick's last workgroup tells the host
        }
    }
}
