// ubench_launch -- what one launch costs the HOST on this platform, against kernarg size and launch flavour
// ("measure the launch call against kernarg size at 10 k").
//   * hipModuleLaunchKernel vs hipExtModuleLaunchKernel with a stop event (what an enqueued request list's last kernel carries)
//   * kernarg blocks of 64 B .. 4 KiB
//   * the pipelined tick loop of a small world: launch(k+1), wait(event k) around a ~5.6 us kernel
// Build: make -C scripts/ubench_launch     Run on the GPU box: ./scripts/ubench_launch/ubench_launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "scripts/ubench_launch/kernels.hsaco";
    hipModule_t mod; CK(hipModuleLoad(&mod, path));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* d_out; CK(hipMalloc((void**)&d_out, 1 << 20));
    const int sizes[] = {64, 256, 512, 1024, 2112, 4096};
    std::vector<unsigned char> buf(4096, 0);
    *(unsigned long long**)buf.data() = d_out;
    printf("{\n");
    for (int s : sizes) {
        char name[32]; snprintf(name, sizeof name, "k_args_%d", s);
        hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, name));
        size_t sz = (size_t)s;
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        void* params[] = {buf.data()};
        for (int flavour = 0; flavour < 3; ++flavour) {        // 0: hipModuleLaunchKernel(params), 1: ...(extra buffer), 2: hipExtModuleLaunchKernel + stop event
            hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            const int N = 20000;
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipStreamSynchronize(st));
                const double t0 = now_us();
                for (int i = 0; i < N; ++i) {
                    if (flavour == 0) CK(hipModuleLaunchKernel(fn, 40, 1, 1, 256, 1, 1, 0, st, params, nullptr));
                    else if (flavour == 1) CK(hipModuleLaunchKernel(fn, 40, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
                    else CK(hipExtModuleLaunchKernel(fn, 40 * 256, 1, 1, 256, 1, 1, 0, st, params, nullptr, nullptr, ev, 0));
                }
                const double t1 = now_us();
                CK(hipStreamSynchronize(st));
                const double t2 = now_us();
                const double per = (t2 - t0) / N;
                if (per < best) best = per;
                if (rep == 2) printf("  \"args%d_%s\": {\"us_per_launch_incl_drain\": %.3f, \"host_call_us\": %.3f},\n", s, flavour == 0 ? "params" : flavour == 1 ? "extra" : "ext_event", best, (t1 - t0) / N);
            }
            CK(hipEventDestroy(ev));
        }
    }
    // ---- the pipelined tick of a small world: one ~kernel_us kernel per tick, launch(k+1) then wait(event k)
    hipFunction_t busy; CK(hipModuleGetFunction(&busy, mod, "k_busy"));
    for (int kernel_100ns : {20, 56, 80}) {
        struct { unsigned long long* out; unsigned int n, pad; unsigned char fill[48]; } a; memset(&a, 0, sizeof a);
        a.out = d_out; a.n = (unsigned)kernel_100ns * 10;       // wall_clock64 ticks at 100 MHz: 10 per us
        void* params[] = {&a};
        hipEvent_t ev[4]; for (int q = 0; q < 4; ++q) CK(hipEventCreateWithFlags(&ev[q], hipEventDisableTiming));
        const int N = 20000;
        for (int mode = 0; mode < 5; ++mode) {                  // 0: event on the launch (ext) + hipEventSynchronize, 1: hipEventRecord behind it, 2: ext + hipEventQuery spin,
                                                                // 3: ext + sync, TWO ticks in flight, 4: ext + query spin with 1.0 us of host work per tick
            CK(hipStreamSynchronize(st));
            const double t0 = now_us();
            auto launch = [&](int k) -> hipError_t {
                if (mode != 1) return hipExtModuleLaunchKernel(busy, 40 * 256, 1, 1, 256, 1, 1, 0, st, params, nullptr, nullptr, ev[k & 3], 0);
                hipError_t e = hipModuleLaunchKernel(busy, 40, 1, 1, 256, 1, 1, 0, st, params, nullptr);
                return e != hipSuccess ? e : hipEventRecord(ev[k & 3], st);
            };
            auto wait = [&](int k) -> hipError_t {
                if (mode == 2 || mode == 4) { hipError_t e; while ((e = hipEventQuery(ev[k & 3])) == hipErrorNotReady) __builtin_ia32_pause(); return e; }
                return hipEventSynchronize(ev[k & 3]);
            };
            const int ahead = mode == 3 ? 2 : 1;
            for (int k = 0; k < ahead; ++k) CK(launch(k));
            for (int k = ahead; k < N; ++k) {
                if (mode == 4) { const double w0 = now_us(); while (now_us() - w0 < 1.0) {} }
                CK(launch(k)); CK(wait(k - ahead));
            }
            for (int k = N - ahead; k < N; ++k) CK(wait(k));
            const double per = (now_us() - t0) / N;
            const char* names[] = {"ext_event_sync", "event_record_sync", "ext_event_query_spin", "ext_event_sync_2_in_flight", "ext_event_query_spin_plus_1us_host"};
            printf("  \"pipelined_tick_kernel_%.1fus_%s\": %.3f,\n", kernel_100ns / 10.0, names[mode], per);
        }
        for (int q = 0; q < 4; ++q) CK(hipEventDestroy(ev[q]));
    }
    {   // ---- the resident-kernel mailbox form of the same tick: no launch per tick, the host writes a tick number into pinned memory and polls another
        hipFunction_t mail; CK(hipModuleGetFunction(&mail, mod, "k_mailbox"));
        volatile unsigned long long* h; CK(hipHostMalloc((void**)&h, 4096, hipHostMallocMapped));
        unsigned long long* d_h; CK(hipHostGetDevicePointer((void**)&d_h, (void*)h, 0));
        unsigned long long* d_flags; CK(hipMalloc((void**)&d_flags, 256));
        for (int kernel_100ns : {20, 56, 80}) {
            const int N = 20000;
            h[0] = 0; h[64] = 0;                                  // cmd and done in different cache lines
            CK(hipMemset(d_flags, 0, 256));
            struct { volatile unsigned long long* cmd; volatile unsigned long long* done; unsigned long long* dev; unsigned int n, ticks; } a;
            a.cmd = d_h; a.done = d_h + 64; a.dev = d_flags; a.n = (unsigned)kernel_100ns * 10; a.ticks = N;
            void* params[] = {&a};
            CK(hipModuleLaunchKernel(mail, 40, 1, 1, 256, 1, 1, 0, st, params, nullptr));
            const double t0 = now_us();
            bool ok = true;
            for (unsigned long long k = 1; k <= (unsigned long long)N && ok; ++k) {
                __atomic_store_n(&h[0], k, __ATOMIC_RELEASE);
                const double w0 = now_us();
                while (__atomic_load_n(&h[64], __ATOMIC_ACQUIRE) < k) { __builtin_ia32_pause(); if (now_us() - w0 > 2e6) { ok = false; break; } }
            }
            const double per = (now_us() - t0) / N;
            CK(hipStreamSynchronize(st));
            printf("  \"resident_mailbox_tick_kernel_%.1fus\": %s%.3f,\n", kernel_100ns / 10.0, ok ? "" : "-", per);
        }
        CK(hipFree(d_flags)); CK(hipHostFree((void*)h));
    }
    {   // what the two waits cost on an event that is already complete
        hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, "k_args_64"));
        void* params[] = {buf.data()};
        CK(hipExtModuleLaunchKernel(fn, 256, 1, 1, 256, 1, 1, 0, st, params, nullptr, nullptr, e, 0));
        CK(hipStreamSynchronize(st));
        double t0 = now_us(); for (int i = 0; i < 100000; ++i) (void)hipEventQuery(e); const double q_us = (now_us() - t0) / 100000;
        t0 = now_us(); for (int i = 0; i < 100000; ++i) (void)hipEventSynchronize(e); const double s_us = (now_us() - t0) / 100000;
        printf("  \"complete_event_query_us\": %.3f, \"complete_event_synchronize_us\": %.3f,\n", q_us, s_us);
    }
    printf("  \"end\": 0\n}\n");
    return 0;
}
