// ubench4.hip -- what bounds the row-version tick (ggrs_jit_tick, round 3): 1 M slots, per slot 32 B read once (six 4-byte
// words + one 8-byte word), the same 32 B stored to 8 ring blocks and the live block, 4 mask words per 64 slots per block,
// optionally the tick's hash work (8 SeaHash diffuses per slot and Save).  Every variant moves the same bytes; they differ in
//   V      slots per lane (1: 4-byte stores as the generated kernel does, 2: 8-byte, 4: 16-byte)
//   MASK   0 no mask stores, 1 four single-lane 8-byte stores per wave and Save (the generated kernel), 2 one 4-lane store
//   ALU    diffuses per slot and Save (0 / 8)
//   layout 0 the engine's (8192-slot layout tiles: a workgroup's piece of a row is 1 KiB, rows 32 KiB apart)
//          1 workgroup-contiguous (the 7 rows of a workgroup's 256 slots back to back: one 8 KiB piece per block)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench4.hip -o scripts/ubench4      Run: ./scripts/ubench4 [slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NB = 9;                                   // 8 ring blocks + live
struct Args { uint8_t* src; uint8_t* dst[NB]; uint64_t n_slots; uint64_t col_off[7]; uint64_t ts; uint64_t mask_off[4]; int layout; };

__device__ __forceinline__ uint64_t diffuse(uint64_t x) {
    x *= 0x6eed0e9da4d94a4fULL; x ^= (x >> 32) >> (x >> 60); x *= 0x6eed0e9da4d94a4fULL; return x;
}
template <int B> struct Vec;
template <> struct Vec<4> { typedef uint32_t T; };
template <> struct Vec<8> { typedef u32x2 T; };
template <> struct Vec<16> { typedef u32x4 T; };
template <int B> __device__ __forceinline__ void stnt(uint8_t* p, typename Vec<B>::T v) { __builtin_nontemporal_store(v, reinterpret_cast<typename Vec<B>::T*>(p)); }

// one wave = 64 * V consecutive slots; lane l owns slots [l*V, l*V + V) of them
template <int V, int MASK, int ALU>
__global__ __launch_bounds__(256) void tick(Args a) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t unit = ((uint64_t)blockIdx.x * 4 + wave);                 // 64*V slots
    const uint64_t s0 = unit * 64 * V + (uint64_t)lane * V;
    if (unit * 64 * V >= a.n_slots) return;
    // address of word c of slot s (4-byte words c = 0..5, 8-byte word c = 6)
    auto addr = [&](uint64_t s, int c) -> uint64_t {
        const uint32_t wb = c == 6 ? 8u : 4u;
        if (a.layout == 0) return a.col_off[c] + (s >> 13) * a.ts + (s & 8191) * wb;
        const uint64_t g = s >> 8, i = s & 255;                              // workgroup-contiguous: 8 KiB per 256 slots
        return g * 8192 + (c == 6 ? 6144 + i * 8 : (uint64_t)c * 1024 + i * 4);
    };
    typename Vec<4 * V>::T w[6]; typename Vec<4 * V>::T t2[2];               // the 8-byte word as two 4*V-byte halves (V slots x 8 B)
#pragma unroll
    for (int c = 0; c < 6; ++c) w[c] = *reinterpret_cast<const typename Vec<4 * V>::T*>(a.src + addr(s0, c));
    t2[0] = *reinterpret_cast<const typename Vec<4 * V>::T*>(a.src + addr(s0, 6));
    t2[1] = *reinterpret_cast<const typename Vec<4 * V>::T*>(a.src + addr(s0, 6) + 4 * V);
    uint64_t mk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) mk[m] = MASK ? *reinterpret_cast<const uint64_t*>(a.src + a.mask_off[m] + unit * V * 8) : 0;
    uint64_t acc = 0;
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
        uint8_t* d = a.dst[b];
#pragma unroll
        for (int c = 0; c < 6; ++c) stnt<4 * V>(d + addr(s0, c), w[c]);
        stnt<4 * V>(d + addr(s0, 6), t2[0]); stnt<4 * V>(d + addr(s0, 6) + 4 * V, t2[1]);
        if (MASK == 1) { if (lane == 0) { for (int m = 0; m < 4; ++m) for (int v = 0; v < V; ++v) *reinterpret_cast<uint64_t*>(d + a.mask_off[m] + (unit * V + v) * 8) = mk[m] + b; } }
        if (MASK == 2) { if (lane < 4) { const uint64_t val = lane == 0 ? mk[0] : lane == 1 ? mk[1] : lane == 2 ? mk[2] : mk[3];
                                         for (int v = 0; v < V; ++v) *reinterpret_cast<uint64_t*>(d + a.mask_off[lane] + (unit * V + v) * 8) = val + b; } }
        if (ALU) {
            const uint32_t* x = reinterpret_cast<const uint32_t*>(&w[0]);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                uint64_t h = ((uint64_t)x[v] << 32) | x[V + v] | b;
#pragma unroll
                for (int k = 0; k < ALU; ++k) h = diffuse(h ^ k);
                acc ^= h;
            }
        }
        // the "advance": registers change between Saves
#pragma unroll
        for (int c = 0; c < 6; ++c) w[c] += 1u;
    }
    if (ALU && acc == 0x1234567ull) *reinterpret_cast<uint64_t*>(a.dst[0]) = acc;    // keep the hash alive
}

// ---- the same traffic with the generated kernel's skeleton added step by step (V = 1, engine layout only):
//   S >= 1  saddr-form inline-asm nt stores (wave-uniform base in an SGPR pair + one 32-bit lane offset), as kernel_gen.hpp emits them
//   S >= 2  the request-group loop: op_bits walks Save / Advance ops, destination and row mask of every Save come from the kernarg
//           segment by a dynamic index (s_load per op)
//   S >= 3  per-workgroup checksum partials: LDS rows zeroed + __syncthreads up front, DPP-free wave XOR (shfl) + one LDS atomic per
//           Save and component, partial rows written to global at the end
struct Args2 { uint8_t* src; uint8_t* save_dst[16]; uint64_t save_rows[16]; uint32_t dt_bits[16]; uint64_t op_bits; uint32_t n_ops, n_saves; uint8_t* live;
               uint64_t n_slots; uint64_t col_off[7]; uint64_t ts; uint64_t mask_off[4]; uint64_t* parts; uint32_t part_stride; };
__device__ __forceinline__ void st4s(const uint8_t* base, uint32_t lo, uint32_t v) { const unsigned long b = (unsigned long)base; asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(lo), "v"(v), "s"(b) : "memory"); }
__device__ __forceinline__ void st8s(const uint8_t* base, uint32_t lo, uint64_t v) { const unsigned long b = (unsigned long)base; asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(lo), "v"(v), "s"(b) : "memory"); }
template <int S, int MASK, int ALU>
__global__ __launch_bounds__(256) void tick2(Args2 a) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    __shared__ uint64_t s_acc[16 * 3];
    if (S >= 3) { for (uint32_t i = tid; i < 48u; i += 256u) s_acc[i] = 0; __syncthreads(); }
    const uint32_t tile = blockIdx.x;
    const uint32_t gu = tile * 4u + wave;
    if ((uint64_t)gu * 64 >= a.n_slots) return;
    const uint64_t tbase = (uint64_t)(gu >> 7) * a.ts;
    const uint32_t ei = (gu & 127u) * 64u + lane, lo4 = ei * 4u, lo8 = ei * 8u;
    const uint64_t wi8 = (uint64_t)gu * 8u;
    uint32_t w[6]; uint64_t t;
#pragma unroll
    for (int c = 0; c < 6; ++c) w[c] = *reinterpret_cast<const uint32_t*>(a.src + a.col_off[c] + tbase + lo4);
    t = *reinterpret_cast<const uint64_t*>(a.src + a.col_off[6] + tbase + lo8);
    uint64_t mk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) mk[m] = MASK ? *reinterpret_cast<const uint64_t*>(a.src + a.mask_off[m] + wi8) : 0;
    uint64_t acc = 0;
    auto save = [&](uint8_t* d, uint32_t b) {
#pragma unroll
        for (int c = 0; c < 6; ++c) st4s(d + a.col_off[c] + tbase, lo4, w[c]);
        st8s(d + a.col_off[6] + tbase, lo8, t);
        if (MASK == 1 && lane == 0) { for (int m = 0; m < 4; ++m) *reinterpret_cast<uint64_t*>(d + a.mask_off[m] + wi8) = mk[m] + b; }
    };
    auto hash = [&](uint32_t b, uint32_t si) {
        if (!ALU) return;
        uint64_t h0 = ((uint64_t)w[0] << 32) | w[1] | b, h1 = ((uint64_t)w[3] << 32) | w[4] | b;
#pragma unroll
        for (int k = 0; k < ALU / 2; ++k) { h0 = diffuse(h0 ^ k); h1 = diffuse(h1 ^ k); }
        if (S >= 3) {
            for (int o = 32; o; o >>= 1) { h0 ^= __shfl_xor(h0, o); h1 ^= __shfl_xor(h1, o); }
            if (lane == 0) { atomicXor((unsigned long long*)&s_acc[si * 3], h0); atomicXor((unsigned long long*)&s_acc[si * 3 + 1], h1); atomicAdd((unsigned long long*)&s_acc[si * 3 + 2], 64ull); }
        } else acc ^= h0 ^ h1;
    };
    if (S >= 2) {
        uint32_t si = 0, sj = 0;
        for (uint32_t op = 0; op < a.n_ops; ++op) {
            if (!((a.op_bits >> op) & 1ull)) {
                uint8_t* d = a.save_dst[si];
                if (d && a.save_rows[si] == 0x3c07ull) save(d, si);
                hash(si, si);
                ++si;
            } else {
                const float dt = __uint_as_float(a.dt_bits[sj]);
#pragma unroll
                for (int c = 0; c < 6; ++c) w[c] = __float_as_uint(__uint_as_float(w[c]) + dt);
                t -= 1; ++sj;
            }
        }
        save(a.live, 8);
    } else {
#pragma unroll 1
        for (uint32_t b = 0; b < NB; ++b) {
            save(b < 8 ? a.save_dst[b] : a.live, b);
            hash(b, b & 7u);
#pragma unroll
            for (int c = 0; c < 6; ++c) w[c] += 1u;
        }
    }
    if (S >= 3) {
        __syncthreads();
        for (uint32_t i = tid; i < a.n_saves * 3u; i += 256u) a.parts[(uint64_t)i * a.part_stride + tile] = s_acc[i];
    } else if (ALU && acc == 0x1234567ull) *reinterpret_cast<uint64_t*>(a.live) = acc;
}
__global__ void k_small(uint64_t* p) { if (threadIdx.x == 0 && p[0] == 0x1234567ull) p[1] = 1; }
// the library's way of timing (ProfScope): one event pair per launch, a small dependent kernel between the launches
template <int S, int MASK, int ALU> float run2_bracketed(const Args2& a, int reps, hipStream_t st) {
    const uint32_t grid = (uint32_t)((a.n_slots + 255) / 256);
    std::vector<hipEvent_t> ev(2 * reps); for (auto& e : ev) CK(hipEventCreate(&e));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tick2<S, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(ev[2 * i], st));
        hipLaunchKernelGGL((tick2<S, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
        CK(hipEventRecord(ev[2 * i + 1], st));
        hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, st, a.parts);
    }
    CK(hipStreamSynchronize(st));
    double sum = 0; for (int i = 0; i < reps; ++i) { float ms = 0; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); sum += ms; }
    for (auto& e : ev) CK(hipEventDestroy(e));
    return (float)(sum * 1000.0 / reps);
}
template <int S, int MASK, int ALU> float run2(const Args2& a, int reps, hipStream_t st) {
    const uint32_t grid = (uint32_t)((a.n_slots + 255) / 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tick2<S, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tick2<S, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

template <int V, int MASK, int ALU> float run(const Args& a, int reps, hipStream_t st) {
    const uint32_t grid = (uint32_t)((a.n_slots + 256ull * V - 1) / (256ull * V));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((tick<V, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tick<V, MASK, ALU>), dim3(grid), dim3(256), 0, st, a);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000ull;
    const uint64_t cap = (n + 8191) / 8192 * 8192;
    // engine layout: header 4 KiB + 4 masks, then tile-major columns: 15 words per slot (60 B), hot words 0,1,2,10,11,12,13 as in the particles world
    const uint64_t mask_bytes = (cap / 8 + 4095) / 4096 * 4096;
    const uint64_t cols_base = 4096 + 4 * mask_bytes;
    const uint64_t ts = 8192ull * 60;
    const uint64_t block = cols_base + (cap / 8192) * ts;
    uint8_t* mem = nullptr; CK(hipMalloc((void**)&mem, block * (NB + 1))); CK(hipMemset(mem, 1, block * (NB + 1)));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int layout = 0; layout < 2; ++layout) {
        Args a; memset(&a, 0, sizeof a);
        a.src = mem; for (int b = 0; b < NB; ++b) a.dst[b] = mem + (uint64_t)(b + 1) * block;
        a.n_slots = n; a.ts = ts; a.layout = layout;
        const int word_of[7] = {0, 1, 2, 10, 11, 12, 13};
        for (int c = 0; c < 7; ++c) a.col_off[c] = cols_base + (uint64_t)word_of[c] * 8192 * 4;
        if (layout == 1) for (int c = 0; c < 7; ++c) a.col_off[c] = 0;
        for (int m = 0; m < 4; ++m) a.mask_off[m] = 4096 + m * mask_bytes;
        if (layout == 1) { a.src += cols_base; for (int b = 0; b < NB; ++b) a.dst[b] += cols_base; for (int m = 0; m < 4; ++m) a.mask_off[m] -= cols_base; }
        const double mb = n * 32.0 * (NB + 1) / 1e6;
        printf("layout %d (%s), %llu slots, %.0f MB per launch\n", layout, layout ? "workgroup-contiguous 8 KiB pieces" : "engine: 8192-slot layout tiles", (unsigned long long)n, mb);
#define R(V, M, A) { const float us = run<V, M, A>(a, 100, st); printf("  V=%d mask=%d alu=%d : %7.2f us  %5.2f TB/s\n", V, M, A, us, mb / us); }
        R(1, 0, 0) R(1, 1, 0) R(1, 2, 0) R(1, 1, 8) R(1, 2, 8)
        R(2, 0, 0) R(2, 1, 0) R(2, 2, 0) R(2, 1, 8) R(2, 2, 8)
        R(4, 0, 0) R(4, 1, 0) R(4, 2, 0) R(4, 1, 8) R(4, 2, 8)
        if (layout == 0) {
            Args2 b; memset(&b, 0, sizeof b);
            b.src = a.src; for (int k = 0; k < 8; ++k) { b.save_dst[k] = a.dst[k]; b.save_rows[k] = 0x3c07ull; b.dt_bits[k] = 0x3c888889u; }
            b.live = a.dst[8]; b.n_saves = 8; b.n_ops = 16; b.op_bits = 0xAAAAull;              // Save, Advance, Save, Advance, ...
            b.n_slots = n; b.ts = ts; for (int c = 0; c < 7; ++c) b.col_off[c] = a.col_off[c]; for (int m = 0; m < 4; ++m) b.mask_off[m] = a.mask_off[m];
            b.part_stride = (uint32_t)((n + 255) / 256); CK(hipMalloc((void**)&b.parts, (size_t)b.part_stride * 24 * 8));
#define R2(S, M, A) { const float us = run2<S, M, A>(b, 100, st); printf("  skeleton S=%d mask=%d alu=%d : %7.2f us  %5.2f TB/s\n", S, M, A, us, mb / us); }
            { const float us = run2_bracketed<3, 1, 8>(b, 100, st); printf("  skeleton S=3 mask=1 alu=8, one event pair per launch + a small kernel between launches : %7.2f us\n", us); }
            { const float us = run2_bracketed<3, 0, 8>(b, 100, st); printf("  skeleton S=3 mask=0 alu=8, one event pair per launch + a small kernel between launches : %7.2f us\n", us); }
            R2(1, 0, 0) R2(1, 1, 0) R2(1, 1, 8) R2(2, 0, 0) R2(2, 1, 0) R2(2, 1, 8) R2(3, 0, 8) R2(3, 1, 8)
        }
    }
    return 0;
}
