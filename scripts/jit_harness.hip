// jit_harness.hip -- time a generated request-group kernel (the text `ggrs_hip_generated_kernel_source` returns for the headline
// particles world of 1 M entities) outside the library: compile it with hiprtc, build the argument block of a steady-state SyncTest
// tick by hand (Load from a ring slot, 8 x (Advance, Save), live written once; row masks 0x3c07 = the 7 words the systems write),
// launch it the way the library does (one event pair per launch, a small dependent kernel in between) and print µs per launch.
// Purpose: edit the kernel TEXT and see what each piece costs (profiles/r03n) without touching the generator.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/jit_harness.hip -lhiprtc -o scripts/jit_harness
// Run:   scripts/jit_harness kernel.hip [more.hip ...]        (sources generated for capacity 1 000 000, max_depth 9)
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct GgrsJitArgs {                                   // kernel_gen.hpp GGRS_COMPONENT_TEXT: must stay in step with it
    const unsigned char* src; unsigned char* live;
    unsigned char* save_dst[16]; int save_frame[16];
    unsigned long long save_rows[16]; unsigned long long live_rows, load_rows;
    unsigned save_pmask[16]; unsigned live_pmask, pad1;
    unsigned dt_bits[24]; unsigned aux_bits[24];
    unsigned char inputs[24][16]; unsigned char n_inputs[24];
    int step_frame[24]; int step_confirmed[24]; unsigned char step_flags[24];
    unsigned long long op_bits; unsigned n_ops, n_saves, n_steps, src_is_live, skip_live, dp_s;
    unsigned long long len;
    unsigned long long* parts; unsigned part_stride, nt;
    unsigned n_units, cached_saves;
    unsigned long long* fold_wg_parts; unsigned* fold_ticket; unsigned long long* fold_out;
};
static_assert(sizeof(GgrsJitArgs) == 1328, "argument block");

__global__ void k_small(unsigned long long* p) { if (threadIdx.x == 0 && p[0] == 0x1234567ull) p[1] = 1; }
__global__ void k_fill64(unsigned long long* p, size_t n, unsigned long long v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }

int main(int argc, char** argv) {
    const unsigned long long n = 1000000ull, cap_pad = 1007616ull;          // 123 layout tiles of 8192 slots
    const unsigned long long off_cols = 507904ull, ts = 491520ull;
    const unsigned long long state = (off_cols + (cap_pad / 8192) * ts + 4095) / 4096 * 4096;
    unsigned char* mem = nullptr; CK(hipMalloc((void**)&mem, state * 10)); CK(hipMemset(mem, 0, state * 10));
    hipStream_t st; CK(hipStreamCreate(&st));
    // source block: everything alive and present (mask words all ones), Ttl far from zero (8-byte words at row offset 704512)
    for (int b = 0; b < 10; ++b) { unsigned char* m = mem + b * state; CK(hipMemset(m + 256, 0xFF, cap_pad / 8)); CK(hipMemset(m + 126208, 0xFF, cap_pad / 8)); CK(hipMemset(m + 252160, 0xFF, cap_pad / 8)); CK(hipMemset(m + 378112, 0xFF, cap_pad / 8)); }   // every block: presence masks are not re-stored when the destination holds them
    for (unsigned long long t = 0; t < cap_pad / 8192; ++t) hipLaunchKernelGGL(k_fill64, dim3(64), dim3(256), 0, st, (unsigned long long*)(mem + 704512ull + t * ts), (size_t)8192, 1ull << 40);
    if (getenv("RANDOM_DATA")) {                                              // realistic values instead of zeros: floats in [-200, 200) in the 6 f32 rows the systems write
        std::vector<float> h(8192); unsigned long long x = 88172645463325252ull;
        for (unsigned long long t = 0; t < cap_pad / 8192; ++t) for (int c = 0; c < 6; ++c) {
            for (auto& f : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; f = (float)((double)(x >> 11) / 9007199254740992.0 * 400.0 - 200.0); }
            const unsigned long long row = (c < 3 ? 507904ull + c * 32768ull : 606208ull + (c - 3) * 32768ull);
            CK(hipMemcpy(mem + row + t * ts, h.data(), 32768, hipMemcpyHostToDevice));
        }
    }
    CK(hipStreamSynchronize(st));
    const unsigned n_units = (unsigned)((n + 63) / 64), tiles = (n_units + 3) / 4, grid = 8 * ((tiles + 7) / 8);
    unsigned long long* parts = nullptr; CK(hipMalloc((void**)&parts, (size_t)grid * 16 * 3 * 8 + 4096)); CK(hipMemset(parts, 0, (size_t)grid * 16 * 3 * 8 + 4096));
    GgrsJitArgs a; memset(&a, 0, sizeof a);
    a.src = mem; a.live = mem + 9 * state;
    for (int k = 0; k < 8; ++k) { a.save_dst[k] = mem + (1 + k) * state; a.save_frame[k] = 100 + k; a.save_rows[k] = 0x3c07ull; }
    a.live_rows = 0x3c07ull; a.load_rows = 0x3c07ull;
    for (int k = 0; k < 24; ++k) a.dt_bits[k] = 0x3c888889u;
    a.op_bits = 0x5555ull; a.n_ops = 16; a.n_saves = 8; a.n_steps = 8;      // Advance, Save, Advance, Save, ... (bit set = Advance)
    if (getenv("CACHED")) a.cached_saves = (unsigned)atoi(getenv("CACHED"));
    if (getenv("PMASK")) { for (int k = 0; k < 8; ++k) a.save_pmask[k] = (unsigned)atoi(getenv("PMASK")); a.live_pmask = (unsigned)atoi(getenv("PMASK")); }
    if (getenv("NT")) a.nt = (unsigned)atoi(getenv("NT"));
    if (getenv("NOADV")) { a.op_bits = 0; a.n_ops = 8; a.n_steps = 0; }      // 8 Saves of unchanged data (Load + Saves only): separates "the ring rotates" from "the data changes"
    a.len = n; a.parts = parts; a.part_stride = grid; a.nt = 1; a.n_units = n_units;

    // every source is compiled once; then ROUNDS passes over all of them (A B C A B C ...), 40 bracketed launches each: the figure
    // printed is the MEDIAN of a variant's rounds, so box drift and clock ramps hit every variant alike
    struct Var { std::string name; hipModule_t mod; hipFunction_t fn; std::vector<double> us; };
    std::vector<Var> vars;
    for (int f = 1; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb"); if (!fp) { printf("%s: cannot open\n", argv[f]); continue; }
        std::string src; char buf[65536]; size_t got; while ((got = fread(buf, 1, sizeof buf, fp)) > 0) src.append(buf, got); fclose(fp);
        hiprtcProgram prog; hiprtcCreateProgram(&prog, src.c_str(), "k.hip", 0, nullptr, nullptr);
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"};
        if (hiprtcCompileProgram(prog, 6, opts) != HIPRTC_SUCCESS) {
            size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); printf("%s: compile failed\n%s\n", argv[f], log.c_str()); continue;
        }
        size_t cs = 0; hiprtcGetCodeSize(prog, &cs); std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
        Var v; v.name = argv[f]; CK(hipModuleLoadData(&v.mod, code.data())); CK(hipModuleGetFunction(&v.fn, v.mod, "ggrs_jit_tick"));
        vars.push_back(v);
    }
    const int ROUNDS = getenv("ROUNDS") ? atoi(getenv("ROUNDS")) : 5, reps = 40;
    const unsigned lds = getenv("LDS") ? (unsigned)atoi(getenv("LDS")) : 8192u;      // dynamic LDS: the per-lane checksum rows (8 Saves x 2 components x 512 B)
    std::vector<hipEvent_t> ev(2 * reps); for (auto& e : ev) CK(hipEventCreate(&e));
    void* params[] = {&a};
    const bool rotate = getenv("ROTATE") != nullptr; unsigned long long tick_no = 0;
    for (int r = 0; r < ROUNDS; ++r) for (auto& v : vars) {
        for (int i = 0; i < 5; ++i) CK(hipModuleLaunchKernel(v.fn, grid, 1, 1, 256, 1, 1, lds, st, params, nullptr));
        for (int i = 0; i < reps; ++i) {
            if (rotate) {                                                     // the ring as the library walks it: this tick's source is the oldest snapshot, the 8 Saves overwrite the other 8 slots
                tick_no++;
                a.src = mem + (tick_no % 9) * state;
                for (int k = 0; k < 8; ++k) a.save_dst[k] = mem + ((tick_no + 1 + k) % 9) * state;
            }
            CK(hipEventRecord(ev[2 * i], st));
            CK(hipModuleLaunchKernel(v.fn, grid, 1, 1, 256, 1, 1, lds, st, params, nullptr));
            CK(hipEventRecord(ev[2 * i + 1], st));
            hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, st, parts);
        }
        CK(hipStreamSynchronize(st));
        double sum = 0; for (int i = 0; i < reps; ++i) { float ms = 0; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); sum += ms; }
        v.us.push_back(sum * 1000.0 / reps);
    }
    for (auto& v : vars) {
        CK(hipMemset(parts, 0, (size_t)grid * 16 * 3 * 8)); CK(hipModuleLaunchKernel(v.fn, grid, 1, 1, 256, 1, 1, lds, st, params, nullptr)); CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hp((size_t)grid * 24);
        CK(hipMemcpy(hp.data(), parts, hp.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long x0 = 0, x1 = 0, cnt = 0, y0 = 0;                  // Save 0: XOR of both components' partials, live count; Save 7 component 0
        for (unsigned t = 0; t < grid; ++t) { x0 ^= hp[(size_t)0 * grid + t]; x1 ^= hp[(size_t)1 * grid + t]; cnt += hp[(size_t)2 * grid + t]; y0 ^= hp[(size_t)21 * grid + t]; }
        std::vector<double> u = v.us; std::sort(u.begin(), u.end());
        printf("%-30s median %6.2f  min %6.2f  max %6.2f us   Save0 %016llx %016llx Save7 %016llx\n", v.name.c_str(), u[u.size() / 2], u.front(), u.back(), x0, x1, y0);
    }
    return 0;
}
