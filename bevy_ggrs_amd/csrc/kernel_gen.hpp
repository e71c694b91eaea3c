// kernel_gen.hpp -- run-time kernel generation for libggrs_hip.so (host code; included by ggrs_hip.hip inside its anonymous
// namespace, after `struct ggrs_world`).  Two users:
//   * ggrs_hip_add_custom_system: a user's per-entity GgrsSchedule system, compiled as its own one-launch-per-request kernel;
//   * seal(): the request-group kernel WRITTEN FOR THE WORLD (jit_source) -- one slot per lane, every registered word in a
//     register, built-in and user systems inlined in registration order, checksum specs unrolled.
// Both go through hiprtc (dlopen'ed: no link-time dependency) with the floating-point contract of csrc/Makefile.
#pragma once

// ---- GGRS_SYS_CUSTOM: user-written per-entity systems, compiled with hiprtc for gfx950 ----------------------------
// The argument block of the generated kernel.  The SAME text is compiled on the host (below) and pasted into the
// generated device source, and the device source static_asserts the host's sizeof: the two cannot drift.
#define GGRS_CUSTOM_ABI_TEXT \
    "typedef unsigned long long ggrs_u64; typedef unsigned int ggrs_u32;\n" \
    "struct GgrsFrame { float dt; int frame; ggrs_u32 n_inputs; unsigned char input[16]; float fparam[4]; long long iparam[2]; };\n" \
    "struct GgrsCustomArgs {\n" \
    "    unsigned char* state;\n" \
    "    ggrs_u64 off_alive, off_disabled, off_dframe, len_pad64;\n" \
    "    ggrs_u64 off_present[8], col_off[8];\n" \
    "    ggrs_u32 ts[8];\n" \
    "    int defer, pad;\n" \
    "    GgrsFrame fr;\n" \
    "};\n"
typedef unsigned long long ggrs_u64; typedef unsigned int ggrs_u32;
struct GgrsFrame { float dt; int frame; ggrs_u32 n_inputs; unsigned char input[16]; float fparam[4]; long long iparam[2]; };
struct GgrsCustomArgs {
    unsigned char* state;
    ggrs_u64 off_alive, off_disabled, off_dframe, len_pad64;
    ggrs_u64 off_present[8], col_off[8];
    ggrs_u32 ts[8];
    int defer, pad;
    GgrsFrame fr;
};
static_assert(GGRS_CUSTOM_MAX_BINDINGS == 8, "GgrsCustomArgs is sized for 8 bindings");

struct Hiprtc {
    void* lib = nullptr; bool tried = false; std::string why;
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    decltype(&hiprtcGetErrorString) err_str = nullptr;
};
void hiprtc_load(Hiprtc& r);
Hiprtc& hiprtc() {
    static Hiprtc r;
    static std::once_flag once;
    std::call_once(once, [] { hiprtc_load(r); });          // worlds may be created from several threads
    return r;
}
void hiprtc_load(Hiprtc& r) {
    r.tried = true;
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) { r.why = "libhiprtc.so not found (custom systems and generated kernels need the ROCm runtime compiler)"; return; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) { ok = false; r.why = std::string("libhiprtc lacks ") + n; } return p; };
    r.create = (decltype(r.create))sym("hiprtcCreateProgram");
    r.compile = (decltype(r.compile))sym("hiprtcCompileProgram");
    r.log_size = (decltype(r.log_size))sym("hiprtcGetProgramLogSize");
    r.log = (decltype(r.log))sym("hiprtcGetProgramLog");
    r.code_size = (decltype(r.code_size))sym("hiprtcGetCodeSize");
    r.code = (decltype(r.code))sym("hiprtcGetCode");
    r.destroy = (decltype(r.destroy))sym("hiprtcDestroyProgram");
    r.err_str = (decltype(r.err_str))sym("hiprtcGetErrorString");
    if (!ok) r.lib = nullptr;
}

// The entity view a custom system sees (include/ggrs_hip.h, ggrs_hip_add_custom_system) -- one text for the per-request
// kernel of a custom system and for the generated request-group kernel.
#define GGRS_ENTITY_TEXT \
    "struct GgrsEntity {\n" \
    "    ggrs_u64 slot; ggrs_u64 w[8]; int kill;\n" \
    "    __device__ float& f32(int i) { return *reinterpret_cast<float*>(&w[i]); }\n" \
    "    __device__ ggrs_u32& u32(int i) { return *reinterpret_cast<ggrs_u32*>(&w[i]); }\n" \
    "    __device__ int& i32(int i) { return *reinterpret_cast<int*>(&w[i]); }\n" \
    "    __device__ ggrs_u64& u64(int i) { return w[i]; }\n" \
    "    __device__ void despawn() { if (kill == 0) kill = 1; }\n" \
    "    __device__ void despawn_rollback() { kill = 2; }\n" \
    "};\n"

// hiprtc: source -> code object -> module + kernel handle.  A compile error fails with the compiler log in w->err.
int hiprtc_build(ggrs_world* w, const std::string& src, const char* what, const char* kernel, hipModule_t* mod, hipFunction_t* fn) {
    Hiprtc& rtc = hiprtc();
    if (!rtc.lib) return w->fail(GGRS_E_HIP, "%s: %s", what, rtc.why.c_str());
    hiprtcProgram prog = nullptr;
    hiprtcResult r = rtc.create(&prog, src.c_str(), "ggrs_generated.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) return w->fail(GGRS_E_HIP, "hiprtcCreateProgram: %s", rtc.err_str(r));
    // the same floating-point contract as the statically compiled kernels (csrc/Makefile): no contraction, no fast-math
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"};
    r = rtc.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    if (r != HIPRTC_SUCCESS) {
        size_t n = 0; std::string log;
        if (rtc.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); (void)rtc.log(prog, &log[0]); }
        (void)rtc.destroy(&prog);
        if (log.size() > 3000) log.resize(3000);
        w->err = std::string(what) + " does not compile (" + rtc.err_str(r) + "):\n" + log;   // the whole log, not fail()'s 512 bytes
        return GGRS_E_INVALID;
    }
    size_t nbytes = 0;
    std::vector<char> image;
    if (rtc.code_size(prog, &nbytes) == HIPRTC_SUCCESS && nbytes) { image.resize(nbytes); r = rtc.code(prog, image.data()); } else r = HIPRTC_ERROR_INTERNAL_ERROR;
    (void)rtc.destroy(&prog);
    if (r != HIPRTC_SUCCESS) return w->fail(GGRS_E_HIP, "hiprtcGetCode: %s", rtc.err_str(r));
    if (!mod) return GGRS_OK;                                      // compile check only (GGRS_WORLD_LAYOUT_ONLY)
    HIPCHK(w, hipModuleLoadData(mod, image.data()));
    if (hipModuleGetFunction(fn, *mod, kernel) != hipSuccess) { (void)hipModuleUnload(*mod); *mod = nullptr; return w->fail(GGRS_E_HIP, "%s: kernel symbol missing from the compiled module", what); }
    return GGRS_OK;
}

// ---- the generated request-group kernel ("ggrs_jit_tick") ---------------------------------------------------------------
// k_tick / k_tick3 are hand-specialised to the particles world; every other world used to fall to k_tick_gen, which keeps
// the state in LDS and INTERPRETS the registered systems and checksum specs (0.45 of the HBM roofline at 1 M entities,
// 45 us per 100 k tick), and a world with a user-written system had no fused path at all.  At seal the library now writes
// the fused kernel FOR THIS WORLD -- one slot per lane, every registered word of the slot in a named register, the systems
// (built-in kinds and the user's sources alike) inlined in registration order, every checksum spec unrolled -- and compiles
// it with hiprtc.  Same request-group protocol, same per-wave partials + k_gen_finalize, same depth-parallel roles as
// k_tick1.  The SeaHash / box_game arithmetic is device_prelude.hpp, the text the static kernels are compiled from.
static const char kJitPrelude[] =
#define GGRS_SHARED_CODE(...) #__VA_ARGS__
#include "device_prelude.hpp"
#undef GGRS_SHARED_CODE
    ;
#define GGRS_JIT_ABI_TEXT \
    "struct GgrsJitArgs {\n" \
    "    const unsigned char* src; unsigned char* live;\n" \
    "    unsigned char* save_dst[16]; int save_frame[16];\n" \
    "    ggrs_u32 dt_bits[24]; ggrs_u32 aux_bits[24];\n" \
    "    unsigned char inputs[24][16]; unsigned char n_inputs[24];\n" \
    "    int step_frame[24]; int step_confirmed[24]; unsigned char step_flags[24];\n" \
    "    ggrs_u64 op_bits; ggrs_u32 n_ops, n_saves, n_steps, src_is_live, skip_live, dp_s;\n" \
    "    ggrs_u64 len;\n" \
    "    ggrs_u64* parts; ggrs_u32 part_stride, nt;\n" \
    "};\n"
struct GgrsJitArgs {
    const unsigned char* src; unsigned char* live;
    unsigned char* save_dst[16]; int save_frame[16];
    ggrs_u32 dt_bits[24]; ggrs_u32 aux_bits[24];
    unsigned char inputs[24][16]; unsigned char n_inputs[24];
    int step_frame[24]; int step_confirmed[24]; unsigned char step_flags[24];
    ggrs_u64 op_bits; ggrs_u32 n_ops, n_saves, n_steps, src_is_live, skip_live, dp_s;
    ggrs_u64 len;
    ggrs_u64* parts; ggrs_u32 part_stride, nt;       // nt: snapshot stores are non-temporal (big worlds: written once, read a tick later)
};
static_assert(MAX_TICK_SAVES == 16 && MAX_TICK_STEPS == 24, "GgrsJitArgs is sized for 16 Saves / 24 steps per group");
constexpr uint32_t JIT_MAX_UNITS = 64;       // 4-byte register units per slot the generated kernel may hold
// The 4-slots-per-lane form (16-byte accesses, 1024-slot workgroups) is generated and parity-tested but NOT used by default:
// with saddr addressing the 1-slot form runs at 8 waves per SIMD and beats it at every size (profiles/r02jit/ab_v4.txt:
// 1 M 144 vs 203 us per depth-8 tick, 300 k 41 vs 53); GGRS_JIT_V=4 selects it for A/B.

void sfmt(std::string& s, const char* fmt, ...) {
    char buf[4096];
    va_list ap; va_start(ap, fmt); const int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n < 0 || (size_t)n >= sizeof buf) { s += "\n#error generator line too long\n"; return; }   // never silently truncate generated code
    s += buf;
}
std::string f32_lit(float f) { uint32_t b; memcpy(&b, &f, 4); char buf[48]; snprintf(buf, sizeof buf, "__uint_as_float(0x%08xu)", b); return buf; }

// Writes the kernel for this world, V slots per lane: V = 1 (4-byte accesses, 256-slot workgroups: the shortest chain, for
// worlds that live in L2 / the Infinity Cache) or V = 4 (four consecutive slots per lane, 16-byte accesses, 1024-slot
// workgroups: HBM-sized worlds).  Returns false when the world is outside what the generator covers (the caller falls back
// to k_tick_gen or to the per-request path): a system that touches a live-only component other than BOX_MOVE's read-only
// Player.handle, too many words for the register file.
bool jit_source(const ggrs_world* w, std::string& s, int V = 1) {
    const uint32_t nc = (uint32_t)w->comps.size();
    uint32_t units = 0;
    for (auto& c : w->comps) if (!c.no_rollback) units += c.n_words * (c.word_bytes / 4);
    if (units == 0 || units > JIT_MAX_UNITS || (V == 4 && units > JIT_MAX_UNITS / 2)) return false;
    auto rb = [&](uint32_t c) { return c < nc && !w->comps[c].no_rollback; };
    auto col = [&](uint32_t c, uint32_t k) { return w->comps[c].col_base + k; };
    bool marks = false;
    for (auto& d : w->systems) {
        switch (d.kind) {
        case GGRS_SYS_PARTICLES_SPAWN: break;
        case GGRS_SYS_PARTICLES_UPDATE: if (!rb(d.comp[0]) || !rb(d.comp[1])) return false; break;
        case GGRS_SYS_TTL_DESPAWN: case GGRS_SYS_ADD_U32: if (!rb(d.comp[0])) return false; break;
        case GGRS_SYS_SAT_SUB_DESPAWN: if (!rb(d.comp[0])) return false; marks |= d.iparam[1] == GGRS_DESPAWN_ROLLBACK; break;
        case GGRS_SYS_BOX_MOVE: if (!rb(d.comp[0]) || !rb(d.comp[1]) || d.comp[2] >= nc) return false; break;
        case GGRS_SYS_CUSTOM: {
            const ggrs_world::Custom& c = w->customs[d.comp[0]];
            for (uint32_t i = 0; i < c.n_bind; ++i) if (!rb(c.comp[i])) return false;      // may WRITE a live-only word: not replayable
            marks = true;                                                                  // may call despawn_rollback()
        } break;
        default: return false;
        }
    }
    std::vector<uint32_t> cks_comp;                                  // checksummed components in id order (== w->cks_comp once sealed)
    for (uint32_t c = 0; c < nc; ++c) if (w->comps[c].checksummed) { if (!rb(c)) return false; cks_comp.push_back(c); }
    const uint32_t n_cks = (uint32_t)cks_comp.size();
    const unsigned long long OFF_ALIVE = w->off_alive, OFF_DIS = w->marks.off_disabled, OFF_DF = w->marks.off_dframe;
    const int SLOTS = 256 * V;                                       // slots per workgroup

    s.clear();
    s += "typedef unsigned long uint64_t; typedef unsigned int uint32_t; typedef unsigned short uint16_t; typedef unsigned char uint8_t;\n"
         "typedef long int64_t; typedef int int32_t;\n"
         "typedef unsigned long long ggrs_u64; typedef unsigned int ggrs_u32;\n"
         "typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));\n"
         "typedef int i32x4 __attribute__((ext_vector_type(4)));\n"
         "#define GGRS_G __attribute__((address_space(1)))\n"
         "// a wave-uniform pointer pinned into an SGPR pair: `sgpr_base(p) + lane_offset_u32` selects the saddr form of\n"
         "// global_load / global_store (no 64-bit VALU address arithmetic, no 64-bit address registers per word)\n"
         "__device__ __forceinline__ GGRS_G unsigned char* sgpr_base(const unsigned char* p) { unsigned long x = (unsigned long)p; asm volatile(\"\" : \"+s\"(x)); return (GGRS_G unsigned char*)x; }\n"
         "namespace ggrs {\n";
    s += kJitPrelude;
    s += "\n}\nusing namespace ggrs;\n";
    s += "struct GgrsFrame { float dt; int frame; ggrs_u32 n_inputs; unsigned char input[16]; float fparam[4]; long long iparam[2]; };\n";
    s += GGRS_ENTITY_TEXT;
    s += GGRS_JIT_ABI_TEXT;
    sfmt(s, "static_assert(sizeof(GgrsJitArgs) == %zu, \"host/device argument block mismatch\");\n", sizeof(GgrsJitArgs));
    for (size_t i = 0; i < w->customs.size(); ++i) {
        std::string nm = w->customs[i].name;
        for (char& ch : nm) if (!isalnum((unsigned char)ch) && ch != '_') ch = '_';
        sfmt(s, "namespace ggrs_sys_%zu {\n#line 1 \"%s\"\n", i, nm.c_str());
        s += w->customs[i].source;
        s += "\n}\n";
    }
    s += "#line 1 \"ggrs_jit_tick\"\n";
    sfmt(s, "extern \"C\" __global__ __launch_bounds__(256%s) void ggrs_jit_tick(GgrsJitArgs a) {\n"
            "    const uint32_t t = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;\n"
            "    const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;\n"
            "    const uint32_t o_first = a.dp_s ? blockIdx.y * a.dp_s : 0u;          // depth-parallel roles, as in k_tick1\n"
            "    const uint32_t o_last = a.dp_s ? min(o_first + a.dp_s, a.n_saves + 1u) : a.n_saves + 1u;\n"
            "    const bool my_live = o_last == a.n_saves + 1u;\n"
            "    if (a.dp_s && o_first == a.n_saves && !writes_live) return;\n"
            "    // per-workgroup checksum partials [Save][component .. live count]: the waves fold into LDS, ONE row per workgroup goes\n"
            "    // to memory at the end (k_gen_finalize then reads a quarter of what per-wave rows would be)\n"
            "    __shared__ ggrs_u64 s_acc[16 * %u];\n"
            "    for (uint32_t i = tid; i < 16u * %uu; i += 256u) s_acc[i] = 0;\n"
            "    __syncthreads();\n"
            "    const uint64_t e0 = (uint64_t)t * %du + tid * %du;                  // this lane's first slot (of %d)\n"
            "    const bool in_len = (uint64_t)t * %du < a.len;                      // workgroup-uniform\n"
            "    // word c of slot e lives at col_off[c] + (e >> 13) * tile_stride + (e & 8191) * word_bytes: the layout tile is the\n"
            "    // workgroup's (uniform: SGPRs), the lane contributes one 32-bit offset per word size -> saddr-form accesses\n"
            "    const uint64_t tbase = (uint64_t)(t >> %d) * %uull;\n"
            "    const uint32_t ei = (t & %uu) * %du + tid * %du, lo4 = ei * 4u, lo8 = ei * 8u;\n",
         V == 4 ? ", 4" : "", n_cks + 1, n_cks + 1, SLOTS, V, V, SLOTS, LT_SHIFT - (V == 4 ? 10 : 8), w->ts, (unsigned)(LAYOUT_TILE / SLOTS - 1), SLOTS, V);
    if (V == 1) s += "    const uint64_t wi8 = ((uint64_t)t * 4u + wave) * 8u;                // this wave's mask word: bit `lane` is this slot\n"
                     "    const uint32_t sh = lane;\n";
    else s += "    const uint64_t wi8 = ((uint64_t)t * 16u + wave * 4u + (lane >> 4)) * 8u; // the mask word of this lane's 4 slots: bits sh .. sh+3\n"
              "    const uint32_t sh = (lane & 15u) * 4u;\n";
    const char* mask_writer = V == 1 ? "lane == 0" : "(lane & 15u) == 0";
    // a 64-bit mask word from one bit per slot of every lane that shares it
    auto emit_word_from_bits = [&](const char* name, const char* bit, const char* indent) {
        if (V == 1) { sfmt(s, "%sconst uint64_t %s = __ballot(%s_0);\n", indent, name, bit); return; }
        sfmt(s, "%suint64_t %s = (uint64_t)((%s_0 ? 1u : 0u) | (%s_1 ? 2u : 0u) | (%s_2 ? 4u : 0u) | (%s_3 ? 8u : 0u)) << sh;\n"
                "%s%s |= __shfl_xor(%s, 1, 64); %s |= __shfl_xor(%s, 2, 64); %s |= __shfl_xor(%s, 4, 64); %s |= __shfl_xor(%s, 8, 64);\n",
             indent, name, bit, bit, bit, bit, indent, name, name, name, name, name, name, name, name);
    };
    // ---- masks and words of the lane's slots
    sfmt(s, "    const uint64_t mk_alive = *reinterpret_cast<const uint64_t*>(a.src + %lluull + wi8);\n", OFF_ALIVE);
    for (int j = 0; j < V; ++j) sfmt(s, "    bool alive_%d = (mk_alive >> (sh + %du)) & 1ull;\n", j, j);
    for (uint32_t c = 0; c < nc; ++c) if (rb(c)) {
        sfmt(s, "    const uint64_t mk%u = *reinterpret_cast<const uint64_t*>(a.src + %lluull + wi8);\n", c, (unsigned long long)w->off_present[c]);
        for (int j = 0; j < V; ++j) sfmt(s, "    const bool p%u_%d = (mk%u >> (sh + %du)) & 1ull;\n", c, j, c, j);
    }
    auto wtype = [&](uint32_t c) { return w->comps[c].word_bytes == 8 ? "uint64_t" : "uint32_t"; };
    for (uint32_t c = 0; c < nc; ++c) if (rb(c)) for (uint32_t k = 0; k < w->comps[c].n_words; ++k) {
        const uint32_t cl = col(c, k), wb = w->comps[c].word_bytes;
        if (w->col_ts[cl] != w->ts) return false;                    // every rollback column shares the tile stride
        sfmt(s, "#define o%u(blk) (sgpr_base((blk) + (%lluull + tbase)) + lo%u)\n   ", cl, (unsigned long long)w->col_off[cl], wb);
        for (int j = 0; j < V; ++j) sfmt(s, " %s w%u_%d = 0;", wtype(c), cl, j);
        s += "\n";
    }
    // loads / stores of all words of the lane's slots from / to a block
    auto emit_load = [&](const char* blk, const char* indent) {
        for (uint32_t c = 0; c < nc; ++c) if (rb(c)) for (uint32_t k = 0; k < w->comps[c].n_words; ++k) {
            const uint32_t cl = col(c, k); const bool w8 = w->comps[c].word_bytes == 8;
            if (V == 1) sfmt(s, "%sw%u_0 = *(const GGRS_G %s*)o%u(%s);\n", indent, cl, wtype(c), cl, blk);
            else if (!w8) sfmt(s, "%s{ const u32x4 v = *(const GGRS_G u32x4*)o%u(%s); w%u_0 = v.x; w%u_1 = v.y; w%u_2 = v.z; w%u_3 = v.w; }\n", indent, cl, blk, cl, cl, cl, cl);
            else sfmt(s, "%s{ const u64x2 v = *(const GGRS_G u64x2*)o%u(%s); const u64x2 u = *(const GGRS_G u64x2*)(o%u(%s) + 16u); w%u_0 = v.x; w%u_1 = v.y; w%u_2 = u.x; w%u_3 = u.y; }\n",
                      indent, cl, blk, cl, blk, cl, cl, cl, cl);
        }
    };
    auto emit_words_out = [&](const char* dst, const char* indent, bool nt) {
        for (uint32_t c = 0; c < nc; ++c) if (rb(c)) for (uint32_t k = 0; k < w->comps[c].n_words; ++k) {
            const uint32_t cl = col(c, k); const bool w8 = w->comps[c].word_bytes == 8;
            char val[2][96], ty[16];
            if (V == 1) { snprintf(val[0], sizeof val[0], "w%u_0", cl); snprintf(ty, sizeof ty, "%s", wtype(c)); }
            else if (!w8) { snprintf(val[0], sizeof val[0], "u32x4{w%u_0, w%u_1, w%u_2, w%u_3}", cl, cl, cl, cl); snprintf(ty, sizeof ty, "u32x4"); }
            else { snprintf(val[0], sizeof val[0], "u64x2{w%u_0, w%u_1}", cl, cl); snprintf(val[1], sizeof val[1], "u64x2{w%u_2, w%u_3}", cl, cl); snprintf(ty, sizeof ty, "u64x2"); }
            const int pieces = (V == 4 && w8) ? 2 : 1;
            for (int q = 0; q < pieces; ++q) {
                if (nt) sfmt(s, "%s__builtin_nontemporal_store(%s, (GGRS_G %s*)(o%u(%s) + %du));\n", indent, val[q], ty, cl, dst, q * 16);
                else sfmt(s, "%s*(GGRS_G %s*)(o%u(%s) + %du) = %s;\n", indent, ty, cl, dst, q * 16, val[q]);
            }
        }
    };
    auto emit_store = [&](const char* dst, const char* alive_word, const char* indent, bool nt_variant) {
        std::string in2 = std::string(indent) + "    ", in3 = in2 + "    ";
        sfmt(s, "%sif (in_len) {\n", indent);
        if (nt_variant) {
            sfmt(s, "%sif (a.nt) {\n", in2.c_str());
            emit_words_out(dst, in3.c_str(), true);
            sfmt(s, "%s} else {\n", in2.c_str());
            emit_words_out(dst, in3.c_str(), false);
            sfmt(s, "%s}\n", in2.c_str());
        } else emit_words_out(dst, in2.c_str(), false);
        sfmt(s, "%s}\n%sif (%s) {\n%s    *reinterpret_cast<uint64_t*>(%s + %lluull + wi8) = %s;\n", indent, indent, mask_writer, indent, dst, OFF_ALIVE, alive_word);
        for (uint32_t c = 0; c < nc; ++c) if (rb(c))
            sfmt(s, "%s    *reinterpret_cast<uint64_t*>(%s + %lluull + wi8) = mk%u;\n", indent, dst, (unsigned long long)w->off_present[c], c);
        sfmt(s, "%s}\n", indent);
    };
    s += "    if (in_len) {\n";
    emit_load("a.src", "        ");
    s += "    }\n";
    for (int j = 0; j < V; ++j) sfmt(s, "    const uint64_t ordB_%d = sea_order_lane(e0 + %du);\n", j, j);
    if (marks) {
        sfmt(s, "    // RollbackDespawned markers (despawn.rs:45-46): live-only, never part of a snapshot\n"
                "    const uint64_t mk_dis = *reinterpret_cast<const uint64_t*>(a.live + %lluull + wi8);\n", OFF_DIS);
        for (int j = 0; j < V; ++j) sfmt(s, "    bool dis_%d = (mk_dis >> (sh + %du)) & 1ull;\n", j, j);
        if (V == 1) sfmt(s, "    int df_0 = *reinterpret_cast<const int*>(a.live + %lluull + e0 * 4u);\n", OFF_DF);
        else sfmt(s, "    int df_0, df_1, df_2, df_3; { const i32x4 v = *reinterpret_cast<const i32x4*>(a.live + %lluull + e0 * 4u); df_0 = v.x; df_1 = v.y; df_2 = v.z; df_3 = v.w; }\n", OFF_DF);
    }
    // live-only columns a built-in system READS (BOX_MOVE: Player.handle when Player is not registered for rollback)
    for (size_t i = 0; i < w->systems.size(); ++i) {
        const ggrs_system_desc& d = w->systems[i];
        if (d.kind != GGRS_SYS_BOX_MOVE || rb(d.comp[2])) continue;
        const uint32_t hc = col(d.comp[2], d.word[2]);
        sfmt(s, "    const uint64_t side_mk%zu = *reinterpret_cast<const uint64_t*>(a.live + %lluull + wi8);\n", i, (unsigned long long)w->off_present[d.comp[2]]);
        for (int j = 0; j < V; ++j)
            sfmt(s, "    const bool side_p%zu_%d = (side_mk%zu >> (sh + %du)) & 1ull; const uint64_t side_h%zu_%d = *reinterpret_cast<const uint64_t*>(a.live + %lluull + (e0 >> %d) * %uull + ((e0 & %uull) + %du) * 8ull);\n",
                 i, j, i, j, i, j, (unsigned long long)w->col_off[hc], LT_SHIFT, w->col_ts[hc], (unsigned)(LAYOUT_TILE - 1), j);
    }
    s += "    uint32_t si = 0, sj = 0;\n"
         "    for (uint32_t op = 0; op < a.n_ops; ++op) {\n"
         "        if (!((a.op_bits >> op) & 1ull)) {\n"
         "            // ---------------- SaveWorld\n"
         "            if (si < o_first) { ++si; continue; }                          // another role's snapshot\n"
         "            if (si >= o_last) break;\n"
         "            unsigned char* dst = a.save_dst[si];\n";
    emit_word_from_bits("alive_now", "alive", "            ");
    s += "            if (dst) {\n";
    emit_store("dst", "alive_now", "                ", true);
    s += "                if (t == 0 && tid == 0) {\n"
         "                    Header h; h.len = a.len; h.frame = a.save_frame[si]; h.pad0 = 0; h.active = 0; h.checksum[0] = 0; h.checksum[1] = 0;\n"
         "                    *reinterpret_cast<Header*>(dst) = h;\n"
         "                }\n"
         "            }\n";
    sfmt(s, "            ggrs_u64* acc = s_acc + si * %uu;                                 // this Save's partials of the workgroup (LDS)\n", n_cks + 1);
    for (uint32_t k = 0; k < n_cks; ++k) {
        const uint32_t c = cks_comp[k];
        const Comp& cc = w->comps[c];
        s += "            {   // ComponentChecksumPlugin::update (component_checksum.rs:77-90): per-entity hash, paired with the order index\n"
             "                uint64_t hx = 0;\n";
        for (int j = 0; j < V; ++j) {
            sfmt(s, "                { SeaStream st;");
            for (uint32_t wi : cc.cks_words) {
                const uint32_t cl = col(c, wi);
                if (cc.word_bytes == 8) sfmt(s, " st.unit((uint32_t)w%u_%d); st.unit((uint32_t)(w%u_%d >> 32));", cl, j, cl, j);
                else sfmt(s, " st.unit(w%u_%d);", cl, j);
            }
            sfmt(s, " hx ^= (alive_%d && p%u_%d) ? sea_pair_pre(ordB_%d, st.finish()) : 0ull; }\n", j, c, j, j);
        }
        sfmt(s, "                hx = wave_xor(hx);\n"
                "                if (lane == 0) atomicXor(&acc[%u], (ggrs_u64)hx);\n"
                "            }\n", k);
    }
    s += "            { uint32_t cnt = 0;\n";
    for (int j = 0; j < V; ++j) sfmt(s, "              cnt += (uint32_t)__popcll(__ballot(alive_%d));\n", j);
    sfmt(s, "              if (lane == 0) atomicAdd(&acc[%u], (ggrs_u64)cnt); }\n"
            "            ++si;\n"
            "            if (si >= o_last) break;\n"
            "        } else {\n"
            "            // ---------------- AdvanceWorld: the registered systems, in order\n"
            "            const float dt = __uint_as_float(a.dt_bits[sj]);\n", n_cks);
    if (marks) {
        s += "            const uint32_t sflags = a.step_flags[sj];\n"
             "            const bool defer = sflags & 2u;                                            // despawn_rollback() defers (despawn.rs:129-137)\n";
        for (int j = 0; j < V; ++j)
            sfmt(s, "            if ((sflags & 1u) && dis_%d && df_%d <= a.step_confirmed[sj]) dis_%d = false;   // DespawnConfirmed (despawn.rs:89-112)\n", j, j, j);
    }
    for (size_t i = 0; i < w->systems.size(); ++i) {
        const ggrs_system_desc& d = w->systems[i];
        if (d.kind == GGRS_SYS_CUSTOM) {
            sfmt(s, "            GgrsFrame fr%zu; fr%zu.dt = dt; fr%zu.frame = a.step_frame[sj]; fr%zu.n_inputs = a.n_inputs[sj];\n"
                    "            for (int k = 0; k < 16; ++k) fr%zu.input[k] = a.inputs[sj][k];\n", i, i, i, i, i);
            for (int k = 0; k < 4; ++k) sfmt(s, "            fr%zu.fparam[%d] = %s;\n", i, k, f32_lit(d.fparam[k]).c_str());
            sfmt(s, "            fr%zu.iparam[0] = %lldll; fr%zu.iparam[1] = %lldll;\n", i, (long long)d.iparam[0], i, (long long)d.iparam[1]);
        }
        for (int j = 0; j < V; ++j) switch (d.kind) {
        case GGRS_SYS_PARTICLES_UPDATE: {
            sfmt(s, "            if (alive_%d && p%u_%d && p%u_%d) {                                     // particles.rs:272-280\n", j, d.comp[0], j, d.comp[1], j);
            for (uint32_t k = 0; k < 3; ++k) {
                const uint32_t x = col(d.comp[0], d.word[0] + k), v = col(d.comp[1], d.word[1] + k);
                sfmt(s, "                { const float nv = __uint_as_float(w%u_%d) + %s * dt; w%u_%d = __float_as_uint(nv); w%u_%d = __float_as_uint(__uint_as_float(w%u_%d) + nv * dt); }\n",
                     v, j, f32_lit(d.fparam[k]).c_str(), v, j, x, j, x, j);
            }
            s += "            }\n";
        } break;
        case GGRS_SYS_TTL_DESPAWN: {
            const uint32_t q = col(d.comp[0], d.word[0]);
            sfmt(s, "            if (alive_%d && p%u_%d) { w%u_%d -= 1; if (w%u_%d == 0) alive_%d = false; }      // particles.rs:282-289\n", j, d.comp[0], j, q, j, q, j, j);
        } break;
        case GGRS_SYS_ADD_U32: {
            const uint32_t q = col(d.comp[0], d.word[0]);
            sfmt(s, "            if (alive_%d && p%u_%d) w%u_%d += %uu;                                    // benches/bench.rs:30-46\n", j, d.comp[0], j, q, j, (uint32_t)d.iparam[0]);
        } break;
        case GGRS_SYS_SAT_SUB_DESPAWN: {
            const uint32_t q = col(d.comp[0], d.word[0]);
            sfmt(s, "            if (alive_%d && p%u_%d) {                                              // tests/synctest.rs:37-44\n"
                    "                w%u_%d = w%u_%d >= %uu ? w%u_%d - %uu : 0u;\n"
                    "                if (w%u_%d == 0) {\n", j, d.comp[0], j, q, j, q, j, (uint32_t)d.iparam[0], q, j, (uint32_t)d.iparam[0], q, j);
            if (d.iparam[1] == GGRS_DESPAWN_ROLLBACK) sfmt(s, "                    if (defer) { dis_%d = true; df_%d = a.step_frame[sj]; }\n", j, j);
            sfmt(s, "                    alive_%d = false;\n                }\n            }\n", j);
        } break;
        case GGRS_SYS_BOX_MOVE: {
            const bool h_rb = rb(d.comp[2]);
            char hp[64], hv[64];
            if (h_rb) { snprintf(hp, sizeof hp, "p%u_%d", d.comp[2], j); snprintf(hv, sizeof hv, "w%u_%d", col(d.comp[2], d.word[2]), j); }
            else { snprintf(hp, sizeof hp, "side_p%zu_%d", i, j); snprintf(hv, sizeof hv, "side_h%zu_%d", i, j); }
            const uint32_t x = col(d.comp[0], d.word[0]), v = col(d.comp[1], d.word[1]);
            sfmt(s, "            if (alive_%d && p%u_%d && p%u_%d && %s && %s < a.n_inputs[sj]) {               // box_game.rs:154-206\n"
                    "                float x = __uint_as_float(w%u_%d), y = __uint_as_float(w%u_%d), z = __uint_as_float(w%u_%d);\n"
                    "                float vx = __uint_as_float(w%u_%d), vy = __uint_as_float(w%u_%d), vz = __uint_as_float(w%u_%d);\n",
                 j, d.comp[0], j, d.comp[1], j, hp, hv, x, j, x + 1, j, x + 2, j, v, j, v + 1, j, v + 2, j);
            sfmt(s, "                box_move_math(x, y, z, vx, vy, vz, a.inputs[sj][%s], dt, __uint_as_float(a.aux_bits[sj]), %s, %s, %s);\n"
                    "                w%u_%d = __float_as_uint(x); w%u_%d = __float_as_uint(y); w%u_%d = __float_as_uint(z);\n"
                    "                w%u_%d = __float_as_uint(vx); w%u_%d = __float_as_uint(vy); w%u_%d = __float_as_uint(vz);\n"
                    "            }\n",
                 hv, f32_lit(d.fparam[0]).c_str(), f32_lit(d.fparam[1]).c_str(), f32_lit(d.fparam[3]).c_str(), x, j, x + 1, j, x + 2, j, v, j, v + 1, j, v + 2, j);
        } break;
        case GGRS_SYS_CUSTOM: {
            const ggrs_world::Custom& c = w->customs[d.comp[0]];
            sfmt(s, "            if (alive_%d", j);
            for (uint32_t pz = 0; pz < c.n_pres; ++pz) sfmt(s, " && p%u_%d", c.pres_comp[pz], j);
            sfmt(s, ") {                                                   // user system %u\n"
                    "                GgrsEntity ent; ent.slot = e0 + %du; ent.kill = 0;\n", d.comp[0], j);
            for (uint32_t b = 0; b < c.n_bind; ++b) sfmt(s, "                ent.w[%u] = w%u_%d;\n", b, col(c.comp[b], c.word[b]), j);
            sfmt(s, "                ggrs_sys_%u::ggrs_system(ent, fr%zu);\n", d.comp[0], i);
            for (uint32_t b = 0; b < c.n_bind; ++b)
                sfmt(s, "                w%u_%d = (%s)ent.w[%u];\n", col(c.comp[b], c.word[b]), j, wtype(c.comp[b]), b);
            sfmt(s, "                if (ent.kill) { if (ent.kill == 2 && defer) { dis_%d = true; df_%d = a.step_frame[sj]; } alive_%d = false; }\n"
                    "            }\n", j, j, j);
        } break;
        default: break;
        }
    }
    s += "            ++sj;\n"
         "        }\n"
         "    }\n"
         "    // ---- the live world, written once\n"
         "    if (my_live && writes_live) {\n";
    emit_word_from_bits("alive_now", "alive", "        ");
    emit_store("a.live", "alive_now", "        ", false);
    s += "    }\n";
    if (marks) {
        s += "    if (my_live && a.n_steps) {\n";
        emit_word_from_bits("dis_w", "dis", "        ");
        sfmt(s, "        if (%s) *reinterpret_cast<uint64_t*>(a.live + %lluull + wi8) = dis_w;\n", mask_writer, OFF_DIS);
        if (V == 1) sfmt(s, "        *reinterpret_cast<int*>(a.live + %lluull + e0 * 4u) = df_0;\n", OFF_DF);
        else sfmt(s, "        *reinterpret_cast<i32x4*>(a.live + %lluull + e0 * 4u) = i32x4{df_0, df_1, df_2, df_3};\n", OFF_DF);
        s += "    }\n";
    }
    sfmt(s, "    // ---- this workgroup's partial rows (blockIdx.z: member of a batch of identical checksum-only groups)\n"
            "    __syncthreads();\n"
            "    for (uint32_t i = tid; i < a.n_saves * %uu; i += 256u) {\n"
            "        const uint32_t sv = i / %uu;\n"
            "        if (sv >= o_first && sv < o_last)\n"
            "            a.parts[((uint64_t)blockIdx.z * a.n_saves * %uu + i) * a.part_stride + t] = s_acc[i];\n"
            "    }\n", n_cks + 1, n_cks + 1, n_cks + 1);
    s += "}\n";
    return true;
}

// One compile per distinct (device, source) in the process: worlds of the same shape and capacity share the module
// (a session restart, a test suite).  Modules live until the process ends.
int jit_cached(ggrs_world* w, const std::string& src, hipFunction_t* fn) {
    static std::mutex mu;
    static std::map<std::pair<int, std::string>, hipFunction_t> cache;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(w->device, src);
    auto it = cache.find(key);
    if (it != cache.end()) { *fn = it->second; return GGRS_OK; }
    hipModule_t mod = nullptr;
    const int rc = hiprtc_build(w, src, "generated request-group kernel", "ggrs_jit_tick", &mod, fn);
    if (rc == GGRS_OK) cache[key] = *fn;
    return rc;
}

