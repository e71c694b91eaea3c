// ggrs_hip.hip -- libggrs_hip.so: host runtime + C ABI (see include/ggrs_hip.h).  ONE translation unit, in parts:
//   kernels.hpp / device_prelude.hpp   gfx950 device code (the per-request kernels, the checksum folds, the text shared with generated kernels)
//   host_world.hpp                     struct ggrs_world, knobs, layout of a packed state block, row versions
//   kernel_gen.hpp                     hiprtc plumbing, custom systems, the per-world request-group kernel generator, module cache
//   host_seal.hpp                      sealing: fused-path recognition, kernel generation, arena carve
//   host_requests.hpp                  one launch per request (SaveWorld / LoadWorld / AdvanceWorld), ring, spawns, host-side fold
//   host_groups.hpp                    fused request groups: run_request_groups_gen
//   host_fanout.hpp                    speculative fan-out over RCCL (ggrs_hip_fanout_*)
//   this file                          the C ABI entry points
//
// One ggrs_world owns: a device arena carved into identical *packed state blocks* (one live block + up to max_depth ring
// slots), the host-side ring bookkeeping (an exact mirror of GgrsSnapshots<_, _>, /root/reference/src/snapshot/mod.rs:121-243,
// over slot indices instead of HashMaps), the registered component/system tables and one HIP stream.  Every request becomes
// kernel launches on that stream; the only host<->device synchronisation in handle_requests is one checksum read-back at the
// end of the batch.
//
// There is NO CPU fallback: without a HIP device world creation fails with GGRS_E_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>      // hipExtModuleLaunchKernel: event pairs that ride on a dispatch (profiling)
#include <hip/hiprtc.h>   // types and prototypes only: resolved with dlsym on first use (no link-time dependency)
#include <rccl/rccl.h>      // types and prototypes only: every entry point is resolved with dlsym (no link-time dependency)
#include <dlfcn.h>
#include <math.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <cctype>
#include <chrono>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/ggrs_hip.h"
#include "kernels.hpp"

using namespace ggrs;

#include "host_world.hpp"
namespace {
#include "kernel_gen.hpp"
}
#include "host_seal.hpp"
#include "host_requests.hpp"
#include "host_groups.hpp"

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int ggrs_hip_abi_version(void) { return GGRS_HIP_ABI_VERSION; }
int ggrs_hip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

int ggrs_hip_world_create_ex(const ggrs_world_desc* d, ggrs_world** out) {
    if (!d || !out || d->capacity == 0 || d->capacity > (1ULL << 28)) return GGRS_E_INVALID;   // 32-bit lane offsets in the kernels
    if (d->flags & GGRS_WORLD_LAYOUT_ONLY) {
        ggrs_world* w = new ggrs_world();
        w->layout_only = true;
        w->device = d->device; w->capacity = d->capacity; w->cap_pad = align_up(d->capacity, LAYOUT_TILE);
        w->max_depth = d->max_depth ? d->max_depth : 8; w->flags = d->flags; w->depth = w->max_depth;
        w->knobs = Knobs::from_env();
        *out = w;
        return GGRS_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || d->device >= n) return GGRS_E_NO_DEVICE;
    if (hipSetDevice(d->device) != hipSuccess) return GGRS_E_NO_DEVICE;
    ggrs_world* w = new ggrs_world();
    w->device = d->device; w->capacity = d->capacity; w->cap_pad = align_up(d->capacity, LAYOUT_TILE);
    w->max_depth = d->max_depth ? d->max_depth : 8; w->flags = d->flags;
    w->depth = w->max_depth;
    w->nt_copy = (d->flags & GGRS_WORLD_NT_COPY) != 0;
    w->knobs = Knobs::from_env();
    { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d->device) == hipSuccess && v > 0) w->n_cu = v; }
    if (d->stream) w->stream = (hipStream_t)d->stream;
    else {
        if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) { delete w; return GGRS_E_HIP; }
        w->own_stream = true;
    }
    if (d->arena && d->arena_bytes) { w->arena = (uint8_t*)d->arena; w->arena_bytes = d->arena_bytes; }
    *out = w;
    return GGRS_OK;
}
int ggrs_hip_world_create(int device, uint64_t capacity, uint32_t max_depth, ggrs_world** out) {
    ggrs_world_desc d; memset(&d, 0, sizeof d);
    d.device = device; d.capacity = capacity; d.max_depth = max_depth;
    return ggrs_hip_world_create_ex(&d, out);
}
uint64_t ggrs_hip_arena_bytes(uint64_t capacity, uint32_t max_depth, uint32_t n_components, uint32_t bytes_per_slot) {
    const uint64_t cap_pad = align_up(capacity, LAYOUT_TILE);
    const uint64_t mask = align_up(cap_pad / 8, ALIGN);
    // header + liveness/presence masks, 4 KiB aligned, then the tile-major word columns (bytes_per_slot x 8192 per layout tile; for a component
    // under a Strategy count its Stored words too)
    // ... and the value tags: one u32 per 64-slot unit and word column (at most one column per registered byte)
    const uint64_t state = align_up(align_up(align_up(ALIGN + (1 + (uint64_t)n_components) * mask, 4096) + cap_pad * bytes_per_slot, ALIGN) + (cap_pad / 64) * 4ull * bytes_per_slot, 4096);
    const uint64_t parts = align_up((uint64_t)(n_components + 1) * (cap_pad / TILE + 4096) * 8, ALIGN);
    const uint64_t side = align_up(mask + align_up(cap_pad * 4, ALIGN) + (uint64_t)n_components * mask + cap_pad * bytes_per_slot + (uint64_t)(bytes_per_slot + 1) * ALIGN, 4096);
    // + checksum units, mask scratch, the spawn staging buffer's device twin (GGRS_STAGE_BYTES, default 8 MiB; at most 1 GiB is accounted for here)
    const Knobs k = Knobs::from_env();
    return (uint64_t)(max_depth + 1) * state + side + parts + (uint64_t)(GGRS_MAX_COMPONENTS * GGRS_MAX_CKS_UNITS + 1) * sizeof(UnitDesc) + ALIGN * 4 + k.stage_bytes + 2 * ALIGN;
}
void ggrs_hip_world_destroy(ggrs_world* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    if (w->fanout_backref) *w->fanout_backref = nullptr;          // a ggrs_fanout destroyed after its world (interpreter shutdown order) finds no world
    for (auto& e : w->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : w->prof_pool) (void)hipEventDestroy(e);
    for (auto& b : w->pending) (void)hipEventDestroy(b.ev);
    for (auto& e : w->event_pool) (void)hipEventDestroy(e);
    for (auto& e : w->ff_events) if (e.ev) (void)hipEventDestroy(e.ev);
    for (auto& c : w->customs) if (c.mod) (void)hipModuleUnload(c.mod);
    jit_spec_retire(w);
    jit_release(w->jit_entry);
    delete w->jl; w->jl = nullptr;
    if (w->d_gen_parts) (void)hipFree(w->d_gen_parts);
    if (w->d_branch_parts) (void)hipFree(w->d_branch_parts);
    if (w->d_skip) (void)hipFree(w->d_skip);
    sp_release(w);
    for (void* p : w->spec_allocs) (void)hipFree(p);
    if (w->h_results) (void)hipHostFree(w->h_results);
    if (w->h_stage) (void)hipHostFree(w->h_stage);
    if (w->h_rows) (void)hipHostFree(w->h_rows);
    arena_release(w);
    if (w->own_stream && w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}
int ggrs_hip_specialise_wait(ggrs_world* w) {
    if (!w) return GGRS_E_INVALID;
    int ready = 0;
    for (auto& s : w->spec_tab) {
        if (!s.spec) continue;
        if (s.spec->th.joinable()) s.spec->th.join();
        if (s.spec->state.load(std::memory_order_acquire) == 2) ready = 1;
    }
    return ready;
}
const char* ggrs_hip_last_error(ggrs_world* w) { return w ? w->err.c_str() : "null world"; }

int ggrs_hip_register_component(ggrs_world* w, const char* name, uint32_t word_bytes, uint32_t n_words, uint32_t* comp_id) {
    if (!w || !name) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "register_component after the world was sealed");
    if (w->comps.size() >= GGRS_MAX_COMPONENTS) return w->fail(GGRS_E_INVALID, "at most %d components per world (GGRS_MAX_COMPONENTS)", GGRS_MAX_COMPONENTS);
    if (n_words == 0 || n_words > GGRS_MAX_WORDS || (word_bytes != 1 && word_bytes != 2 && word_bytes != 4 && word_bytes != 8))
        return w->fail(GGRS_E_INVALID, "bad component shape (word_bytes must be 1, 2, 4 or 8; 1..%d words)", GGRS_MAX_WORDS);
    Comp c; c.name = name; c.word_bytes = word_bytes; c.n_words = n_words;
    c.defaults.assign((size_t)word_bytes * n_words, 0);
    w->comps.push_back(c);
    if (comp_id) *comp_id = (uint32_t)w->comps.size() - 1;
    return GGRS_OK;
}
int ggrs_hip_register_component_ex(ggrs_world* w, const char* name, uint32_t word_bytes, uint32_t n_words, uint32_t flags, uint32_t* comp_id) {
    if (flags & ~GGRS_COMP_NO_ROLLBACK) return w ? w->fail(GGRS_E_INVALID, "unknown component flags %u", flags) : GGRS_E_INVALID;
    uint32_t c = 0;
    const int rc = ggrs_hip_register_component(w, name, word_bytes, n_words, &c);
    if (rc) return rc;
    w->comps[c].no_rollback = (flags & GGRS_COMP_NO_ROLLBACK) != 0;
    if (comp_id) *comp_id = c;
    return GGRS_OK;
}
int ggrs_hip_set_component_default(ggrs_world* w, uint32_t c, const void* words) {
    if (!w || c >= w->comps.size() || !words) return GGRS_E_INVALID;
    memcpy(w->comps[c].defaults.data(), words, w->comps[c].defaults.size());
    return GGRS_OK;
}
int ggrs_hip_checksum_component(ggrs_world* w, uint32_t c, const uint32_t* word_idx, uint32_t n) {
    if (!w || c >= w->comps.size()) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "checksum_component after the world was sealed");
    Comp& cc = w->comps[c];
    if (n && !word_idx) return w->fail(GGRS_E_INVALID, "word list is NULL");
    cc.cks_words.clear(); cc.cks_source.clear();
    for (uint32_t k = 0; k < n; ++k) { if (word_idx[k] >= cc.n_words) return w->fail(GGRS_E_INVALID, "word index out of range"); cc.cks_words.push_back(word_idx[k]); }
    if (n > GGRS_MAX_CKS_UNITS) return w->fail(GGRS_E_INVALID, "checksum spec too long");
    cc.checksummed = true;
    return GGRS_OK;
}
int ggrs_hip_checksum_component_custom(ggrs_world* w, uint32_t c, const char* source) {
    if (!w || c >= w->comps.size() || !source || !*source) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "checksum_component after the world was sealed");
    Comp& cc = w->comps[c];
    if (cc.no_rollback) return w->fail(GGRS_E_INVALID, "component %u is not registered for rollback: it has no checksum", c);
    cc.cks_words.clear();
    cc.cks_source = source;
    cc.checksummed = true;
    return GGRS_OK;
}
int ggrs_hip_add_system(ggrs_world* w, const ggrs_system_desc* d) {
    if (!w || !d) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "add_system after the world was sealed");
    if (w->systems.size() >= GGRS_MAX_SYSTEMS) return w->fail(GGRS_E_INVALID, "too many systems");
    const uint32_t nc = (uint32_t)w->comps.size();
    auto comp_ok = [&](uint32_t c, uint32_t wb, uint32_t word, uint32_t span) { return c < nc && w->comps[c].word_bytes == wb && word + span <= w->comps[c].n_words; };
    bool ok = false;
    switch (d->kind) {
    case GGRS_SYS_PARTICLES_UPDATE: ok = comp_ok(d->comp[0], 4, d->word[0], 3) && comp_ok(d->comp[1], 4, d->word[1], 3); break;
    case GGRS_SYS_TTL_DESPAWN: ok = comp_ok(d->comp[0], 8, d->word[0], 1); break;
    case GGRS_SYS_PARTICLES_SPAWN: ok = comp_ok(d->comp[0], 4, 0, 3) && comp_ok(d->comp[1], 4, 0, 3) && comp_ok(d->comp[2], 8, 0, 1); break;
    case GGRS_SYS_ADD_U32: case GGRS_SYS_SAT_SUB_DESPAWN: ok = comp_ok(d->comp[0], 4, d->word[0], 1); break;
    case GGRS_SYS_BOX_MOVE: ok = comp_ok(d->comp[0], 4, d->word[0], 3) && comp_ok(d->comp[1], 4, d->word[1], 3) && comp_ok(d->comp[2], 8, d->word[2], 1); break;
    default: ok = false;
    }
    if (!ok) return w->fail(GGRS_E_INVALID, "system %u does not match the registered components", d->kind);
    w->systems.push_back(*d);
    return GGRS_OK;
}
int ggrs_hip_add_custom_system(ggrs_world* w, const ggrs_custom_system_desc* d) {
    if (!w || !d || !d->source) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "add_custom_system after the world was sealed");
    if (w->systems.size() >= GGRS_MAX_SYSTEMS) return w->fail(GGRS_E_INVALID, "too many systems");
    if (d->n_bindings == 0 || d->n_bindings > GGRS_CUSTOM_MAX_BINDINGS) return w->fail(GGRS_E_INVALID, "custom system: 1..%d bindings", GGRS_CUSTOM_MAX_BINDINGS);
    ggrs_world::Custom c;
    c.name = d->name ? d->name : "custom";
    c.n_bind = d->n_bindings;
    for (uint32_t i = 0; i < c.n_bind; ++i) {
        if (d->comp[i] >= w->comps.size() || d->word[i] >= w->comps[d->comp[i]].n_words)
            return w->fail(GGRS_E_INVALID, "custom system '%s': binding %u names word %u of component %u, which is not registered", c.name.c_str(), i, d->word[i], d->comp[i]);
        c.comp[i] = d->comp[i]; c.word[i] = d->word[i];
        bool seen = false;
        for (uint32_t p = 0; p < c.n_pres; ++p) seen |= c.pres_comp[p] == d->comp[i];
        if (!seen) c.pres_comp[c.n_pres++] = d->comp[i];
    }
    DeviceGuard dg(w);
    c.source = d->source;
    c.may_defer = source_has_token(c.source, "despawn_rollback") || source_has_token(c.source, "kill");
    const std::string src = custom_source(w, c, d->source);
    const std::string what = "custom system '" + c.name + "'";
    const int rc = hiprtc_build(w, src, what.c_str(), "ggrs_custom_kernel", w->layout_only ? nullptr : &c.mod, &c.fn);
    if (rc) return rc;
    ggrs_system_desc sd; memset(&sd, 0, sizeof sd);
    sd.kind = GGRS_SYS_CUSTOM; sd.comp[0] = (uint32_t)w->customs.size();
    sd.iparam[0] = d->iparam[0]; sd.iparam[1] = d->iparam[1];
    for (int k = 0; k < 4; ++k) sd.fparam[k] = d->fparam[k];
    w->customs.push_back(std::move(c));
    w->systems.push_back(sd);
    return GGRS_OK;
}
// RollbackApp::rollback_component_with_{copy,clone,reflect} are instances of  ComponentSnapshotPlugin<S: Strategy>  (snapshot/strategy.rs:22-40,
// component_snapshot.rs:42-63): S::Stored is what a snapshot holds.  Here: the Stored words and the HIP C++ of store / load.
int ggrs_hip_register_component_strategy(ggrs_world* w, uint32_t c, uint32_t stored_word_bytes, uint32_t stored_n_words, const char* source) {
    if (!w || c >= w->comps.size() || !source || !*source) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "register_component_strategy after the world was sealed");
    Comp& cc = w->comps[c];
    if (cc.no_rollback) return w->fail(GGRS_E_INVALID, "component %u is not registered for rollback: it has no snapshot strategy", c);
    if (stored_n_words == 0 || stored_n_words > GGRS_MAX_WORDS || (stored_word_bytes != 1 && stored_word_bytes != 2 && stored_word_bytes != 4 && stored_word_bytes != 8))
        return w->fail(GGRS_E_INVALID, "bad Stored shape (word bytes must be 1, 2, 4 or 8; 1..%d words)", GGRS_MAX_WORDS);
    cc.s_word_bytes = stored_word_bytes; cc.s_n_words = stored_n_words; cc.strat_source = source;
    return GGRS_OK;
}
// PlayerInputs<T>(Vec<(T::Input, InputStatus)>)  (src/lib.rs:98; inserted by AdvanceFrame, schedule_systems.rs:262-265)
int ggrs_hip_set_input_layout(ggrs_world* w, uint32_t input_bytes, uint32_t max_players) {
    if (!w) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "set_input_layout after the world was sealed");
    if (!w->systems.empty()) return w->fail(GGRS_E_INVALID, "set_input_layout must precede the first system of the schedule (custom systems are compiled against the layout when they are added)");
    if (input_bytes == 0 || input_bytes > GGRS_MAX_INPUT_BYTES || max_players == 0 || max_players > GGRS_MAX_PLAYERS)
        return w->fail(GGRS_E_INVALID, "input layout: 1..%d bytes per player, 1..%d players", GGRS_MAX_INPUT_BYTES, GGRS_MAX_PLAYERS);
    w->input_bytes = input_bytes; w->max_players = max_players;
    return GGRS_OK;
}
// A GgrsSchedule system that spawns Rollback entities (snapshot/rollback.rs:45-59; examples/stress_tests/particles.rs:258-270 is the built-in
// GGRS_SYS_PARTICLES_SPAWN): how many is the host's decision per AdvanceFrame (ggrs_request::spawn_count), what they are is the user's source.
int ggrs_hip_add_spawn_system(ggrs_world* w, const ggrs_spawn_system_desc* d) {
    if (!w || !d || !d->source) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "add_spawn_system after the world was sealed");
    if (w->systems.size() >= GGRS_MAX_SYSTEMS) return w->fail(GGRS_E_INVALID, "too many systems");
    for (auto& s : w->systems) if (s.kind == GGRS_SYS_SPAWN_CUSTOM || s.kind == GGRS_SYS_PARTICLES_SPAWN) return w->fail(GGRS_E_INVALID, "the schedule already holds a spawn system (one per world)");
    if (d->n_bindings > GGRS_CUSTOM_MAX_BINDINGS) return w->fail(GGRS_E_INVALID, "spawn system: at most %d bindings", GGRS_CUSTOM_MAX_BINDINGS);
    if (d->bundle_mask == 0 || (w->comps.size() < 64 && (d->bundle_mask >> w->comps.size()) != 0)) return w->fail(GGRS_E_INVALID, "spawn system: bundle_mask names no / unregistered components");
    ggrs_world::SpawnSys sp;
    sp.name = d->name ? d->name : "spawn"; sp.source = d->source; sp.n_bind = d->n_bindings; sp.bundle_mask = d->bundle_mask; sp.payload_stride = d->payload_stride;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if (((sp.bundle_mask >> c) & 1ull) && w->comps[c].no_rollback)
        return w->fail(GGRS_E_INVALID, "spawn system '%s': component %u of the bundle is not registered for rollback", sp.name.c_str(), c);
    for (uint32_t i = 0; i < sp.n_bind; ++i) {
        if (d->comp[i] >= w->comps.size() || d->word[i] >= w->comps[d->comp[i]].n_words || !((sp.bundle_mask >> d->comp[i]) & 1ull))
            return w->fail(GGRS_E_INVALID, "spawn system '%s': binding %u names word %u of component %u, which is not a word of the bundle", sp.name.c_str(), i, d->word[i], d->comp[i]);
        sp.comp[i] = d->comp[i]; sp.word[i] = d->word[i];
    }
    ggrs_system_desc sd; memset(&sd, 0, sizeof sd);
    sd.kind = GGRS_SYS_SPAWN_CUSTOM; sd.comp[0] = (uint32_t)w->spawn_customs.size();
    sd.iparam[0] = d->iparam[0]; sd.iparam[1] = d->iparam[1];
    for (int k = 0; k < 4; ++k) sd.fparam[k] = d->fparam[k];
    w->spawn_customs.push_back(std::move(sp));
    w->systems.push_back(sd);
    return GGRS_OK;
}
// the file name a shipped code object of this source must carry (scripts/aot_build.py): NUL-terminated into buf
int ggrs_hip_aot_object_name(const char* source, char* buf, uint64_t cap) {
    if (!source || !buf || cap < 40) return GGRS_E_INVALID;
    const std::string n = jit_aot_name(source);
    memcpy(buf, n.c_str(), n.size() + 1);
    return GGRS_OK;
}
int ggrs_hip_generated_kernel_source(ggrs_world* w, uint32_t form, char* buf, uint64_t cap, uint64_t* needed, int compile) {
    if (!w || (form != GGRS_KERNEL_FORM_TILES && form != GGRS_KERNEL_FORM_STEADY)) return GGRS_E_INVALID;
    if (!w->sealed) {
        if (!w->layout_only) { DeviceGuard dg(w); const int rc = seal(w); if (rc) return rc; }
        else build_layout(w);                                      // host arithmetic only: offsets of every mask and column
    }
    std::string src;
    if (!jit_source(w, src)) return w->fail(GGRS_E_INVALID, "the kernel generator does not cover this world (a system writes a live-only component, or more than %u four-byte units / %u words per entity)", JIT_MAX_UNITS, JIT_MAX_COLS);
    if (form == GGRS_KERNEL_FORM_STEADY) {
        src = jit_specialise(src, jit_steady_sig(w));
        if (src.empty()) return w->fail(GGRS_E_INVALID, "the generated kernel's text could not be specialised");
    }
    if (needed) *needed = src.size() + 1;
    if (buf && cap) { const uint64_t n = std::min<uint64_t>(cap, src.size() + 1); memcpy(buf, src.c_str(), n); buf[n - 1] = 0; }
    if (compile) { hipFunction_t fn = nullptr; return hiprtc_build(w, src, "generated request-group kernel", "ggrs_jit_tick", nullptr, &fn); }
    return GGRS_OK;
}
// Test hook (not part of the C ABI: no ggrs_hip_ prefix, not in include/ggrs_hip.h): the specialiser's token rule on arbitrary text --
// tests/test_generated_kernel.py feeds it identifiers no generator emits yet.  Returns the number of replacements, -1 when `out` is too small.
int ggrs_dbg_replace_token(const char* body, const char* tok, const char* val, char* out, uint64_t cap) {
    if (!body || !tok || !val || !out || !*tok) return -1;
    std::string b = body;
    const uint32_t n = jit_replace_token(b, tok, val);
    if (b.size() + 1 > cap) return -1;
    memcpy(out, b.c_str(), b.size() + 1);
    return (int)n;
}
// Test hooks (no ggrs_hip_ prefix, not in the header; neither changes a result):
//   ggrs_dbg_set_lazy_live     0 = every tick writes the live block (the A/B of profiles/r05h), 2 = every eligible list leaves it unwritten whatever its size and
//                              streak (the fuzzer: tests/test_fuzz_requests.py), 1 = the default policy
//   ggrs_dbg_set_spec_shapes   places in the table of group shapes / specialised kernels (default 16), so that a P2P session's eight rollback lengths
//                              exercise the least-recently-used eviction (tests/test_gpu_gen_groups.py)
//   ggrs_dbg_set_value_tags    0 = no Save skips a column on its value tags, 1 = every world does, -1 = by size (the default: host_world.hpp VTAGS_MIN_BYTES);
//                              before the world is sealed
int ggrs_dbg_set_value_tags(ggrs_world* w, int mode) { if (!w || w->sealed) return -1; w->vtags_mode = mode; return 0; }
int ggrs_dbg_set_lazy_live(ggrs_world* w, int on) { if (!w) return -1; w->lazy_live_on = on; return 0; }
int ggrs_dbg_set_spec_shapes(ggrs_world* w, int n) { if (!w || n < 1 || n > 64) return -1; w->spec_shapes = n; return 0; }
int ggrs_hip_set_frame_rate(ggrs_world* w, uint64_t fps) { if (!w || fps == 0) return GGRS_E_INVALID; w->fps = fps; return GGRS_OK; }

// entry points that read or edit the live block's BYTES: a lazily skipped live block (host_groups.hpp) is materialised first; a world whose systems spawn
// on the device learns its RollbackOrdered::len from the last launch (len_sync)
static int seal_live(ggrs_world* w) { int rc = seal(w); if (rc) return rc; rc = len_sync(w); if (rc) return rc; return materialise_live(w); }

int ggrs_hip_spawn(ggrs_world* w, uint64_t count, uint64_t comp_mask, const void* const* cols, uint64_t* first_slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (!range_ok(w->len, count, w->capacity)) return w->fail(GGRS_E_CAPACITY, "spawn of %llu exceeds capacity %llu", (unsigned long long)count, (unsigned long long)w->capacity);
    const uint64_t first = w->len;
    if (first_slot) *first_slot = first;
    if (count == 0) return GGRS_OK;
    uint32_t ci = 0;
    for (uint32_t c = 0; c < w->comps.size(); ++c) {
        if (!((comp_mask >> c) & 1ULL)) continue;
        const Comp& cc = w->comps[c];
        bool any_null = false;
        for (uint32_t k = 0; k < cc.n_words; ++k) if (!cols || !cols[ci + k]) any_null = true;
        if (any_null) { rc = fill_defaults(w, c, first, count); if (rc) return rc; }
        for (uint32_t k = 0; k < cc.n_words; ++k) {
            const void* src = cols ? cols[ci + k] : nullptr;
            if (src) { rc = copy_column(w, cc.col_base + k, first, count, const_cast<void*>(src), true); if (rc) return rc; }
        }
        ver_touch_comp(w, c); live_tags_lost_comp(w, c);            // new rows in every column of the bundle
        ci += cc.n_words;
    }
    ver_sync_live(w);
    rc = set_masks_for_range(w, first, count, comp_mask); if (rc) return rc;
    w->len += count;
    w->live.dirty_len = std::max(w->live.dirty_len, w->len);
    if (w->dev_spawn) { const uint64_t l = w->len; HIPCHK(w, hipMemcpyAsync(w->live.ptr, &l, 8, hipMemcpyHostToDevice, w->stream)); }   // the kernels read RollbackOrdered::len from the block's header
    w->pending_valid = false;
    HIPCHK(w, hipStreamSynchronize(w->stream));     // host buffers may be freed on return
    return GGRS_OK;
}
int ggrs_hip_despawn(ggrs_world* w, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (slot >= w->len) return w->fail(GGRS_E_INVALID, "slot out of range");
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_alive, slot, 0);
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_despawn_rollback(ggrs_world* w, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (slot >= w->len) return w->fail(GGRS_E_INVALID, "slot out of range");
    if (w->confirmed < w->frame) {                 // despawn.rs:129-137: insert RollbackDespawned(frame)
        w->marks_possible = true;
        hipLaunchKernelGGL(k_mark_despawned, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_alive, w->marks, slot, w->frame);
    } else {                                       // despawn.rs:140-142: frame already confirmed -> plain despawn
        hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_alive, slot, 0);
    }
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_insert_component(ggrs_world* w, uint32_t c, uint64_t slot, const void* words) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size() || slot >= w->len || !words) return w->fail(GGRS_E_INVALID, "bad insert_component arguments");
    const Comp& cc = w->comps[c];
    for (uint32_t k = 0; k < cc.n_words; ++k)
        { rc = copy_column(w, cc.col_base + k, slot, 1, const_cast<uint8_t*>((const uint8_t*)words + (size_t)k * cc.word_bytes), true); if (rc) return rc; }
    ver_touch_comp(w, c); live_tags_lost_comp(w, c); ver_sync_live(w);
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_present[c], slot, 1);
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}
int ggrs_hip_remove_component(ggrs_world* w, uint32_t c, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size() || slot >= w->len) return w->fail(GGRS_E_INVALID, "bad remove_component arguments");
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_present[c], slot, 0);
    HIPCHK(w, hipGetLastError());
    ver_touch(w, ver_presence(w, c)); ver_sync_live(w);           // the presence mask changed, the columns did not
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_upload_word(ggrs_world* w, uint32_t c, uint32_t word, uint64_t first, uint64_t count, const void* src) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size() || word >= w->comps[c].n_words || !range_ok(first, count, w->capacity) || !src) return w->fail(GGRS_E_INVALID, "bad upload_word arguments");
    const Comp& cc = w->comps[c];
    rc = copy_column(w, cc.col_base + word, first, count, const_cast<void*>(src), true); if (rc) return rc;
    ver_touch(w, cc.col_base + word); live_tags_lost(w, cc.col_base + word); ver_sync_live(w);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_download_word(ggrs_world* w, uint32_t c, uint32_t word, uint64_t first, uint64_t count, void* dst) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size() || word >= w->comps[c].n_words || !range_ok(first, count, w->capacity) || !dst) return w->fail(GGRS_E_INVALID, "bad download_word arguments");
    const Comp& cc = w->comps[c];
    rc = copy_column(w, cc.col_base + word, first, count, dst, false); if (rc) return rc;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}
static int download_mask(ggrs_world* w, uint64_t off, uint64_t* dst, uint64_t n) {
    const uint64_t have = w->cap_pad / 64;
    const uint64_t m = n < have ? n : have;
    HIPCHK(w, hipMemcpyAsync(dst, w->live.ptr + off, m * 8, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (uint64_t k = m; k < n; ++k) dst[k] = 0;
    return GGRS_OK;
}
int ggrs_hip_download_alive(ggrs_world* w, uint64_t* dst, uint64_t n) {
    if (!w || !dst) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    return download_mask(w, w->off_alive, dst, n);
}
int ggrs_hip_download_present(ggrs_world* w, uint32_t c, uint64_t* dst, uint64_t n) {
    if (!w || !dst) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size()) return w->fail(GGRS_E_INVALID, "bad component");
    rc = download_mask(w, w->off_present[c], dst, n); if (rc) return rc;
    if (w->comps[c].no_rollback && n) {
        // a non-rollback component dies with its entity: it exists while the entity is alive or
        // disabled (its live-only presence bit is cleaned lazily, when the slot is re-created)
        std::vector<uint64_t> a(n), d(n);
        rc = download_mask(w, w->off_alive, a.data(), n); if (rc) return rc;
        rc = download_mask(w, w->marks.off_disabled, d.data(), n); if (rc) return rc;
        for (uint64_t k = 0; k < n; ++k) dst[k] &= (a[k] | d[k]);
    }
    return GGRS_OK;
}
int ggrs_hip_download_disabled(ggrs_world* w, uint64_t* dst, uint64_t n) {
    if (!w || !dst) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    return download_mask(w, w->marks.off_disabled, dst, n);
}
int ggrs_hip_download_despawned_frames(ggrs_world* w, uint64_t first, uint64_t count, int32_t* frames) {
    if (!w || !frames) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (!range_ok(first, count, w->capacity)) return w->fail(GGRS_E_INVALID, "bad download_despawned_frames range");
    if (count) HIPCHK(w, hipMemcpyAsync(frames, w->live.ptr + w->marks.off_dframe + first * 4, (size_t)count * 4, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}
int ggrs_hip_column_device_ptr(ggrs_world* w, uint32_t c, uint32_t word, void** p, uint64_t* tile_stride) {
    if (!w || !p) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    if (c >= w->comps.size() || word >= w->comps[c].n_words) return w->fail(GGRS_E_INVALID, "bad column");
    *p = w->live.ptr + w->col_off[w->comps[c].col_base + word];
    w->col_ext[w->comps[c].col_base + word] = 1;                   // whoever holds this pointer may write the column at any time: no row-version shortcuts for it
    if (tile_stride) *tile_stride = w->col_ts[w->comps[c].col_base + word];
    return GGRS_OK;
}
uint64_t ggrs_hip_len(ggrs_world* w) { if (!w) return 0; if (w->len_stale) { DeviceGuard dg(w); (void)len_sync(w); } return w->len; }
int ggrs_hip_active_count(ggrs_world* w, uint64_t* out) {
    if (!w || !out) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    const uint64_t n = (w->live.dirty_len + 63) / 64;
    std::vector<uint64_t> m(n ? n : 1, 0);
    if (n) { rc = download_mask(w, w->off_alive, m.data(), n); if (rc) return rc; }
    uint64_t a = 0; for (uint64_t k = 0; k < n; ++k) a += (uint64_t)__builtin_popcountll(m[k]);
    *out = a;
    return GGRS_OK;
}

int32_t ggrs_hip_frame(ggrs_world* w) { return w ? w->frame : 0; }
int ggrs_hip_set_frame(ggrs_world* w, int32_t f) { if (!w) return GGRS_E_INVALID; w->frame = f; return GGRS_OK; }
int ggrs_hip_set_depth(ggrs_world* w, uint32_t d) {
    if (!w) return GGRS_E_INVALID;
    if (d > w->max_depth) return w->fail(GGRS_E_INVALID, "depth %u exceeds provisioned max_depth %u", d, w->max_depth);
    w->depth = d;
    return GGRS_OK;
}
int ggrs_hip_set_confirmed(ggrs_world* w, int has, int32_t f) { if (!w) return GGRS_E_INVALID; w->has_confirmed = has != 0; w->confirmed = f; return GGRS_OK; }
int ggrs_hip_has_snapshot(ggrs_world* w, int32_t f) {
    if (!w) return 0;
    for (int32_t x : w->ring_frame) if (x == f) return 1;
    return 0;
}
uint64_t ggrs_hip_snapshot_count(ggrs_world* w) { return w ? w->ring_frame.size() : 0; }
int ggrs_hip_set_synctest_check_distance(ggrs_world* w, int32_t cd) { if (!w) return GGRS_E_INVALID; w->synctest_cd = cd; return GGRS_OK; }

int ggrs_hip_save(ggrs_world* w, uint64_t out[2]) {
    if (!w) return GGRS_E_INVALID;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "synchronous request while %zu enqueued batches are uncollected", w->pending.size());
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (GroupRunner run = group_runner(w)) { ggrs_request r; memset(&r, 0, sizeof r); r.kind = GGRS_REQ_SAVE; r.frame = w->frame; return run(w, &r, 1, out, 0, true, nullptr); }
    rc = do_save(w, 0); if (rc) return rc;
    return read_back(w, 1, out);
}
int ggrs_hip_load(ggrs_world* w, int32_t frame) {
    if (!w) return GGRS_E_INVALID;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "synchronous request while %zu enqueued batches are uncollected", w->pending.size());
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (GroupRunner run = group_runner(w)) { ggrs_request r; memset(&r, 0, sizeof r); r.kind = GGRS_REQ_LOAD; r.frame = frame; return run(w, &r, 1, nullptr, 0, true, nullptr); }
    return do_load(w, frame);
}
int ggrs_hip_advance(ggrs_world* w, uint32_t dt_bits, const uint8_t* inputs, uint32_t n_inputs,
                     uint64_t spawn_count, const float* vx, const float* vy) {
    if (!w) return GGRS_E_INVALID;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "synchronous request while %zu enqueued batches are uncollected", w->pending.size());
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    ggrs_request r; memset(&r, 0, sizeof r);
    r.kind = GGRS_REQ_ADVANCE; r.dt_bits = dt_bits; r.inputs = inputs; r.n_inputs = n_inputs;
    r.spawn_count = spawn_count; r.spawn_vx = vx; r.spawn_vy = vy;
    rc = validate_requests(w, &r, 1); if (rc) return rc;
    if (GroupRunner run = group_runner(w)) return run(w, &r, 1, nullptr, 0, true, nullptr);
    return do_advance(w, r);
}

int ggrs_hip_handle_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint64_t* checksums_out) {
    if (!w || (!reqs && n)) return GGRS_E_INVALID;
    TraceRange tr("HandleRequests");
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "handle_requests while %zu enqueued batches are uncollected", w->pending.size());
    rc = validate_requests(w, reqs, n); if (rc) return rc;
    if (GroupRunner run = group_runner(w)) {
        rc = run(w, reqs, n, checksums_out, 0, true, nullptr);
        if (rc && w->stream) { (void)hipStreamSynchronize(w->stream); w->folds.clear(); w->rows_used = 0; w->rows_tail = 0; }   // (nothing is pending in the synchronous API)
        return rc;
    }
    uint32_t ns = 0;
    for (uint32_t i = 0; i < n && rc == GGRS_OK; ++i) {
        const ggrs_request& r = reqs[i];
        trace_request(w, r);
        apply_synctest_confirmed(w);
        switch (r.kind) {
        case GGRS_REQ_SAVE:
            if (ns >= w->max_results) {                   // flush a full result page
                rc = read_back(w, ns, checksums_out); if (rc) break;
                checksums_out += 2 * (uint64_t)ns; ns = 0;
            }
            rc = do_save(w, ns); ++ns; break;
        case GGRS_REQ_LOAD: rc = do_load(w, r.frame); break;
        case GGRS_REQ_ADVANCE: rc = do_advance(w, r); break;
        default: rc = w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
        }
    }
    if (rc) { if (w->stream) (void)hipStreamSynchronize(w->stream); return rc; }
    return read_back(w, ns, checksums_out);
}
// Asynchronous pair: the request list is only ENQUEUED on the world's stream (all host-side bookkeeping --
// frame counters, ring push/confirm/rollback -- happens now, in request order); the Checksum(u128)s are
// fetched later, oldest batch first.  ggrs reads a SaveGameState cell no earlier than the next
// advance_frame(), so a host shim collects right before that call and the GPU tick overlaps the rest of
// the host's frame instead of blocking it.
// bs != nullptr: a branch step (ggrs_hip_fanout_step_branches) -- `reqs` is its prefix, the branches' launch and Checksum(u128)s follow in the same batch
static int enqueue_impl(ggrs_world* w, const ggrs_request* reqs, uint32_t n, const ggrs_branch_step* bs, BranchKeep* keep, uint32_t* n_saves_out) {
    if (!w || (!reqs && n)) return GGRS_E_INVALID;
    TraceRange tr("HandleRequests");
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    const double t_in = w->tl.on ? tl_now_us() : 0;
    rc = validate_requests(w, reqs, n); if (rc) return rc;
    if (bs) { rc = validate_branch_step(w, *bs); if (rc) return rc; }
    if (w->tl.on) w->tl.validate_us += tl_now_us() - t_in;
    uint32_t n_save = 0;
    for (uint32_t i = 0; i < n; ++i) n_save += reqs[i].kind == GGRS_REQ_SAVE;
    const uint32_t n_save_prefix = n_save;
    if (bs) n_save += bs->n_branches * ((bs->flags & GGRS_BRANCH_SAVE_LAST) ? bs->n_frames : bs->n_frames - 1);
    if (n_save > w->max_results / 4 || w->pending_results + n_save > w->max_results / 2 || w->pending.size() >= 16)
        return w->fail(GGRS_E_INVALID, "too many uncollected checksums (%u pending + %u new): call ggrs_hip_collect_checksums", w->pending_results, n_save);
    ggrs_world::PendingBatch b;
    b.first = (w->res_head + n_save > w->max_results) ? 0u : w->res_head;
    b.count = n_save;
    const size_t folds_before = w->folds.size();
    const uint64_t ff_id0 = w->ff_next_id;
    if (w->event_pool.empty()) { hipEvent_t e; HIPCHK(w, hipEventCreateWithFlags(&e, hipEventDisableTiming)); w->event_pool.push_back(e); }
    b.ev = w->event_pool.back(); w->event_pool.pop_back();
    w->batch_ev_attached = false;
    if (w->dev_results_dst) w->dev_results_first = b.first;          // the consumer's device copy starts with this batch's first Checksum(u128)
    if (GroupRunner run = group_runner(w)) {
        w->batch_ev = b.ev;
        rc = run(w, reqs, n, nullptr, b.first, false, nullptr);
        if (rc == GGRS_OK && bs) rc = run_branch_step(w, *bs, b.first + n_save_prefix, keep);
        w->batch_ev = nullptr;
        if (rc) {
            w->event_pool.push_back(b.ev); (void)hipStreamSynchronize(w->stream);
            while (w->folds.size() > folds_before) w->folds.pop_back();
            if (w->ff_pending.valid && w->ff_pending.id >= ff_id0) w->ff_pending.valid = false;      // its HostFold is gone with the failed list
            return rc;
        }
    } else {
        // worlds without request-group kernels: one launch per request, enqueued like the groups are; every
        // Save's fold writes straight into its slot of the pinned result ring, nothing is waited for here
        uint32_t ns = 0;
        for (uint32_t i = 0; i < n && rc == GGRS_OK; ++i) {
            const ggrs_request& r = reqs[i];
            trace_request(w, r);
            apply_synctest_confirmed(w);
            switch (r.kind) {
            case GGRS_REQ_SAVE: rc = do_save(w, b.first + ns); ++ns; break;
            case GGRS_REQ_LOAD: rc = do_load(w, r.frame); break;
            case GGRS_REQ_ADVANCE: rc = do_advance(w, r); break;
            default: rc = w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
            }
        }
        if (rc) { (void)hipStreamSynchronize(w->stream); return rc; }
    }
    if (!w->batch_ev_attached) HIPCHK(w, hipEventRecord(b.ev, w->stream));      // (else the event rides on the list's last kernel)
    w->res_head = b.first + n_save; w->pending_results += n_save;
    b.n_folds = (uint32_t)(w->folds.size() - folds_before);
    b.stage_end = w->stage_used;
    if (n_saves_out) *n_saves_out = n_save;
    w->pending.push_back(std::move(b));
    if (w->tl.on) { w->tl.enqueue_us += tl_now_us() - t_in; ++w->tl.n_enqueue; }
    return GGRS_OK;
}
int ggrs_hip_enqueue_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out) { return enqueue_impl(w, reqs, n, nullptr, nullptr, n_saves_out); }
// the batch's event, waited for by polling (hipEventQuery returns in 0.06 us; hipEventSynchronize on an event that is NOT complete yet costs
// 0.7 us more per tick of a 5.6 us kernel: scripts/ubench_launch, profiles/r05a) for up to GGRS_SPIN_WAIT_US, then by the runtime's wait
static int wait_batch_event(ggrs_world* w, hipEvent_t ev) {
    if (w->knobs.spin_wait_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t it = 1; ; ++it) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return GGRS_OK;
            if (e != hipErrorNotReady) { (void)hipGetLastError(); break; }
            if ((it & 31u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(w->knobs.spin_wait_us)) break;
            cpu_relax();
        }
        (void)hipGetLastError();                                     // (hipErrorNotReady is sticky in hipGetLastError)
    }
    HIPCHK(w, hipEventSynchronize(ev));
    return GGRS_OK;
}
int ggrs_hip_collect_checksums(ggrs_world* w, uint64_t* checksums_out, uint32_t max_saves, uint32_t* n_saves_out) {
    if (!w) return GGRS_E_INVALID;
    if (w->pending.empty()) return w->fail(GGRS_E_INVALID, "no enqueued batch to collect");
    DeviceGuard dg(w);
    ggrs_world::PendingBatch& b = w->pending.front();
    if (b.count > max_saves || (b.count && !checksums_out)) return w->fail(GGRS_E_INVALID, "oldest batch holds %u checksums, room for %u", b.count, max_saves);
    const double t_in = w->tl.on ? tl_now_us() : 0;
    // a fold-forward group whose rows nothing took along yet: queue k_ff_fold BEFORE waiting (it runs right behind the batch's kernel)
    if (w->ff_pending.valid) for (uint32_t k = 0; k < b.n_folds && k < w->folds.size(); ++k) if (w->folds[k].ff_id == w->ff_pending.id) { int rc = ff_flush(w); if (rc) return rc; break; }
    int rc = wait_batch_event(w, b.ev); if (rc) return rc;
    if (w->tl.on) w->tl.wait_us += tl_now_us() - t_in;
    rc = run_host_folds(w, b.n_folds); if (rc) return rc;
    if (b.count) {
        if (!b.host.empty()) memcpy(checksums_out, b.host.data(), (size_t)b.count * 16);
        else memcpy(checksums_out, w->h_results + 2 * (size_t)b.first, (size_t)b.count * 16);
    }
    if (n_saves_out) *n_saves_out = b.count;
    w->pending_results -= b.count;
    w->event_pool.push_back(b.ev);
    const uint64_t stage_end = b.stage_end;
    w->pending.pop_front();
    // the batch's launches are done: its spawn payloads (and everything staged before them) are free again
    if (w->pending.empty()) stage_ring_reset(w); else if (stage_end) w->stage_tail = stage_end;
    // Nothing is queued behind this batch and its results were seen through pinned memory: the runtime itself has not looked at the stream yet.  One
    // non-blocking query lets it retire the finished commands now, so that a device-wide synchronise that follows finds an idle stream
    if (w->pending.empty() && !w->prof) { (void)hipStreamQuery(w->stream); (void)hipGetLastError(); }
    if (w->tl.on) { w->tl.collect_us += tl_now_us() - t_in; ++w->tl.n_collect; }
    return GGRS_OK;
}
// Where the HOST spends a tick.  enable: 1 = reset and start, 0 = stop, -1 = leave as is.  us_out[GGRS_TIMELINE_FIELDS] = microseconds
// summed since the start: {enqueue call, of it: validation, of it: launch calls (hipModuleLaunchKernel / hipExtModuleLaunchKernel), collect call, of it: the
// batch event, of it: fold-forward tags, of it: hashing / folding on the host}; counts_out[3] = {enqueue calls, collect calls, launches}.
int ggrs_hip_host_timeline(ggrs_world* w, int enable, double* us_out, uint64_t* counts_out) {
    if (!w) return GGRS_E_INVALID;
    if (us_out) { const HostTimeline& t = w->tl; const double v[GGRS_TIMELINE_FIELDS] = {t.enqueue_us, t.validate_us, t.launch_us, t.collect_us, t.wait_us, t.tag_wait_us, t.fold_us}; memcpy(us_out, v, sizeof v); }
    if (counts_out) { counts_out[0] = w->tl.n_enqueue; counts_out[1] = w->tl.n_collect; counts_out[2] = w->tl.n_launches; }
    if (enable == 1) { w->tl = HostTimeline{}; w->tl.on = true; }
    else if (enable == 0) w->tl.on = false;
    return GGRS_OK;
}
uint32_t ggrs_hip_pending_batches(ggrs_world* w) { return w ? (uint32_t)w->pending.size() : 0; }

int ggrs_hip_synchronize(ggrs_world* w) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    if (w->sealed) { const int rc = ff_flush(w); if (rc) return rc; }          // rows a fold-forward launch left behind: nothing may be left half-done on the stream
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}

uint64_t ggrs_hip_state_bytes(ggrs_world* w) { if (!w) return 0; DeviceGuard dg(w); if (seal(w)) return 0; return w->state_bytes; }
int ggrs_hip_live_state_ptr(ggrs_world* w, void** p) {
    if (!w || !p) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    // keep the header current so an exported block is self-describing
    Header h = header_of(w);
    HIPCHK(w, hipMemcpyAsync(w->live.ptr, &h, 16, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    *p = w->live.ptr;
    w->live_handed_out = true;                                      // the caller may read the block after any later tick: it is written by every tick from now on
    return GGRS_OK;
}
int ggrs_hip_adopt_live_state(ggrs_world* w) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal_live(w); if (rc) return rc;
    Header h;
    HIPCHK(w, hipMemcpyAsync(&h, w->live.ptr, sizeof h, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    if (h.len > w->capacity) return w->fail(GGRS_E_INVALID, "adopted state has len %llu > capacity", (unsigned long long)h.len);
    w->len = h.len; w->frame = h.frame;
    w->live.dirty_len = std::max(w->live.dirty_len, w->len);
    ver_touch_all(w); w->live.tag_ok = 0; ver_sync_live(w);        // every column is new
    w->pending_valid = false;
    return GGRS_OK;
}

// Which kernel serves this world's request lists right now, on what kind of arena, and the state of the run-time
// compiler -- `key=value` lines (NUL-terminated, truncated to cap).  Nothing here is needed to USE the library: it is what an
// operator reads when a world is slower than expected (e.g. libhiprtc.so missing from a deployment image).
int ggrs_hip_world_kernel_info(ggrs_world* w, char* buf, uint64_t cap, uint64_t* needed) {
    if (!w || (!buf && cap)) return GGRS_E_INVALID;
    std::string s;
    char line[768];
    auto add = [&](const char* k, const std::string& v) { snprintf(line, sizeof line, "%s=%s\n", k, v.c_str()); s += line; };
    add("sealed", w->sealed ? "1" : "0");
    {
        uint32_t n_ready = 0, n_building = 0, n_failed = 0; std::string why;
        for (auto& t : w->spec_tab) {
            const int st = t.spec ? t.spec->state.load(std::memory_order_acquire) : 0;
            n_ready += st == 2; n_building += st == 1;
            if (st == 3) { ++n_failed; why = t.spec->why; }
        }
        std::string v = !w->knobs.jit_specialise_after ? "off (GGRS_JIT_SPECIALISE_AFTER=0)"
                      : n_ready ? "ready (" + std::to_string(n_ready) + " of " + std::to_string(w->spec_tab.size()) + " group shapes seen run on kernels compiled for them" +
                                  (n_building ? ", 1 building" : "") + (n_failed ? ", " + std::to_string(n_failed) + " failed: " + why : "") + ")"
                      : n_building ? "building" : n_failed ? "failed: " + why : "none yet";
        for (char& ch : v) if (ch == '\n') ch = ' ';
        add("specialised_kernel", v);
    }
    add("arena", !w->sealed ? "none" : (!w->own_arena ? "caller-provided" : "paged (hipMalloc)"));
    add("arena_bytes", std::to_string(w->arena_bytes));
    {
        Hiprtc& r = hiprtc_for(w);
        add("hiprtc", r.lib ? "loaded" : ("missing: " + r.why));
    }
    add("generated_kernel", w->jit_fn ? "ok" : w->jit_status);
    if (w->jit_fn) add("generated_kernel_origin", w->jit_origin);
    std::string k;
    const uint64_t cover = std::max(w->len, w->live.dirty_len);
    if (!w->sealed) k = "unknown (not sealed)";
    else if (w->gen_ok) k = "ggrs_jit_tick (generated for this world; one workgroup per 256 slots)";
    else k = "per-request kernels (k_copy_state, one launch per system)";
    add("request_group_kernel", k);
    if (w->sealed && w->gen_ok) {
        // who folds the per-workgroup checksum rows of a plain (no batch) request group of this size
        const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + 255) / 256));
        const bool ff = w->h_rows && !w->device_results_only && g > (uint32_t)w->knobs.fold_forward_min_wgs;
        const bool host = w->h_rows && !w->device_results_only && !ff;
        std::string f = ff ? "fold-forward: the next launch on the stream folds the rows (k_ff_fold when nothing follows) and the host hashes one value per Save and part; blocking calls: " +
                             std::string(g <= HOST_FOLD_MAX_WGS_BLOCKING && g <= (uint32_t)w->knobs.fold_forward_min_wgs ? "the host folds the rows" : "the launch folds its own rows (self-fold; k_gen_finalize for lists it does not cover)")
                      : host ? (g <= HOST_FOLD_MAX_WGS_BLOCKING ? "the host folds the rows at collect time (blocking calls too)" : "the host folds the rows at collect time (blocking calls: k_gen_finalize)")
                             : "k_gen_finalize";
        add("checksum_fold", f);
        add("kernarg_bytes", std::to_string(w->jl ? w->jl->bytes : 0));
        add("lazy_live_block", !lazy_live_possible(w) ? "off (live-only state, a handed-out pointer, results on the device, or per-request launches)"
                                   : cover <= JIT_NT_MIN_SLOTS ? "off (the world fits the caches: the live block's bytes are not what bounds the launch)"
                                   : "on after " + std::to_string(LAZY_LIVE_STREAK) + " lists in a row that open with a LoadGameState: " + std::to_string(w->lazy_skips) + " lists left it unwritten, " +
                                     std::to_string(w->lazy_materialised) + " materialised on demand");
        add("group_caps", std::to_string(w->cap_saves) + " saves / " + std::to_string(w->cap_steps) + " steps");
        bool any_spawn = false;
        for (auto& sd : w->systems) any_spawn |= sd.kind == GGRS_SYS_PARTICLES_SPAWN || sd.kind == GGRS_SYS_SPAWN_CUSTOM;
        if (w->dev_spawn) { char t[160]; snprintf(t, sizeof t, "one cooperative launch per request group: %u workgroups, %d resident per CU x %d CUs (%d VGPRs, %d SGPRs)", 8u * ((w->sp_tiles + 7u) / 8u), w->sp_per_cu, w->n_cu, w->sp_regs, w->sp_sregs); add("device_spawn", t); }
        if (any_spawn) add("spawn_system", w->jit_spawn_sys >= 0 ? "runs inside the request group (rows appended by the group's launch)" : "ends the request group (its own launches)");
    }
    add("blocking_wait", w->knobs.spin_wait_us > 0 ? "polls k_gen_finalize's completion tags when that kernel ends the list (" + std::to_string(w->spin_hits) + " calls so far, " +
                                                      std::to_string(w->spin_misses) + " fell back to the stream wait), else hipStreamSynchronize"
                                                    : "hipStreamSynchronize (GGRS_SPIN_WAIT_US=0)");
    add("slots_covered", std::to_string(cover));
    add("row_versions", w->knobs.row_versions ? "on" : "off (GGRS_ROW_VERSIONS=0)");
    add("value_tags", !w->sealed ? "unknown (not sealed)" : w->vtags ? "on: a Save skips the columns whose 64 values per unit the destination already holds (" + std::to_string(w->prof_skipped) + " bytes not stored in profiled launches; the ids' numbering has started over " + std::to_string(w->tag_wraps) + " times)"
                                 : "off (the world's steady Save moves less than " + std::to_string(VTAGS_MIN_BYTES >> 20) + " MB, or row versions are off)");
    if (needed) *needed = s.size() + 1;
    if (buf && cap) { const uint64_t n = std::min<uint64_t>(cap, s.size() + 1); memcpy(buf, s.c_str(), n); buf[n - 1] = 0; }
    return GGRS_OK;
}

int ggrs_hip_profile_enable(ggrs_world* w, int on) {
    if (!w) return GGRS_E_INVALID;
    w->prof = on != 0;
    if (on) { DeviceGuard dg(w);
              for (auto& e : w->prof_events) { w->prof_pool.push_back(e.a); w->prof_pool.push_back(e.b); } w->prof_events.clear();
              while (w->prof_pool.size() < 512) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) break; w->prof_pool.push_back(e); }     // (outside whatever the caller times)
              for (int i = 0; i < (int)GGRS_KERNEL_CLASSES; ++i) { w->prof_ms[i] = 0; w->prof_n[i] = 0; w->prof_bytes[i] = 0; w->prof_launch_us[i].clear(); }
              w->prof_skipped = 0; if (w->d_skip) { (void)hipStreamSynchronize(w->stream); (void)hipMemset(w->d_skip, 0, 8); } }
    return GGRS_OK;
}
// drains the recorded event pairs into the per-class totals and per-launch lists
static int profile_drain(ggrs_world* w) {
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (auto& e : w->prof_events) {
        float ms = 0; (void)hipEventElapsedTime(&ms, e.a, e.b);
        w->prof_ms[e.cls] += ms; w->prof_n[e.cls] += 1;
        if (w->prof_launch_us[e.cls].size() < 65536) w->prof_launch_us[e.cls].push_back(ms * 1e3f);
        if (w->prof_pool.size() < 4096) { w->prof_pool.push_back(e.a); w->prof_pool.push_back(e.b); } else { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
    w->prof_events.clear();
    // value tags: what the profiled launches did NOT store comes off the bytes they were asked to move
    if (w->d_skip) {
        uint64_t skipped = 0;
        HIPCHK(w, hipMemcpy(&skipped, w->d_skip, 8, hipMemcpyDeviceToHost));
        if (skipped) { HIPCHK(w, hipMemset(w->d_skip, 0, 8)); w->prof_bytes[GGRS_KERNEL_TICK] -= std::min<uint64_t>(skipped, w->prof_bytes[GGRS_KERNEL_TICK]); w->prof_skipped += skipped; }
    }
    return GGRS_OK;
}
int ggrs_hip_profile_read_launches(ggrs_world* w, uint32_t cls, float* us_out, uint32_t cap, uint32_t* n_out) {
    if (!w || cls >= GGRS_KERNEL_CLASSES || (!us_out && cap)) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = profile_drain(w); if (rc) return rc;
    const std::vector<float>& v = w->prof_launch_us[cls];
    const uint32_t n = (uint32_t)std::min<size_t>(v.size(), cap);
    if (n) memcpy(us_out, v.data(), (size_t)n * sizeof(float));
    if (n_out) *n_out = (uint32_t)v.size();
    return GGRS_OK;
}
int ggrs_hip_profile_read_bytes(ggrs_world* w, uint64_t* bytes_out) {
    if (!w || !bytes_out) return GGRS_E_INVALID;
    if (w->d_skip && w->sealed) { DeviceGuard dg(w); const int rc = profile_drain(w); if (rc) return rc; }     // value tags: what the launches did not store is known once they ran
    for (int i = 0; i < (int)GGRS_KERNEL_CLASSES; ++i) bytes_out[i] = w->prof_bytes[i];
    return GGRS_OK;
}
int ggrs_hip_profile_read(ggrs_world* w, double* ms_out, uint64_t* launches_out) {
    if (!w || !ms_out || !launches_out) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = profile_drain(w); if (rc) return rc;
    for (int i = 0; i < (int)GGRS_KERNEL_CLASSES; ++i) { ms_out[i] = w->prof_ms[i]; launches_out[i] = w->prof_n[i]; }
    return GGRS_OK;
}


#include "host_fanout.hpp"
