// ggrs_hip.hip -- host runtime + C ABI of libggrs_hip.so (see include/ggrs_hip.h).
//
// One ggrs_world owns: a device arena carved into identical *packed state blocks* (one live
// block + up to max_depth ring slots), the host-side ring bookkeeping (an exact mirror of
// GgrsSnapshots<_, _>, /root/reference/src/snapshot/mod.rs:121-243, over slot indices instead
// of HashMaps), the registered component/system tables and one HIP stream.  Every request
// becomes kernel launches on that stream; the only host<->device synchronisation in
// handle_requests is one checksum read-back at the end of the batch.
//
// There is NO CPU fallback: without a HIP device world creation fails with GGRS_E_NO_DEVICE.

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>   // types and prototypes only: resolved with dlsym on first use (no link-time dependency)
#include <rccl/rccl.h>      // types and prototypes only: every entry point is resolved with dlsym (no link-time dependency)
#include <dlfcn.h>
#include <math.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <cctype>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ggrs_hip.h"
#include "kernels.hpp"

using namespace ggrs;

namespace {

constexpr uint64_t ALIGN = 256;
constexpr uint64_t TICK_VEC1_MAX_SLOTS = 400 * 1024;   // worlds covering up to this many slots run on k_tick1 (see run_request_groups)
constexpr int TICK2_RESTL_MAX = 7;     // untouched rows k_tick2 / the straight-line k_tick3 keep in registers: EXACTLY this many (the stress_test world)
constexpr int TICK3_RESTL_ANY = 16;    // k_tick3's general instantiation: up to this many untouched 4-byte rows
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

struct Comp {
    std::string name;
    uint32_t word_bytes = 4, n_words = 0;
    std::vector<uint32_t> cks_words;     // as registered
    bool checksummed = false;
    std::vector<uint8_t> defaults;
    uint32_t col_base = 0;               // index of its first column
    bool no_rollback = false;            // GGRS_COMP_NO_ROLLBACK: lives in the side region, outside every snapshot
};

struct Block {                           // one packed state block in the arena
    uint8_t* ptr = nullptr;
    uint64_t dirty_len = 0;              // slots that may hold non-zero mask bits
    uint64_t len = 0;                    // host mirror of Header::len for ring slots
};

struct EventPair { hipEvent_t a, b; uint32_t cls; };

// Every environment variable the library reads, in ONE place, read ONCE per world at creation.  They are A/B and
// debugging aids for measurements (INTEGRATION.md lists them); none of them changes a result.
// (GGRS_HIP_TRACE / GGRS_HIP_ROCTX, the two tracing switches, are process-wide: see Tracer below.)
struct Knobs {
    bool tick_generic = false;     // GGRS_TICK_GENERIC=1   serve every world with the LDS-staged k_tick_gen
    bool tick_ntload = false;      // GGRS_TICK_NTLOAD=1    non-temporal loads of the group's source block
    int tick_vec = 0;              // GGRS_TICK_VEC=1|41|4  force a k_tick shape (0: by world size)
    uint32_t tick_lds = 0;         // GGRS_TICK_LDS=bytes   dynamic LDS per k_tick workgroup (occupancy throttle)
    bool tick_rest_loop = true;    // GGRS_TICK_REST=0      untouched rows fanned out up front instead of with each Save
    int arena_probe = 0;           // GGRS_ARENA_PROBE=n    time n candidate arena placements per round at world creation (default off)
    uint64_t arena_align = 0, arena_skew = 0;   // GGRS_ARENA_ALIGN / GGRS_ARENA_SKEW   placement of the first block inside the allocation
    uint64_t block_pad = 0, col_pad = 0;        // GGRS_BLOCK_PAD / GGRS_COL_PAD        extra bytes between ring blocks / behind the columns
    bool debug_arena = false;      // GGRS_DEBUG_ARENA=1    print the arena placement
    int arena_contig = -1;         // GGRS_ARENA_CONTIG=0|1 physically contiguous arena (hipExtMallocWithFlags + hipDeviceMallocContiguous) for every / no world;
                                   //                       default (-1): only worlds created with GGRS_WORLD_CONTIG_ARENA (k_tick3 worlds up to 1.5 GiB:
                                   //                       111.8 vs 114.5 us per tick at 1 M, slower beyond: 594 vs 480 us at 4 M) -- see the flag's
                                   //                       note in include/ggrs_hip.h for why it is not the default any more
    bool tick2 = true;             // GGRS_TICK2=0          big worlds on the round-1 k_tick + k_tick_finalize pair instead of k_tick2
    int tick2_wgs_per_cu = 2;      // GGRS_TICK2_WGS=n      persistent workgroups per CU of k_tick2 (0: one workgroup per tile, not persistent)
    int tick2_nt = 1;              // GGRS_TICK2_NT=0|1     non-temporal snapshot stores in k_tick2
    int tick1_dp = 1;              // GGRS_TICK1_DP=0       k_tick1 without the depth-parallel grid (one workgroup walks the whole group);
                                   //               =2..9   A/B: that many outputs per role, up to GGRS_TICK1_DP_MAX_SLOTS2 slots
    uint64_t tick1_dp_max_slots = 40 * 1024;    // GGRS_TICK1_DP_MAX_SLOTS   largest world that uses one output per role (x2: two, x4: three)
    uint64_t tick1_dp_max_slots2 = 400 * 1024;
    bool debug_poison = false;     // GGRS_DEBUG_POISON=1   fill fresh arenas / scratch with 0xA5 (uninitialised-read hunting)
    int debug_jit = 0;             // GGRS_DEBUG_JIT=1      say why a generated kernel was rejected; =2 also print its source
    uint64_t jit_particles_max_slots = 416 * 1024;   // GGRS_JIT_PARTICLES_MAX_SLOTS  particles worlds up to this size run on the generated kernel (0: never)
    int host_fold_max_wgs = 256;   // GGRS_HOST_FOLD_MAX_WGS=n   generated kernel: groups of up to n workgroups leave their partial rows in pinned memory and the host folds them (0: always k_gen_finalize)
    int jit_v = 0;                 // GGRS_JIT_V=1|4        A/B: slots per lane of the generated kernel (0: by world size)
    bool tick_jit = true;          // GGRS_TICK_JIT=0       no run-time generated request-group kernel (k_tick_gen / per-request instead)
    int gen_sub = 0;               // GGRS_GEN_SUB=256|512|1024   A/B: slots per k_tick_gen workgroup (0: by world size)
    int gen_dp = 1;                // GGRS_GEN_DP=0         k_tick_gen without depth-parallel roles; =2..9  A/B: that many outputs per role
    uint64_t gen_dp_max_slots = 160 * 1024;     // GGRS_GEN_DP_MAX_SLOTS     largest world the A/B setting applies to
    bool dead_groups = true;       // GGRS_DEAD_GROUPS=0    no dead-snapshot elimination / branch batching (every group stores everything)
    int tick2_ilv = 0;             // GGRS_TICK2_ILV=0|1    Save = store burst + hash (0) or stores spaced out between the hash multiplies (1)
    uint64_t tick2_min_slots = 416 * 1024;   // GGRS_TICK2_MIN_SLOTS  worlds covering more slots than this run on k_tick3 / k_tick2 (profiles/r02jit/cross.txt)
    int tick3 = 2;                 // GGRS_TICK3=0|1|2      wave-specialised k_tick3 (1: workgroup barrier per hand-off, 2: per-pair LDS flags); 0: k_tick2
    static Knobs from_env() {
        Knobs k;
        auto num = [](const char* n, long long dflt) { const char* v = getenv(n); return v ? atoll(v) : dflt; };
        k.tick_generic = num("GGRS_TICK_GENERIC", 0) != 0;
        k.tick_ntload = num("GGRS_TICK_NTLOAD", 0) != 0;
        { const long long x = num("GGRS_TICK_VEC", 0); if (x == 1 || x == 4 || x == 41) k.tick_vec = (int)x; }
        { const long long x = num("GGRS_TICK_LDS", 0); if (x >= 0 && x <= 160 * 1024) k.tick_lds = (uint32_t)x; }
        k.tick_rest_loop = num("GGRS_TICK_REST", 1) != 0;
        k.arena_probe = (int)std::max<long long>(0, num("GGRS_ARENA_PROBE", 0));
        k.arena_align = (uint64_t)std::max<long long>(0, num("GGRS_ARENA_ALIGN", 0));
        k.arena_skew = (uint64_t)std::max<long long>(0, num("GGRS_ARENA_SKEW", 0));
        k.block_pad = (uint64_t)std::max<long long>(0, num("GGRS_BLOCK_PAD", 0));
        k.col_pad = (uint64_t)std::max<long long>(0, num("GGRS_COL_PAD", 0));
        k.debug_arena = num("GGRS_DEBUG_ARENA", 0) != 0;
        k.arena_contig = (int)std::min<long long>(1, std::max<long long>(-1, num("GGRS_ARENA_CONTIG", -1)));
        k.tick2 = num("GGRS_TICK2", 1) != 0;
        k.tick2_wgs_per_cu = (int)std::min<long long>(8, std::max<long long>(0, num("GGRS_TICK2_WGS", 2)));
        k.tick2_nt = num("GGRS_TICK2_NT", 1) != 0;
        k.tick2_ilv = num("GGRS_TICK2_ILV", 0) != 0;
        k.dead_groups = num("GGRS_DEAD_GROUPS", 1) != 0;
        { const long long v = num("GGRS_GEN_SUB", 0); k.gen_sub = (v == 256 || v == 512 || v == 1024) ? (int)v : 0; }
        k.tick_jit = num("GGRS_TICK_JIT", 1) != 0;
        k.debug_jit = (int)num("GGRS_DEBUG_JIT", 0);
        k.host_fold_max_wgs = (int)std::max<long long>(0, std::min<long long>(1 << 20, num("GGRS_HOST_FOLD_MAX_WGS", 256)));
        k.debug_poison = num("GGRS_DEBUG_POISON", 0) != 0;
        k.jit_particles_max_slots = (uint64_t)std::max<long long>(0, num("GGRS_JIT_PARTICLES_MAX_SLOTS", 416 * 1024));
        { const long long v = num("GGRS_JIT_V", 0); k.jit_v = (v == 1 || v == 4) ? (int)v : 0; }
        k.gen_dp = (int)std::min<long long>(9, std::max<long long>(0, num("GGRS_GEN_DP", 1)));
        k.gen_dp_max_slots = (uint64_t)std::max<long long>(0, num("GGRS_GEN_DP_MAX_SLOTS", 160 * 1024));
        k.tick1_dp = (int)std::min<long long>(9, std::max<long long>(0, num("GGRS_TICK1_DP", 1)));
        k.tick1_dp_max_slots = (uint64_t)std::max<long long>(0, num("GGRS_TICK1_DP_MAX_SLOTS", 40 * 1024));
        k.tick1_dp_max_slots2 = (uint64_t)std::max<long long>(0, num("GGRS_TICK1_DP_MAX_SLOTS2", 400 * 1024));
        k.tick3 = (int)std::min<long long>(2, std::max<long long>(0, num("GGRS_TICK3", 2)));
        k.tick2_min_slots = (uint64_t)std::max<long long>(0, num("GGRS_TICK2_MIN_SLOTS", 416 * 1024));
        return k;
    }
};

}  // namespace

static std::atomic<int> g_paged_arena_frees{0};   // paged (cached) arenas this process has handed back: see GGRS_WORLD_CONTIG_ARENA

struct ggrs_world {
    // ---- configuration
    int device = 0;
    uint64_t capacity = 0, cap_pad = 0;
    uint32_t max_depth = 0, flags = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    uint8_t* arena = nullptr; uint64_t arena_bytes = 0; bool own_arena = false, arena_contiguous = false;
    uint8_t* arena_alloc = nullptr;      // what hipMalloc returned (arena may be aligned / skewed inside it)

    std::vector<Comp> comps;
    std::vector<ggrs_system_desc> systems;
    struct Custom {                      // GGRS_SYS_CUSTOM: a hiprtc-compiled per-entity system (systems[i].comp[0] indexes this)
        std::string name, source;
        hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
        uint32_t n_bind = 0, comp[GGRS_CUSTOM_MAX_BINDINGS] = {}, word[GGRS_CUSTOM_MAX_BINDINGS] = {};
        uint32_t n_pres = 0, pres_comp[GGRS_CUSTOM_MAX_BINDINGS] = {};
    };
    std::vector<Custom> customs;
    hipFunction_t jit_fn = nullptr;      // the request-group kernel generated for this world (jit_source), or null
    hipFunction_t jit_fn4 = nullptr;     // its 4-slots-per-lane form (worlds that can grow past JIT_V4_MIN_SLOTS)
    std::string jit_status = "not attempted";   // why the world has / has not a generated kernel (ggrs_hip_world_kernel_info)
    bool jit_marks = false;
    bool jit_reads_inputs = false;       // a system reads PlayerInputs (BOX_MOVE, custom): branches with different inputs differ
    uint32_t gen_parts_saves = 0;        // Save rows of d_gen_parts (room for a batch of checksum-only groups in small worlds)
    bool layout_only = false;            // GGRS_WORLD_LAYOUT_ONLY: no device behind this world
    bool sealed = false;
    int seal_error = 0;                  // a failed seal latches: every later call reports it instead of re-carving the arena
    Knobs knobs;
    std::string err;

    // ---- layout of a packed state block.  Offsets of non-rollback components and of the
    // RollbackDespawned markers are ALSO relative to the live block's base but point past the ring,
    // into the live-only side region (they are only ever applied to the live block).
    uint64_t side_off = 0, side_bytes = 0;
    DespawnMarks marks{};                // disabled mask + despawned-frame column (despawn.rs:45-46)
    bool has_nr = false;                 // any GGRS_COMP_NO_ROLLBACK component
    bool marks_possible = false;         // a RollbackDespawned marker may exist in the live world
    int32_t dc_local = 0;                // Local<ConfirmedFrameCount> of despawn_confirmed_entities (despawn.rs:92)
    uint64_t state_bytes = 0, off_alive = 0;
    std::vector<uint64_t> off_present, col_off;   // col_off: block-relative offset of the column's row in tile 0
    std::vector<uint32_t> col_wb, col_ts;          // word bytes / tile stride of every column (kernels.hpp col_at)
    uint32_t ts = 0;                               // tile stride of the rollback word columns: bytes of all their words x 1024 slots
    CopyPlan plan{};

    // ---- device buffers
    Block live;
    std::vector<Block> slots;            // ring slot pool
    std::vector<int> free_slots;
    uint64_t* d_parts = nullptr; uint32_t part_stride = 0;   // [(n_cks)+1][part_stride], last = counts
    uint64_t* d_results = nullptr; uint64_t* h_results = nullptr; uint32_t max_results = 0;
    UnitDesc* d_units = nullptr;
    uint64_t* d_maskoffs = nullptr;      // scratch for k_set_mask_range
    float* d_stage = nullptr; float* h_stage = nullptr; uint64_t stage_floats = 0, stage_used = 0;

    // ---- checksum specs (device view)
    std::vector<uint32_t> cks_comp;      // checksummed component ids in id order
    CksArgs cks_args{};
    bool fused_ok = false;               // schedule == particles fast path
    bool fused_cks = false;              // ... and every checksum spec is covered by it
    int f_T = -1, f_V = -1, f_L = -1, f_spawn = -1; uint32_t f_tw = 0, f_vw = 0;
    bool f_cksT = false, f_cksV = false;
    float f_g[3] = {0, 0, 0};
    // fused request groups (k_tick): the schedule is exactly the particles systems over three
    // distinct components and every checksum spec is one the kernel computes in registers
    bool tick_ok = false; uint32_t f_lw = 0;
    TickArgs tick_proto{};               // layout part of the kernel arguments, filled at seal
    uint64_t* d_tick_parts = nullptr; uint32_t tick_part_stride = 0;
    uint32_t tick_parts_saves = MAX_TICK_SAVES;   // Save slots of d_tick_parts: > MAX_TICK_SAVES lets small worlds batch identical checksum-only groups
    // k_tick2: persistent grid + in-kernel fold (big worlds)
    bool tick2_ok = false; Tick2Args tick2_proto{};
    uint64_t* d_wg_parts = nullptr; uint32_t* d_ticket = nullptr; int n_cu = 256;
    // generic fused request groups (k_tick_gen): any mix of the supported kernel systems, state staged in LDS
    bool gen_ok = false;
    GenArgs gen_proto{};                 // layout part of the kernel arguments, filled at seal
    uint32_t gen_sub_max = 0;            // largest slots-per-workgroup whose LDS image fits 64 KiB
    GenWord* d_gen_words = nullptr; GenUnit* d_gen_units = nullptr;
    uint64_t* d_gen_parts = nullptr;     // [MAX_TICK_SAVES][n_cks + 1][tick_part_stride]
    int gen_box_sys = -1;                // index of a BOX_MOVE system (its FRICTION.powf(dt) is evaluated per step on the host)
    uint64_t block_pad = 0, col_pad = 0; // extra bytes between ring blocks / columns (Knobs; library-owned arenas only)

    // pending partials produced by the last advance (valid for the live state as-is)
    bool pending_valid = false; uint32_t pending_parts = 0;

    // ---- host mirrors
    uint64_t len = 0;
    int32_t frame = 0;
    bool has_confirmed = true; int32_t confirmed = 0;   // init_resource::<ConfirmedFrameCount>() == 0 (mod.rs:336)
    uint64_t fps = 60;
    int32_t synctest_cd = -1;
    size_t depth = 60;                                   // DEFAULT_FPS until sync_depth (mod.rs:115)
    std::deque<int> ring_slot; std::deque<int32_t> ring_frame;   // newest at the front

    // ---- asynchronous request batches (ggrs_hip_enqueue_requests / ggrs_hip_collect_checksums)
    struct PendingBatch { hipEvent_t ev; uint32_t first, count; std::vector<uint64_t> host; uint32_t n_folds = 0; };
    // Host-side checksum fold of small worlds (generated kernel): its workgroups write their partial rows straight into pinned,
    // device-mapped host memory and the HOST finishes each Save (XOR of g rows + three hashes) when the batch is collected -- a
    // second launch (k_gen_finalize + its dependent-launch gap, ~7 us) costs more than that for worlds of a few hundred workgroups.
    struct HostFold { uint32_t res_slot, n_saves, g, n_cks, members; uint64_t rows_off, total_len; };
    std::deque<HostFold> folds;          // in submission order; a PendingBatch owns the next n_folds of them
    uint64_t* h_rows = nullptr; uint64_t* d_rows = nullptr; uint64_t rows_cap = 0, rows_used = 0, rows_tail = 0;   // ring of partial rows
    bool device_results_only = false;    // a consumer reads the result ring in stream order (ggrs_hip_fanout_*): every fold stays on the device
    std::deque<PendingBatch> pending; uint32_t res_head = 0; uint32_t pending_results = 0;
    std::vector<hipEvent_t> event_pool;

    // ---- profiling
    bool nt_copy = false;               // non-temporal loads/stores in k_copy_state (A/B knob)
    bool prof = false;
    std::vector<EventPair> prof_events;
    double prof_ms[GGRS_KERNEL_CLASSES] = {};
    uint64_t prof_n[GGRS_KERNEL_CLASSES] = {};
    std::vector<float> prof_launch_us[GGRS_KERNEL_CLASSES];   // every launch since enable, in submission order (ggrs_hip_profile_read_launches)

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};

#define HIPCHK(w, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return (w)->fail(GGRS_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                              \
    } while (0)

namespace {

struct ProfScope {
    ggrs_world* w; uint32_t cls; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(ggrs_world* w_, uint32_t c) : w(w_), cls(c) {
        if (w->prof) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, w->stream); }
    }
    ~ProfScope() {
        if (w->prof) { (void)hipEventRecord(b, w->stream); w->prof_events.push_back({a, b, cls}); }
    }
};

#include "kernel_gen.hpp"   // hiprtc plumbing, the custom-system ABI text, the per-world request-group kernel generator

// Computes the packed state layout from the registered components.
void build_layout(ggrs_world* w) {
    const uint64_t mask_bytes = align_up(w->cap_pad / 8, ALIGN);
    uint64_t off = ALIGN;                          // header
    w->off_alive = off; off += mask_bytes;
    w->off_present.assign(w->comps.size(), 0); w->col_off.clear(); w->col_wb.clear();
    w->has_nr = false;
    uint32_t ncols = 0;
    for (auto& c : w->comps) { c.col_base = ncols; ncols += c.n_words; w->has_nr |= c.no_rollback; }
    w->col_off.assign(ncols, 0); w->col_wb.assign(ncols, 4);
    for (size_t c = 0; c < w->comps.size(); ++c) if (!w->comps[c].no_rollback) { w->off_present[c] = off; off += mask_bytes; }
    // rollback word columns, TILE-MAJOR: tile t of every column is contiguous (ts bytes per tile).  Inside a
    // tile the words that GgrsSchedule systems read or write come first, so a per-request AdvanceWorld kernel
    // streams one contiguous span per tile (particles: 32 of the 60 KiB) instead of 4 KiB pieces.
    w->col_ts.assign(ncols, 0);
    std::vector<uint8_t> hot(ncols, 0);
    for (auto& sd : w->systems) {
        auto mark = [&](uint32_t comp, uint32_t word, uint32_t span) {
            if (comp >= w->comps.size()) return;
            for (uint32_t k = 0; k < span && word + k < w->comps[comp].n_words; ++k) hot[w->comps[comp].col_base + word + k] = 1;
        };
        switch (sd.kind) {
        case GGRS_SYS_PARTICLES_UPDATE: mark(sd.comp[0], sd.word[0], 3); mark(sd.comp[1], sd.word[1], 3); break;
        case GGRS_SYS_TTL_DESPAWN: case GGRS_SYS_ADD_U32: case GGRS_SYS_SAT_SUB_DESPAWN: mark(sd.comp[0], sd.word[0], 1); break;
        case GGRS_SYS_BOX_MOVE: mark(sd.comp[0], sd.word[0], 3); mark(sd.comp[1], sd.word[1], 3); mark(sd.comp[2], sd.word[2], 1); break;
        default: break;
        }
    }
    uint64_t tcol = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (auto& c : w->comps) {
            for (uint32_t k = 0; k < c.n_words; ++k) {
                w->col_wb[c.col_base + k] = c.word_bytes;
                if (c.no_rollback || (hot[c.col_base + k] != 0) != (pass == 0)) continue;
                w->col_off[c.col_base + k] = tcol;           // offset inside a tile for now
                tcol += (uint64_t)LAYOUT_TILE * c.word_bytes;
            }
        }
    w->ts = (uint32_t)tcol;
    const uint64_t cols_base = align_up(off, 4096);
    for (auto& c : w->comps) if (!c.no_rollback)
        for (uint32_t k = 0; k < c.n_words; ++k) { w->col_off[c.col_base + k] += cols_base; w->col_ts[c.col_base + k] = w->ts; }
    off = cols_base + (w->cap_pad / LAYOUT_TILE) * (uint64_t)w->ts + w->col_pad;
    w->state_bytes = align_up(off, 4096) + w->block_pad;
    // ---- live-only side region, placed right behind the ring blocks
    w->side_off = (uint64_t)(w->max_depth + 1) * w->state_bytes;
    uint64_t so = w->side_off;
    w->marks.off_disabled = so; so += mask_bytes;
    w->marks.off_dframe = so; so += align_up(w->cap_pad * 4, ALIGN);
    for (size_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback) { w->off_present[c] = so; so += mask_bytes; }
    for (auto& c : w->comps) {
        if (!c.no_rollback) continue;
        // live-only columns are plain arrays: the same addressing formula with tile stride = 1024 words
        for (uint32_t k = 0; k < c.n_words; ++k) { w->col_off[c.col_base + k] = so; w->col_ts[c.col_base + k] = LAYOUT_TILE * c.word_bytes; so += align_up(w->cap_pad * c.word_bytes, ALIGN); }
    }
    w->side_bytes = align_up(so - w->side_off, 4096);

    CopyPlan& p = w->plan;
    memset(&p, 0, sizeof p);
    p.n_masks = 1;
    p.mask_off[0] = w->off_alive;
    uint32_t nr = 0;
    for (size_t c = 0; c < w->comps.size(); ++c) {
        const Comp& cc = w->comps[c];
        if (cc.no_rollback) continue;                       // snapshots hold rollback components only
        p.mask_off[p.n_masks++] = w->off_present[c];
        for (uint32_t k = 0; k < cc.n_words; ++k)
            for (uint32_t r = 0; r < cc.word_bytes / 4; ++r) {
                RowDesc& rd = p.row[nr++];
                rd.col_off = w->col_off[cc.col_base + k]; rd.roff = r * 4096; rd.tile_stride = w->ts; rd.word_bytes = cc.word_bytes; rd.pad = 0;
            }
    }
    p.n_rows = nr;
}

uint32_t total_rows(const ggrs_world* w) {
    uint32_t n = 0;
    for (auto& c : w->comps) if (!c.no_rollback) n += c.n_words * (c.word_bytes / 4);
    return n;
}

// times k_tick on a candidate arena placement (defined next to the launchers)
int probe_arena_placement(ggrs_world* w, uint8_t* base, uint64_t tick_parts_off, float* us_out);

int seal_impl(ggrs_world* w);
// Sealing fixes the layout and carves the arena, lazily, on the first call that needs device state.  It is
// failure-atomic: whatever a failed attempt allocated is released, and the failure LATCHES -- every later call
// reports the same error instead of carving a second arena over half-initialised bookkeeping.
int seal(ggrs_world* w) {
    if (w->layout_only) return w->fail(GGRS_E_NO_DEVICE, "GGRS_WORLD_LAYOUT_ONLY world: there is no device behind it");
    if (w->sealed) return GGRS_OK;
    if (w->seal_error) return w->seal_error;
    const int rc = seal_impl(w);
    if (rc == GGRS_OK) return rc;
    const std::string why = w->err;
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    if (w->d_gen_words) { (void)hipFree(w->d_gen_words); w->d_gen_words = nullptr; }
    if (w->d_gen_units) { (void)hipFree(w->d_gen_units); w->d_gen_units = nullptr; }
    if (w->d_gen_parts) { (void)hipFree(w->d_gen_parts); w->d_gen_parts = nullptr; }
    if (w->h_results) { (void)hipHostFree(w->h_results); w->h_results = nullptr; w->d_results = nullptr; }
    if (w->h_stage) { (void)hipHostFree(w->h_stage); w->h_stage = nullptr; }
    if (w->h_rows) { (void)hipHostFree(w->h_rows); w->h_rows = nullptr; w->d_rows = nullptr; }
    if (w->own_arena && w->arena_alloc) { (void)hipFree(w->arena_alloc); if (!w->arena_contiguous) g_paged_arena_frees.fetch_add(1, std::memory_order_relaxed); w->arena_alloc = nullptr; w->arena = nullptr; w->arena_bytes = 0; w->own_arena = false; }
    (void)hipGetLastError();
    w->slots.clear(); w->free_slots.clear(); w->live = Block{};
    w->sealed = false; w->seal_error = rc;
    w->err = "world could not be sealed (permanent): " + why;
    return rc;
}
int seal_impl(ggrs_world* w) {
    if (total_rows(w) > (uint32_t)MAX_ROWS) return w->fail(GGRS_E_INVALID, "too many registered words (%u rows > %d)", total_rows(w), MAX_ROWS);
    // The particles kernels address their columns with the tile stride of the ROLLBACK columns; a live-only
    // (GGRS_COMP_NO_ROLLBACK) column is a plain array with a different stride.  The flag is set after registration
    // (register_component_ex), so the check lives here rather than in add_system.
    for (auto& sd : w->systems) {
        if (sd.kind != GGRS_SYS_PARTICLES_UPDATE && sd.kind != GGRS_SYS_TTL_DESPAWN && sd.kind != GGRS_SYS_PARTICLES_SPAWN) continue;
        const uint32_t nc = sd.kind == GGRS_SYS_PARTICLES_UPDATE ? 2u : (sd.kind == GGRS_SYS_TTL_DESPAWN ? 1u : 3u);
        for (uint32_t k = 0; k < nc; ++k)
            if (sd.comp[k] >= w->comps.size() || w->comps[sd.comp[k]].no_rollback)
                return w->fail(GGRS_E_INVALID, "system %u runs over component %u, which is not registered for rollback (GGRS_COMP_NO_ROLLBACK): unsupported", sd.kind, sd.comp[k]);
    }
    HIPCHK(w, hipSetDevice(w->device));
    build_layout(w);

    // ---- recognise the particles fast path: [PARTICLES_UPDATE, TTL_DESPAWN] (+ optional SPAWN)
    w->fused_ok = false; w->f_spawn = -1;
    {
        int upd = -1, ttl = -1, other = 0;
        for (size_t i = 0; i < w->systems.size(); ++i) {
            switch (w->systems[i].kind) {
            case GGRS_SYS_PARTICLES_UPDATE: if (upd < 0) upd = (int)i; else ++other; break;
            case GGRS_SYS_TTL_DESPAWN: if (ttl < 0) ttl = (int)i; else ++other; break;
            case GGRS_SYS_PARTICLES_SPAWN: w->f_spawn = (int)i; break;
            default: ++other;
            }
        }
        if (upd >= 0 && ttl >= 0 && other == 0 && !(w->flags & GGRS_WORLD_UNFUSED)) {
            const ggrs_system_desc& u = w->systems[upd]; const ggrs_system_desc& l = w->systems[ttl];
            const Comp& T = w->comps[u.comp[0]]; const Comp& V = w->comps[u.comp[1]]; const Comp& L = w->comps[l.comp[0]];
            if (T.word_bytes == 4 && V.word_bytes == 4 && L.word_bytes == 8 && u.word[0] + 3 <= T.n_words && u.word[1] + 3 <= V.n_words &&
                !T.no_rollback && !V.no_rollback && !L.no_rollback) {
                w->fused_ok = true;
                w->f_T = (int)u.comp[0]; w->f_V = (int)u.comp[1]; w->f_L = (int)l.comp[0];
                w->f_tw = u.word[0]; w->f_vw = u.word[1];
                for (int k = 0; k < 3; ++k) w->f_g[k] = u.fparam[k];
            }
        }
    }
    // ---- checksum specs
    w->cks_comp.clear();
    std::vector<UnitDesc> units;
    memset(&w->cks_args, 0, sizeof w->cks_args);
    for (uint32_t c = 0; c < w->comps.size(); ++c) {
        Comp& cc = w->comps[c];
        if (!cc.checksummed) continue;
        const uint32_t k = (uint32_t)w->cks_comp.size();
        w->cks_comp.push_back(c);
        w->cks_args.off_present[k] = w->off_present[c];
        w->cks_args.unit_base[k] = (uint32_t)units.size();
        for (uint32_t wi : cc.cks_words) {
            const uint64_t co = w->col_off[cc.col_base + wi];
            const uint32_t cts = w->col_ts[cc.col_base + wi];
            if (cc.word_bytes == 4) units.push_back({co, 4, cts});
            else { units.push_back({co, 8, cts}); units.push_back({co + 4, 8, cts}); }
        }
        w->cks_args.n_units[k] = (uint32_t)units.size() - w->cks_args.unit_base[k];
        if (w->cks_args.n_units[k] > (uint32_t)MAX_UNITS) return w->fail(GGRS_E_INVALID, "checksum spec too long");
    }
    w->cks_args.n_cks = (uint32_t)w->cks_comp.size();
    w->cks_args.off_alive = w->off_alive;
    // does the fused step cover every spec?
    w->fused_cks = false; w->f_cksT = w->f_cksV = false;
    if (w->fused_ok) {
        bool all = true;
        for (uint32_t c : w->cks_comp) {
            const Comp& cc = w->comps[c];
            const uint32_t base = ((int)c == w->f_T) ? w->f_tw : w->f_vw;
            const bool is3 = cc.cks_words.size() == 3 && cc.cks_words[0] == base && cc.cks_words[1] == base + 1 && cc.cks_words[2] == base + 2;
            if ((int)c == w->f_T && is3 && w->f_T != w->f_V) w->f_cksT = true;
            else if ((int)c == w->f_V && is3 && w->f_T != w->f_V) w->f_cksV = true;
            else all = false;
        }
        w->fused_cks = all;
        if (!all) w->f_cksT = w->f_cksV = false;
    }

    // ---- fused request groups
    w->tick_ok = false;
    const bool force_generic = w->knobs.tick_generic;
    if (w->fused_ok && !force_generic && !(w->flags & GGRS_WORLD_NO_GROUPS) && w->f_T != w->f_V && w->f_T != w->f_L && w->f_V != w->f_L &&
        (w->fused_cks || w->cks_comp.empty())) {
        for (auto& sd : w->systems) if (sd.kind == GGRS_SYS_TTL_DESPAWN) w->f_lw = sd.word[0];
        TickArgs& a = w->tick_proto;
        memset(&a, 0, sizeof a);
        const Comp& T = w->comps[w->f_T]; const Comp& V = w->comps[w->f_V]; const Comp& L = w->comps[w->f_L];
        a.off_alive = w->off_alive;
        a.off_pT = w->off_present[w->f_T]; a.off_pV = w->off_present[w->f_V]; a.off_pL = w->off_present[w->f_L];
        for (int k = 0; k < 3; ++k) {
            a.off_t[k] = w->col_off[T.col_base + w->f_tw + k];
            a.off_v[k] = w->col_off[V.col_base + w->f_vw + k];
            a.g[k] = w->f_g[k];
        }
        a.off_ttl = w->col_off[L.col_base + w->f_lw];
        a.ts = w->ts;
        a.nt_load = w->knobs.tick_ntload ? 1u : 0u;
        for (uint32_t c = 0; c < w->comps.size(); ++c)
            if ((int)c != w->f_T && (int)c != w->f_V && (int)c != w->f_L && !w->comps[c].no_rollback) a.rest_mask_off[a.n_rest_masks++] = w->off_present[c];
        for (uint32_t c = 0; c < w->comps.size(); ++c) {
            const Comp& cc = w->comps[c];
            if (cc.no_rollback) continue;
            for (uint32_t k2 = 0; k2 < cc.n_words; ++k2) {
                const uint64_t co = w->col_off[cc.col_base + k2];
                bool owned = (co == a.off_ttl);
                for (int j = 0; j < 3; ++j) owned |= (co == a.off_t[j]) || (co == a.off_v[j]);
                if (owned) continue;
                for (uint32_t r = 0; r < cc.word_bytes / 4; ++r) a.rest[a.n_rest_rows++] = RowLite{co, r * 4096, w->ts, cc.word_bytes, 0};
            }
        }
        w->tick_ok = true;
    }
    // k_tick2 keeps the untouched rows in registers and addresses them as one contiguous run behind the schedule-owned rows
    w->tick2_ok = false;
    if (w->tick_ok && w->knobs.tick2 && (w->tick_proto.n_rest_rows == (uint32_t)TICK2_RESTL_MAX || (w->knobs.tick3 && w->tick_proto.n_rest_rows <= (uint32_t)TICK3_RESTL_ANY))) {
        const TickArgs& t = w->tick_proto;
        const uint64_t base = t.n_rest_rows ? t.rest[0].col_off + t.rest[0].roff : 0;
        bool contig = true;
        for (uint32_t j = 0; j < t.n_rest_rows; ++j) contig &= (t.rest[j].word_bytes == 4 && t.rest[j].col_off == base + (uint64_t)j * REST_ROW_STRIDE);
        if (contig) {
            Tick2Args& b = w->tick2_proto;
            memset(&b, 0, sizeof b);
            b.off_alive = t.off_alive; b.off_pT = t.off_pT; b.off_pV = t.off_pV; b.off_pL = t.off_pL;
            for (int k = 0; k < 3; ++k) { b.off_t[k] = t.off_t[k]; b.off_v[k] = t.off_v[k]; b.g[k] = t.g[k]; }
            b.off_ttl = t.off_ttl; b.rest_off = base; b.ts = t.ts;
            b.n_rest_rows = t.n_rest_rows; b.n_rest_masks = t.n_rest_masks;
            for (uint32_t m = 0; m < t.n_rest_masks; ++m) b.rest_mask_off[m] = t.rest_mask_off[m];
            b.fold.n_comp = 2; b.fold.comp_mask = (w->f_cksT ? 1u : 0u) | (w->f_cksV ? 2u : 0u);
            w->tick2_ok = true;
        }
    }


    // ---- generic fused request groups (k_tick_gen): every world whose systems it implements and whose words fit in LDS
    w->gen_ok = false; w->gen_box_sys = -1;
    std::vector<GenWord> gwords; std::vector<GenUnit> gunits;
    if ((!w->tick_ok || w->knobs.tick_jit) && !(w->flags & (GGRS_WORLD_NO_GROUPS | GGRS_WORLD_UNFUSED)) && w->ts > 0 && w->systems.size() <= (size_t)GEN_MAX_SYS &&
        w->cks_comp.size() <= (size_t)GEN_MAX_CKS) {
        GenArgs& a = w->gen_proto;
        memset(&a, 0, sizeof a);
        bool needs_jit = false;
        const uint32_t bps = w->ts / LAYOUT_TILE;                            // bytes per slot of all rollback words
        bool ok = true;
        const uint64_t cols_base = w->plan.n_rows ? w->plan.row[0].col_off : 0;   // every rollback column: cols_base + tcol
        uint64_t min_off = ~0ULL;
        for (uint32_t c = 0; c < w->comps.size(); ++c) if (!w->comps[c].no_rollback)
            for (uint32_t k = 0; k < w->comps[c].n_words; ++k) min_off = std::min(min_off, w->col_off[w->comps[c].col_base + k]);
        a.cols_base = min_off == ~0ULL ? cols_base : min_off;
        auto pso_of = [&](uint32_t col) { return (uint32_t)((w->col_off[col] - a.cols_base) / LAYOUT_TILE); };
        auto mask_index = [&](uint32_t comp) -> uint32_t {                    // index into plan.mask_off (0 = liveness)
            for (uint32_t m = 1; m < w->plan.n_masks; ++m) if (w->plan.mask_off[m] == w->off_present[comp]) return m;
            return ~0u;
        };
        for (uint32_t c = 0; c < w->comps.size(); ++c) {
            const Comp& cc = w->comps[c];
            if (cc.no_rollback) continue;
            for (uint32_t k = 0; k < cc.n_words; ++k) gwords.push_back({(uint32_t)(w->col_off[cc.col_base + k] - a.cols_base), cc.word_bytes, pso_of(cc.col_base + k), 0});
        }
        a.ts = w->ts; a.n_words = (uint32_t)gwords.size(); a.n_masks = w->plan.n_masks;
        for (uint32_t m = 0; m < w->plan.n_masks; ++m) a.mask_off[m] = w->plan.mask_off[m];
        // checksum specs
        a.n_cks = (uint32_t)w->cks_comp.size();
        for (uint32_t k = 0; k < a.n_cks && ok; ++k) {
            const Comp& cc = w->comps[w->cks_comp[k]];
            if (cc.no_rollback) { ok = false; break; }
            a.cks_pmask[k] = mask_index(w->cks_comp[k]);
            a.cks_unit_base[k] = (uint32_t)gunits.size();
            for (uint32_t wi : cc.cks_words) {
                const uint32_t pso = pso_of(cc.col_base + wi);
                gunits.push_back({pso, 0, cc.word_bytes, 0});
                if (cc.word_bytes == 8) gunits.push_back({pso, 4, 8, 0});
            }
            a.cks_n_units[k] = (uint32_t)gunits.size() - a.cks_unit_base[k];
        }
        // systems
        a.n_sys = 0;
        for (size_t i = 0; i < w->systems.size() && ok; ++i) {
            const ggrs_system_desc& d = w->systems[i];
            if (d.kind == GGRS_SYS_PARTICLES_SPAWN) continue;                 // a firing spawn system ends the group (Commands flush)
            GenSys& y = a.sys[a.n_sys];
            memset(&y, 0, sizeof y);
            y.kind = d.kind; y.iparam[0] = d.iparam[0]; y.iparam[1] = d.iparam[1];
            for (int k = 0; k < 4; ++k) y.fparam[k] = d.fparam[k];
            y.pmask[0] = y.pmask[1] = y.pmask[2] = ~0u;
            auto rb = [&](uint32_t comp) { return comp < w->comps.size() && !w->comps[comp].no_rollback; };
            switch (d.kind) {
            case GGRS_SYS_PARTICLES_UPDATE: case GGRS_SYS_BOX_MOVE:
                if (!rb(d.comp[0]) || !rb(d.comp[1])) { ok = false; break; }
                for (uint32_t k = 0; k < 3; ++k) {
                    y.pso[k] = pso_of(w->comps[d.comp[0]].col_base + d.word[0] + k);
                    y.pso[3 + k] = pso_of(w->comps[d.comp[1]].col_base + d.word[1] + k);
                }
                y.pmask[0] = mask_index(d.comp[0]); y.pmask[1] = mask_index(d.comp[1]);
                if (d.kind == GGRS_SYS_BOX_MOVE) {
                    const uint32_t hc = w->comps[d.comp[2]].col_base + d.word[2];
                    if (rb(d.comp[2])) { y.pmask[2] = mask_index(d.comp[2]); y.pso_h = pso_of(hc); }
                    else { y.side_off = w->col_off[hc]; y.side_ts = w->col_ts[hc]; y.side_pmask_off = w->off_present[d.comp[2]]; }
                    w->gen_box_sys = (int)i;
                }
                break;
            case GGRS_SYS_TTL_DESPAWN: case GGRS_SYS_ADD_U32:
                if (!rb(d.comp[0])) { ok = false; break; }
                y.pso[0] = pso_of(w->comps[d.comp[0]].col_base + d.word[0]); y.pmask[0] = mask_index(d.comp[0]);
                break;
            case GGRS_SYS_SAT_SUB_DESPAWN:
                if (!rb(d.comp[0])) { ok = false; break; }
                if (d.iparam[1] == GGRS_DESPAWN_ROLLBACK) { a.marks = 1; a.dm = w->marks; }    // markers staged in LDS too
                y.pso[0] = pso_of(w->comps[d.comp[0]].col_base + d.word[0]); y.pmask[0] = mask_index(d.comp[0]);
                break;
            case GGRS_SYS_CUSTOM: needs_jit = true; break;     // only the generated kernel runs a user's source
            default: ok = false;
            }
            ++a.n_sys;
        }
        // largest slots-per-workgroup whose LDS image (words + masks + staged tables) fits 64 KiB
        uint32_t sub = 0;
        const uint64_t tables = (uint64_t)(bps / 4 + 4) * 4 + gunits.size() * sizeof(GenUnit);
        const uint64_t per_slot = (uint64_t)bps + (a.marks ? 4 : 0);                     // + the despawned-frame column
        const uint64_t n_masks_lds = (uint64_t)w->plan.n_masks + (a.marks ? 1 : 0);      // + the disabled mask
        for (uint32_t cand : {1024u, 512u, 256u}) if (per_slot * cand + n_masks_lds * (cand / 8) + tables <= 65536) { sub = cand; break; }
        // the kernel generated for this world (jit_source): preferred; k_tick_gen stays as the fallback for worlds it covers
        if (ok && w->knobs.tick_jit) {
            std::string src;
            if (!jit_source(w, src, 1)) w->jit_status = "not covered by the generator (a system writes a live-only component, or too many words per entity)";
            else {
                if (w->knobs.debug_jit > 1) fprintf(stderr, "%s\n", src.c_str());
                const std::string keep = w->err;
                if (jit_cached(w, src, &w->jit_fn) != GGRS_OK) {
                    if (w->knobs.debug_jit) fprintf(stderr, "[ggrs_hip] generated request-group kernel rejected: %s\n", w->err.c_str());
                    w->jit_status = (hiprtc().lib ? "rejected: " : "hiprtc unavailable: ") + w->err.substr(0, 300);
                    w->jit_fn = nullptr; w->err = keep;
                } else w->jit_status = "ok";
                if (w->jit_fn && w->knobs.jit_v == 4 && jit_source(w, src, 4) && jit_cached(w, src, &w->jit_fn4) != GGRS_OK) {
                    if (w->knobs.debug_jit) fprintf(stderr, "[ggrs_hip] generated request-group kernel (4 slots per lane) rejected: %s\n", w->err.c_str());
                    w->jit_fn4 = nullptr; w->err = keep;
                }
                if (w->jit_fn) {
                    for (auto& d : w->systems) w->jit_reads_inputs |= d.kind == GGRS_SYS_CUSTOM || d.kind == GGRS_SYS_BOX_MOVE;
                    for (auto& d : w->systems) w->jit_marks |= d.kind == GGRS_SYS_CUSTOM || (d.kind == GGRS_SYS_SAT_SUB_DESPAWN && d.iparam[1] == GGRS_DESPAWN_ROLLBACK);
                    if (w->jit_marks) { a.marks = 1; a.dm = w->marks; }
                }
            }
        }
        if (ok && (w->jit_fn || (sub && !needs_jit))) { w->gen_ok = true; w->gen_sub_max = sub; }
    }

    // ---- arena carve
    const uint32_t n_tiles = (uint32_t)(w->cap_pad / TILE);
    w->tick_part_stride = 4 * (uint32_t)(w->cap_pad / TILE1);     // one partial per wave of the finest tiling
    w->tick_parts_saves = w->cap_pad <= TICK_VEC1_MAX_SLOTS + 112 * 1024 ? 8 * MAX_TICK_SAVES : MAX_TICK_SAVES;   // k_tick1 worlds: room for a batch of 16 eight-Save groups
    const uint64_t tick_parts_bytes = align_up((uint64_t)w->tick_parts_saves * 3 * w->tick_part_stride * 8, ALIGN);
    w->part_stride = n_tiles + 4096 / 1;            // + room for spawn partial blocks
    const uint64_t parts_bytes = align_up((uint64_t)(w->cks_args.n_cks + 1) * w->part_stride * 8, ALIGN);
    w->max_results = 16384;                             // pinned result ring (256 KiB): a fan-out step of 256 branches x 8 frames alone is 2048
    const uint64_t res_bytes = align_up((uint64_t)w->max_results * 16, ALIGN);
    const uint64_t units_bytes = align_up((units.size() + 1) * sizeof(UnitDesc), ALIGN);
    w->stage_floats = 1u << 20;
    const uint64_t stage_bytes = w->stage_floats * 4;
    const uint64_t wg_parts_bytes = align_up((uint64_t)std::max(n_tiles, 1u) * 4 * MAX_TICK_SAVES * std::max<uint64_t>(3, w->cks_args.n_cks + 1) * 8, ALIGN) + ALIGN;   // one partial row per workgroup of the finest grid (256-slot k_tick1 workgroups) + the ticket
    const uint64_t need = (uint64_t)(w->max_depth + 1) * w->state_bytes + w->side_bytes + parts_bytes + tick_parts_bytes + res_bytes + units_bytes + ALIGN + stage_bytes + wg_parts_bytes;
    if (w->arena) {
        if (w->arena_bytes < need) return w->fail(GGRS_E_INVALID, "arena too small: need %llu bytes, have %llu", (unsigned long long)need, (unsigned long long)w->arena_bytes);
    } else {
        // A/B knobs: arena_align (power of two) aligns the first block inside an over-allocation, arena_skew then shifts it
        const uint64_t al = w->knobs.arena_align, skew = align_up(w->knobs.arena_skew, ALIGN);
        const bool dbg = w->knobs.debug_arena;
        // Placement probe.  The dominant kernel of a big world runs in one of two latency modes (~120 vs ~129 us at 1 M
        // entities) depending on where the arena lands in the physical address space -- nothing in the virtual address
        // predicts it (profiles/README.md, "mode_probe").  For HBM-sized worlds of the stress_test shape, allocate a few
        // candidate arenas, time k_tick on each (uninitialised memory: the traffic is what matters), keep the fastest,
        // free the rest.  OPT-IN (GGRS_ARENA_PROBE=<n> candidates per round): it costs world-creation latency and transient
        // memory, which a library constructor must not spend unasked.
        int n_cand = std::max(1, w->knobs.arena_probe);
        if (!(w->tick_ok && w->max_depth >= 3)) n_cand = 1;
        // transient memory: the kept best + the previous round's losers + the new batch <= 16 GiB
        if (n_cand > 1) n_cand = (int)std::min<uint64_t>((uint64_t)n_cand, std::max<uint64_t>(1, ((16ull << 30) / (need + al + skew) - 1) / 2));
        auto place = [&](uint8_t* alloc) { uint8_t* b = alloc; if (al) b = (uint8_t*)align_up((uint64_t)alloc, al); return b + skew; };
        // Up to 4 rounds of n_cand candidates: a round whose best is not clearly faster than its median (no fast
        // placement among them -- about 1 placement in 5-10 is fast) keeps its best and draws a new batch.  The losers
        // of a round stay allocated until the next batch exists, or the allocator would hand the same spots back.
        std::vector<uint8_t*> cand, losers;
        size_t best = 0; float best_us = 0;
        const uint64_t tick_parts_off = (uint64_t)(w->max_depth + 1) * w->state_bytes + w->side_bytes + parts_bytes;
        for (int round = 0; round < (n_cand > 1 ? 4 : 1); ++round) {
            std::vector<uint8_t*> batch;
            for (int k = 0; k < n_cand; ++k) {
                uint8_t* pa = nullptr;
                // contiguous (write-through, uncached) arenas are what k_tick3's dense nt store streams want -- and a
                // hazard when their physical pages were used through a cached mapping earlier in the process (include/ggrs_hip.h,
                // GGRS_WORLD_CONTIG_ARENA): opt-in per world, never for the generated kernel's worlds (its 4-byte stores prefer
                // plain pages anyway: profiles/r02jit/big2.txt, 1 M 123 vs 144 us)
                // safety net for the one cached mapping the library knows about: once this process has freed a PAGED arena of its
                // own, a later world's request is ignored (a session restart that re-creates the world keeps its contiguous arena:
                // freeing uncached memory leaves no lines behind)
                const bool contig = w->knobs.arena_contig >= 0 ? w->knobs.arena_contig != 0
                                                               : ((w->flags & GGRS_WORLD_CONTIG_ARENA) && w->tick2_ok && (need + al + skew) <= (1536ull << 20) &&
                                                                  g_paged_arena_frees.load(std::memory_order_relaxed) == 0);
                hipError_t me = contig ? hipExtMallocWithFlags((void**)&pa, need + al + skew, hipDeviceMallocContiguous) : hipErrorUnknown;
                w->arena_contiguous = me == hipSuccess;
                if (me != hipSuccess) { (void)hipGetLastError(); pa = nullptr; me = hipMalloc((void**)&pa, need + al + skew); }   // no contiguous range free: plain pages
                if (me != hipSuccess) { (void)hipGetLastError(); break; }
                if (dbg) fprintf(stderr, "[ggrs arena] %s allocation of %llu bytes\n", contig ? "contiguous" : "paged", (unsigned long long)(need + al + skew));
                batch.push_back(pa);
            }
            for (auto q : losers) (void)hipFree(q);
            losers.clear();
            if (batch.empty()) break;
            if (n_cand == 1) { cand = batch; break; }
            std::vector<float> times;
            int prc = GGRS_OK;
            for (size_t k = 0; k < batch.size() && prc == GGRS_OK; ++k) {
                float us = 0;
                prc = probe_arena_placement(w, place(batch[k]), tick_parts_off, &us);
                times.push_back(us);
                if (dbg) fprintf(stderr, "[ggrs arena] round %d candidate %zu at %p: k_tick %.1f us\n", round, k, (void*)place(batch[k]), us);
            }
            if (prc) { for (auto q : batch) (void)hipFree(q); for (auto q : cand) (void)hipFree(q); return prc; }
            size_t bi = 0;
            for (size_t k = 1; k < times.size(); ++k) if (times[k] < times[bi]) bi = k;
            std::vector<float> sorted_t = times; std::sort(sorted_t.begin(), sorted_t.end());
            const float median = sorted_t[sorted_t.size() / 2];
            const bool improved = cand.empty() || times[bi] < best_us;
            for (size_t k = 0; k < batch.size(); ++k) if (!(improved && k == bi)) losers.push_back(batch[k]);
            if (improved) { for (auto q : cand) losers.push_back(q); cand.assign(1, batch[bi]); best_us = times[bi]; }
            if (batch.size() < 2 || best_us < 0.965f * median) break;        // a clearly fast placement: done
        }
        for (auto q : losers) (void)hipFree(q);
        if (cand.empty()) return w->fail(GGRS_E_HIP, "hipMalloc of %llu bytes failed", (unsigned long long)(need + al + skew));
        w->arena_alloc = cand[best];
        w->arena = place(w->arena_alloc);
        w->arena_bytes = need; w->own_arena = true;
        if (dbg) fprintf(stderr, "[ggrs arena] alloc=%p base=%p need=%llu state_bytes=%llu (0x%llx)\n", (void*)w->arena_alloc, (void*)w->arena, (unsigned long long)need, (unsigned long long)w->state_bytes, (unsigned long long)w->state_bytes);
    }
    // GGRS_DEBUG_POISON=1: fill a library-owned arena with a garbage pattern before anything is initialised -- a read of memory the
    // library never wrote (hidden by whatever a previous allocation left there) then fails the parity tests every time
    if (w->knobs.debug_poison && w->own_arena) HIPCHK(w, hipMemsetAsync(w->arena, 0xA5, need, w->stream));
    uint8_t* p = w->arena;
    w->live.ptr = p; p += w->state_bytes;
    w->slots.resize(w->max_depth);
    for (uint32_t i = 0; i < w->max_depth; ++i) { w->slots[i].ptr = p; p += w->state_bytes; w->free_slots.push_back((int)(w->max_depth - 1 - i)); }
    uint8_t* const side = p; p += w->side_bytes;          // == live.ptr + side_off (build_layout)
    w->d_parts = (uint64_t*)p; p += parts_bytes;
    w->d_tick_parts = (uint64_t*)p; p += tick_parts_bytes;
    p += res_bytes;                                  // (reserved; results live in pinned host memory, see below)
    w->d_units = (UnitDesc*)p; p += units_bytes;
    w->d_maskoffs = (uint64_t*)p; p += ALIGN;
    w->d_stage = (float*)p; p += stage_bytes;
    w->d_wg_parts = (uint64_t*)p; p += wg_parts_bytes - ALIGN;
    w->d_ticket = (uint32_t*)p; p += ALIGN;
    w->cks_args.parts = w->d_parts;
    w->cks_args.part_cnt = w->d_parts + (uint64_t)w->cks_args.n_cks * w->part_stride;
    w->cks_args.part_stride = w->part_stride;

    // Checksum(u128) results are written by the kernels straight into pinned, device-mapped host memory:
    // no device->host copy node per request list, one stream sync makes them visible.
    HIPCHK(w, hipHostMalloc((void**)&w->h_results, (size_t)w->max_results * 16, hipHostMallocMapped));
    HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_results, w->h_results, 0));
    HIPCHK(w, hipHostMalloc((void**)&w->h_stage, stage_bytes));
    if (w->jit_fn && w->knobs.host_fold_max_wgs) {
        w->rows_cap = 1u << 20;                                    // 8 MiB of partial rows between two collects
        HIPCHK(w, hipHostMalloc((void**)&w->h_rows, w->rows_cap * 8, hipHostMallocMapped));
        HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_rows, w->h_rows, 0));
    }
    if (w->knobs.debug_poison) { memset(w->h_results, 0xA5, (size_t)w->max_results * 16); memset(w->h_stage, 0xA5, stage_bytes); }
    // zero header + masks of EVERY block (columns need no init: masked by liveness).  Invariant
    // relied on by k_copy_state: mask words beyond a block's dirty_len are zero.
    {
        const uint64_t head = ALIGN + (uint64_t)w->plan.n_masks * align_up(w->cap_pad / 8, ALIGN);   // header + every mask
        HIPCHK(w, hipMemsetAsync(w->live.ptr, 0, head, w->stream));
        for (auto& b : w->slots) HIPCHK(w, hipMemsetAsync(b.ptr, 0, head, w->stream));
        HIPCHK(w, hipMemsetAsync(side, 0, w->side_bytes, w->stream));     // no markers, no non-rollback components yet
        HIPCHK(w, hipMemsetAsync(w->d_ticket, 0, ALIGN, w->stream));      // k_tick2's arrival counter: zero between launches
    }
    if (!units.empty()) HIPCHK(w, hipMemcpyAsync(w->d_units, units.data(), units.size() * sizeof(UnitDesc), hipMemcpyHostToDevice, w->stream));
    if (w->gen_ok) {
        HIPCHK(w, hipMalloc((void**)&w->d_gen_words, (gwords.size() + 1) * sizeof(GenWord)));
        HIPCHK(w, hipMalloc((void**)&w->d_gen_units, (gunits.size() + 1) * sizeof(GenUnit)));
        w->gen_parts_saves = w->tick_parts_saves;                    // same rule as k_tick1's partial rows (a batch of 16 eight-Save groups in small worlds)
        HIPCHK(w, hipMalloc((void**)&w->d_gen_parts, (size_t)w->gen_parts_saves * (w->gen_proto.n_cks + 1) * w->tick_part_stride * 8));
        if (w->knobs.debug_poison) HIPCHK(w, hipMemsetAsync(w->d_gen_parts, 0xA5, (size_t)w->gen_parts_saves * (w->gen_proto.n_cks + 1) * w->tick_part_stride * 8, w->stream));
        if (!gwords.empty()) HIPCHK(w, hipMemcpyAsync(w->d_gen_words, gwords.data(), gwords.size() * sizeof(GenWord), hipMemcpyHostToDevice, w->stream));
        if (!gunits.empty()) HIPCHK(w, hipMemcpyAsync(w->d_gen_units, gunits.data(), gunits.size() * sizeof(GenUnit), hipMemcpyHostToDevice, w->stream));
        w->gen_proto.words = w->d_gen_words; w->gen_proto.units = w->d_gen_units; w->gen_proto.n_units = (uint32_t)gunits.size();
        w->gen_proto.parts = w->d_gen_parts; w->gen_proto.part_stride = w->tick_part_stride;
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->sealed = true;
    return GGRS_OK;
}

inline uint32_t tiles_for(uint64_t n) { return (uint32_t)((n + TILE - 1) / TILE); }

Header header_of(const ggrs_world* w) {
    Header h; memset(&h, 0, sizeof h);
    h.len = w->len; h.frame = w->frame;
    return h;
}

FinalizeArgs no_finalize() { FinalizeArgs f; memset(&f, 0, sizeof f); return f; }

int launch_copy(ggrs_world* w, const Block& src, Block& dst, uint64_t len, uint32_t cls, const FinalizeArgs& fin) {
    const uint64_t cover = std::max(std::max(src.dirty_len, dst.dirty_len), len);
    const uint32_t g = std::max(1u, tiles_for(cover));
    {
        ProfScope ps(w, cls);
        if (w->nt_copy)
            hipLaunchKernelGGL((k_copy_state<true>), dim3(g), dim3(TPB), 0, w->stream, (const uint8_t*)src.ptr, dst.ptr, w->plan, len, header_of(w), fin);
        else
            hipLaunchKernelGGL((k_copy_state<false>), dim3(g), dim3(TPB), 0, w->stream, (const uint8_t*)src.ptr, dst.ptr, w->plan, len, header_of(w), fin);
    }
    HIPCHK(w, hipGetLastError());
    dst.dirty_len = src.dirty_len;
    dst.len = len;
    return GGRS_OK;
}

// Generic checksum pass over the live block -> partials
int launch_checksum(ggrs_world* w) {
    const uint32_t g = std::max(1u, tiles_for(w->live.dirty_len));
    CksArgs a = w->cks_args; a.state = w->live.ptr;
    {
        ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
        hipLaunchKernelGGL(k_checksum, dim3(g, std::max(1u, a.n_cks)), dim3(TPB), 0, w->stream, a, (const UnitDesc*)w->d_units);
    }
    HIPCHK(w, hipGetLastError());
    w->pending_valid = true; w->pending_parts = g;
    return GGRS_OK;
}

FinalizeArgs finalize_args(ggrs_world* w, uint32_t result_idx) {
    FinalizeArgs f; memset(&f, 0, sizeof f);
    f.parts = w->d_parts; f.part_cnt = w->cks_args.part_cnt;
    f.n_cks = w->cks_args.n_cks; f.part_stride = w->part_stride; f.n_parts = w->pending_parts; f.enabled = 1;
    f.total_len = w->len;
    f.out = w->d_results + 2 * (uint64_t)result_idx;
    f.live_hdr = (Header*)w->live.ptr;
    return f;
}

// ---- ring: exact mirror of GgrsSnapshots::{push,confirm,rollback} over slot indices
void ring_pop_front(ggrs_world* w) { w->free_slots.push_back(w->ring_slot.front()); w->ring_slot.pop_front(); w->ring_frame.pop_front(); }
void ring_pop_back(ggrs_world* w) { w->free_slots.push_back(w->ring_slot.back()); w->ring_slot.pop_back(); w->ring_frame.pop_back(); }

void ring_confirm(ggrs_world* w, int32_t confirmed) {          // mod.rs:185-202
    while (!w->ring_frame.empty() && w->ring_frame.back() < confirmed) ring_pop_back(w);
}
int ring_push(ggrs_world* w, int32_t frame, int* slot_out) {    // mod.rs:147-181
    while (!w->ring_frame.empty()) {
        const int32_t current = w->ring_frame.front();
        const uint32_t ad = current >= frame ? (uint32_t)current - (uint32_t)frame : (uint32_t)frame - (uint32_t)current;
        const bool wrapped = ad > (UINT32_MAX / 2);
        if ((current >= frame && !wrapped) || (frame >= current && wrapped)) ring_pop_front(w); else break;
    }
    // evict from the back first so the new slot can reuse the oldest block (same end state as
    // push_front followed by pop_back while len > depth)
    while (!w->ring_frame.empty() && w->ring_frame.size() + 1 > w->depth) ring_pop_back(w);
    if (w->depth == 0) { *slot_out = -1; return GGRS_OK; }
    if (w->free_slots.empty()) return w->fail(GGRS_E_INVALID, "ring depth %zu exceeds provisioned max_depth %u", w->depth, w->max_depth);
    const int s = w->free_slots.back(); w->free_slots.pop_back();
    w->ring_slot.push_front(s); w->ring_frame.push_front(frame);
    *slot_out = s;
    return GGRS_OK;
}
bool ring_rollback(ggrs_world* w, int32_t frame) {             // mod.rs:210-226
    for (;;) {
        if (w->ring_frame.empty()) return false;
        if (w->ring_frame.front() != frame) ring_pop_front(w); else return true;
    }
}

// ---- RollbackDespawned (snapshot/despawn.rs)
inline uint32_t blocks_for_slots(uint64_t n) { return (uint32_t)((align_up(std::max<uint64_t>(n, 1), 64) + TPB - 1) / TPB); }

// LoadWorldSystems::EntityResurrect + the non-rollback side of the entity reconcile; must be queued
// before the kernel that overwrites the live liveness mask.  w->frame is already the loaded frame.
int launch_load_reconcile(ggrs_world* w, const Block& snap) {
    if (!w->has_nr && !w->marks_possible) return GGRS_OK;
    ReconcileArgs a; memset(&a, 0, sizeof a);
    a.live = w->live.ptr; a.snap = snap.ptr; a.off_alive = w->off_alive; a.dm = w->marks; a.frame = w->frame;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback) a.nr_present_off[a.n_nr++] = w->off_present[c];
    const uint64_t cover = std::max(std::max(w->live.dirty_len, snap.dirty_len), w->len);
    a.n_slots_pad64 = align_up(std::max<uint64_t>(cover, 1), 64);
    hipLaunchKernelGGL(k_load_reconcile, dim3(blocks_for_slots(cover)), dim3(TPB), 0, w->stream, a);
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}
// AdvanceWorldSystems::DespawnConfirmed (despawn.rs:89-112), with its Local<ConfirmedFrameCount>
int step_despawn_confirmed(ggrs_world* w) {
    if (w->confirmed == w->dc_local) return GGRS_OK;          // "No work necessary"
    w->dc_local = w->confirmed;
    if (!w->marks_possible) return GGRS_OK;                   // no marker was ever set: nothing to free
    const uint64_t cover = std::max(w->live.dirty_len, w->len);
    hipLaunchKernelGGL(k_despawn_confirmed, dim3(blocks_for_slots(cover)), dim3(TPB), 0, w->stream, w->live.ptr, w->marks,
                       w->confirmed, align_up(std::max<uint64_t>(cover, 1), 64));
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

// ---- SaveWorld
int do_save(ggrs_world* w, uint32_t result_idx) {
    int rc = seal(w); if (rc) return rc;
    // SaveWorldSystems::Checksum -> ChecksumPlugin::update
    if (!w->pending_valid) { rc = launch_checksum(w); if (rc) return rc; }
    // ChecksumPlugin::update (fold) runs inside workgroup 0 of the snapshot copy kernel
    const FinalizeArgs fin = finalize_args(w, result_idx);
    // SaveWorldSystems::Snapshot: sync_depth (caller) -> discard_old_snapshots -> save
    if (w->has_confirmed) ring_confirm(w, w->confirmed);
    int s = -1;
    rc = ring_push(w, w->frame, &s); if (rc) return rc;
    if (s >= 0) { rc = launch_copy(w, w->live, w->slots[s], w->len, GGRS_KERNEL_SAVE, fin); if (rc) return rc; }
    else {
        // depth 0: nothing is stored, but the checksum is still due -> copy live onto itself
        // (no rows: len 0) just to run the fold
        Block self = w->live;
        rc = launch_copy(w, w->live, self, 0, GGRS_KERNEL_SAVE, fin); if (rc) return rc;
    }
    return GGRS_OK;
}

// ---- LoadWorld
int do_load(ggrs_world* w, int32_t frame) {
    int rc = seal(w); if (rc) return rc;
    w->frame = frame;                                           // schedule_systems.rs:244-247
    if (!ring_rollback(w, frame))
        return w->fail(GGRS_E_NO_SNAPSHOT, "Could not rollback to %d: no snapshot at that moment could be found.", frame);
    Block& s = w->slots[w->ring_slot.front()];
    rc = launch_load_reconcile(w, s); if (rc) return rc;        // LoadWorldSystems::EntityResurrect
    // entity.rs:55-99 + component_snapshot.rs:95-123 + RollbackOrdered restore (mod.rs:342):
    // masks, columns and len of the live block := the snapshot's
    w->len = s.len;
    rc = launch_copy(w, s, w->live, s.len, GGRS_KERNEL_LOAD, no_finalize()); if (rc) return rc;
    w->pending_valid = false;
    return GGRS_OK;
}

// ---- spawn bookkeeping shared by the API call and the in-schedule spawn system
int set_masks_for_range(ggrs_world* w, uint64_t first, uint64_t count, uint64_t comp_mask) {
    if (count == 0) return GGRS_OK;
    MaskOffs mo, mc; uint32_t n = 0, nc = 0;
    mo.off[n++] = w->off_alive;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if ((comp_mask >> c) & 1ULL) mo.off[n++] = w->off_present[c];
    // a fresh entity carries no RollbackDespawned marker and only the non-rollback components of its
    // bundle (those masks are live-only: no LoadWorld copy ever cleans them)
    if (w->marks_possible) mc.off[nc++] = w->marks.off_disabled;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback && !((comp_mask >> c) & 1ULL)) mc.off[nc++] = w->off_present[c];
    const uint64_t words = ((first + count - 1) >> 6) - (first >> 6) + 1;
    hipLaunchKernelGGL(k_set_mask_range, dim3((uint32_t)((words + TPB - 1) / TPB)), dim3(TPB), 0, w->stream,
                       w->live.ptr, first, count, n, mo, nc, mc);
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

// Host <-> device copy of `count` words of one column starting at slot `first` (tile-major columns: a head
// piece, the full tiles as one pitched 2D copy, a tail piece; plain arrays: one copy).
int copy_column(ggrs_world* w, uint32_t col, uint64_t first, uint64_t count, void* host, bool to_device) {
    if (count == 0) return GGRS_OK;
    const uint32_t wb = w->col_wb[col], ts = w->col_ts[col];
    uint8_t* h = (uint8_t*)host;
    auto dev = [&](uint64_t slot) { return w->live.ptr + col_at(w->col_off[col], ts, wb, slot); };
    auto piece = [&](uint64_t slot, uint64_t n) -> hipError_t {
        return to_device ? hipMemcpyAsync(dev(slot), h + (slot - first) * wb, n * wb, hipMemcpyHostToDevice, w->stream)
                         : hipMemcpyAsync(h + (slot - first) * wb, dev(slot), n * wb, hipMemcpyDeviceToHost, w->stream);
    };
    if (ts == LAYOUT_TILE * wb) { HIPCHK(w, piece(first, count)); return GGRS_OK; }
    uint64_t s0 = first, end = first + count;
    if (s0 % LAYOUT_TILE) { const uint64_t n = std::min<uint64_t>(end - s0, LAYOUT_TILE - s0 % LAYOUT_TILE); HIPCHK(w, piece(s0, n)); s0 += n; }
    const uint64_t full = (end - s0) / LAYOUT_TILE;
    if (full) {
        const size_t width = (size_t)LAYOUT_TILE * wb;
        if (to_device) HIPCHK(w, hipMemcpy2DAsync(dev(s0), ts, h + (s0 - first) * wb, width, width, full, hipMemcpyHostToDevice, w->stream));
        else HIPCHK(w, hipMemcpy2DAsync(h + (s0 - first) * wb, width, dev(s0), ts, width, full, hipMemcpyDeviceToHost, w->stream));
        s0 += full * LAYOUT_TILE;
    }
    if (s0 < end) HIPCHK(w, piece(s0, end - s0));
    return GGRS_OK;
}

int fill_defaults(ggrs_world* w, uint32_t c, uint64_t first, uint64_t count) {
    const Comp& cc = w->comps[c];
    for (uint32_t k = 0; k < cc.n_words; ++k) {
        uint64_t v = 0; memcpy(&v, &cc.defaults[(size_t)k * cc.word_bytes], cc.word_bytes);
        hipLaunchKernelGGL(k_fill_col, dim3((uint32_t)((count + TPB - 1) / TPB)), dim3(TPB), 0, w->stream,
                           w->live.ptr, w->col_off[cc.col_base + k], w->col_ts[cc.col_base + k], cc.word_bytes, first, count, v);
    }
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

int stage_floats(ggrs_world* w, const float* src, uint64_t n, float** dev_out) {
    if (n > w->stage_floats) return w->fail(GGRS_E_CAPACITY, "spawn payload of %llu floats exceeds the staging buffer", (unsigned long long)n);
    if (w->stage_used + n > w->stage_floats) { HIPCHK(w, hipStreamSynchronize(w->stream)); w->stage_used = 0; }
    memcpy(w->h_stage + w->stage_used, src, n * 4);
    HIPCHK(w, hipMemcpyAsync(w->d_stage + w->stage_used, w->h_stage + w->stage_used, n * 4, hipMemcpyHostToDevice, w->stream));
    *dev_out = w->d_stage + w->stage_used;
    w->stage_used += n;
    return GGRS_OK;
}

uint32_t dt_bits_for_frame(uint64_t fps, int32_t frame) {
    // GgrsTimePlugin::update (time.rs:63-87): runtime = frame * 1e9 / fps ns; the clock's previous
    // elapsed is runtime(frame-1) (restored by its own snapshot on load, time.rs:111), and
    // Time::delta_secs = Duration::as_secs_f32 = secs as f32 + nanos as f32 / 1e9 as f32.
    const uint64_t f = (uint64_t)(int64_t)frame;
    const uint64_t d = f * 1000000000ULL / fps - (f - 1) * 1000000000ULL / fps;
    const uint64_t secs = d / 1000000000ULL; const uint32_t nanos = (uint32_t)(d % 1000000000ULL);
    volatile float a = (float)secs;
    volatile float b = (float)nanos / (float)1000000000u;
    const float r = a + b;
    uint32_t bits; memcpy(&bits, &r, 4);
    return bits;
}

// Commands are deferred: spawns materialise after every system of the frame ran (set.rs:118-134).
int run_spawn_systems(ggrs_world* w, const uint8_t* inputs, uint32_t n_inputs, uint64_t spawn_count,
                      const float* spawn_vx, const float* spawn_vy) {
    int rc = GGRS_OK;
    const uint32_t n_cks = w->cks_args.n_cks;
    uint64_t* part_cnt = w->cks_args.part_cnt;
    for (auto& s : w->systems) {
        if (s.kind != GGRS_SYS_PARTICLES_SPAWN) continue;
        bool pressed = false;                                   // spawn_pressed, particles.rs:254-256
        for (uint32_t k = 0; k < n_inputs; ++k) pressed |= (inputs[k] & (uint8_t)s.iparam[1]) != 0;
        if (!pressed || spawn_count == 0) continue;
        if (w->len + spawn_count > w->capacity) return w->fail(GGRS_E_CAPACITY, "spawn of %llu exceeds capacity %llu", (unsigned long long)spawn_count, (unsigned long long)w->capacity);
        const uint32_t cT = s.comp[0], cV = s.comp[1], cL = s.comp[2];
        const Comp& T = w->comps[cT]; const Comp& V = w->comps[cV]; const Comp& L = w->comps[cL];
        const uint64_t first = w->len;
        float *dvx = nullptr, *dvy = nullptr;
        rc = stage_floats(w, spawn_vx, spawn_count, &dvx); if (rc) return rc;
        rc = stage_floats(w, spawn_vy, spawn_count, &dvy); if (rc) return rc;
        rc = fill_defaults(w, cT, first, spawn_count); if (rc) return rc;
        SpawnArgs a; memset(&a, 0, sizeof a);
        a.state = w->live.ptr;
        for (int k = 0; k < 3; ++k) {
            a.off_t[k] = w->col_off[T.col_base + k]; a.off_v[k] = w->col_off[V.col_base + k];
            memcpy(&a.t_default[k], &T.defaults[(size_t)(w->fused_ok ? w->f_tw + k : k) * 4], 4);
        }
        a.off_ttl = w->col_off[L.col_base + 0];
        a.ts = w->ts;
        a.vx = dvx; a.vy = dvy; a.first = first; a.count = spawn_count; a.ttl = (uint64_t)s.iparam[0];
        const uint32_t gs = (uint32_t)((spawn_count + TPB - 1) / TPB);
        const bool keep = w->pending_valid && (w->pending_parts + gs <= w->part_stride);
        // partial slots appended after the step's (scratch at the tail when partials are not kept)
        const uint32_t pbase = keep ? w->pending_parts : (w->part_stride - std::min(gs, w->part_stride));
        uint64_t* scratch = w->d_parts;   // column 0 exists whenever n_cks > 0; else counts column
        a.part_T = a.part_V = (n_cks ? scratch : part_cnt) + pbase;
        a.cks_T = a.cks_V = 0;
        if (keep) {
            for (uint32_t k = 0; k < n_cks; ++k) {
                if ((int)w->cks_comp[k] == w->f_T && w->f_cksT) { a.part_T = w->d_parts + (uint64_t)k * w->part_stride + pbase; a.cks_T = 1; }
                if ((int)w->cks_comp[k] == w->f_V && w->f_cksV) { a.part_V = w->d_parts + (uint64_t)k * w->part_stride + pbase; a.cks_V = 1; }
            }
        }
        a.part_cnt = part_cnt + pbase;
        if (gs > w->part_stride) return w->fail(GGRS_E_CAPACITY, "spawn too large for partial buffer");
        hipLaunchKernelGGL(k_spawn_particles, dim3(gs), dim3(TPB), 0, w->stream, a);
        HIPCHK(w, hipGetLastError());
        rc = set_masks_for_range(w, first, spawn_count, (1ULL << cT) | (1ULL << cV) | (1ULL << cL)); if (rc) return rc;
        w->len += spawn_count;
        w->live.dirty_len = std::max(w->live.dirty_len, w->len);
        if (keep) w->pending_parts += gs; else w->pending_valid = false;
    }
    return GGRS_OK;
}

template <bool CT, bool CV>
void launch_step_fused(ggrs_world* w, const StepArgs& a, uint32_t g) {
    hipLaunchKernelGGL((k_particles_step<true, true, CT, CV>), dim3(g), dim3(TPB), 0, w->stream, a);
}

// The generated translation unit: ABI text, the entity view, the user's source, and a one-slot-per-lane kernel whose
// binding count and word widths are compile-time constants (so e.w[] lives in registers, not scratch).
std::string custom_source(const ggrs_world* w, const ggrs_world::Custom& c, const char* user) {
    std::string s;
    s += GGRS_CUSTOM_ABI_TEXT;
    char buf[256];
    snprintf(buf, sizeof buf, "static_assert(sizeof(GgrsCustomArgs) == %zu, \"host/device argument block mismatch\");\n", sizeof(GgrsCustomArgs));
    s += buf;
    s += GGRS_ENTITY_TEXT;
    s += "#line 1 \"ggrs_system\"\n";
    s += user;
    snprintf(buf, sizeof buf, "\n#line 1 \"ggrs_custom_kernel\"\n#define GGRS_N_BIND %u\n#define GGRS_N_PRES %u\n", c.n_bind, c.n_pres);
    s += buf;
    s += "__device__ constexpr int GGRS_WB[8] = {";
    for (uint32_t i = 0; i < 8; ++i) { snprintf(buf, sizeof buf, "%u,", i < c.n_bind ? w->comps[c.comp[i]].word_bytes : 4u); s += buf; }
    s += "};\n";
    snprintf(buf, sizeof buf, "#define GGRS_LT_SHIFT %d\n", LT_SHIFT);
    s += buf;
    s += "extern \"C\" __global__ __launch_bounds__(256) void ggrs_custom_kernel(GgrsCustomArgs a) {\n"
         "    const ggrs_u64 e = (ggrs_u64)blockIdx.x * 256 + threadIdx.x;\n"
         "    if (e >= a.len_pad64) return;                         // whole waves only (len padded to 64)\n"
         "    const ggrs_u64 aw = *reinterpret_cast<const ggrs_u64*>(a.state + a.off_alive + (e >> 6) * 8);\n"
         "    ggrs_u64 on = aw;\n"
         "    #pragma unroll\n"
         "    for (int p = 0; p < GGRS_N_PRES; ++p) on &= *reinterpret_cast<const ggrs_u64*>(a.state + a.off_present[p] + (e >> 6) * 8);\n"
         "    bool alive = (aw >> (e & 63)) & 1ULL;\n"
         "    int kill = 0;\n"
         "    if ((on >> (e & 63)) & 1ULL) {\n"
         "        GgrsEntity ent; ent.slot = e; ent.kill = 0;\n"
         "        unsigned char* at[8];\n"
         "        #pragma unroll\n"
         "        for (int i = 0; i < GGRS_N_BIND; ++i) {\n"
         "            at[i] = a.state + a.col_off[i] + (e >> GGRS_LT_SHIFT) * a.ts[i] + (e & ((1ULL << GGRS_LT_SHIFT) - 1)) * GGRS_WB[i];\n"
         "            ent.w[i] = GGRS_WB[i] == 8 ? *reinterpret_cast<const ggrs_u64*>(at[i]) : (ggrs_u64)*reinterpret_cast<const ggrs_u32*>(at[i]);\n"
         "        }\n"
         "        ggrs_system(ent, a.fr);\n"
         "        #pragma unroll\n"
         "        for (int i = 0; i < GGRS_N_BIND; ++i) {\n"
         "            if (GGRS_WB[i] == 8) *reinterpret_cast<ggrs_u64*>(at[i]) = ent.w[i];\n"
         "            else *reinterpret_cast<ggrs_u32*>(at[i]) = (ggrs_u32)ent.w[i];\n"
         "        }\n"
         "        kill = ent.kill;\n"
         "    }\n"
         "    if (kill) alive = false;\n"
         "    const ggrs_u64 nw = __builtin_amdgcn_ballot_w64(alive);\n"
         "    if ((threadIdx.x & 63u) == 0 && nw != aw) *reinterpret_cast<ggrs_u64*>(a.state + a.off_alive + (e >> 6) * 8) = nw;\n"
         "    if (a.defer) {                                        // despawn_rollback on an unconfirmed frame: RollbackDespawned(frame)\n"
         "        const bool mark = kill == 2;\n"
         "        const ggrs_u64 kw = __builtin_amdgcn_ballot_w64(mark);\n"
         "        if (mark) *reinterpret_cast<int*>(a.state + a.off_dframe + e * 4) = a.fr.frame;\n"
         "        if ((threadIdx.x & 63u) == 0 && kw) *reinterpret_cast<ggrs_u64*>(a.state + a.off_disabled + (e >> 6) * 8) |= kw;\n"
         "    }\n"
         "}\n";
    return s;
}

int launch_custom(ggrs_world* w, const ggrs_system_desc& s, uint32_t dt_bits, const uint8_t* inputs, uint32_t n_inputs) {
    const ggrs_world::Custom& c = w->customs[s.comp[0]];
    GgrsCustomArgs a; memset(&a, 0, sizeof a);
    a.state = w->live.ptr; a.off_alive = w->off_alive;
    a.off_disabled = w->marks.off_disabled; a.off_dframe = w->marks.off_dframe;
    a.len_pad64 = align_up(w->len, 64);
    for (uint32_t p = 0; p < c.n_pres; ++p) a.off_present[p] = w->off_present[c.pres_comp[p]];
    for (uint32_t i = 0; i < c.n_bind; ++i) {
        const uint32_t col = w->comps[c.comp[i]].col_base + c.word[i];
        a.col_off[i] = w->col_off[col]; a.ts[i] = w->col_ts[col];
    }
    // despawn_rollback (despawn.rs:129-142): only an unconfirmed frame defers the despawn.  Whether the source calls it is
    // not known to the host, so the markers are assumed possible whenever deferral is on.
    a.defer = (w->confirmed < w->frame) ? 1 : 0;
    if (a.defer) w->marks_possible = true;
    memcpy(&a.fr.dt, &dt_bits, 4);
    a.fr.frame = w->frame;
    a.fr.n_inputs = std::min<uint32_t>(n_inputs, 16);
    for (uint32_t k = 0; k < a.fr.n_inputs; ++k) a.fr.input[k] = inputs[k];
    for (int k = 0; k < 4; ++k) a.fr.fparam[k] = s.fparam[k];
    a.fr.iparam[0] = s.iparam[0]; a.fr.iparam[1] = s.iparam[1];
    const uint32_t gx = (uint32_t)((a.len_pad64 + 255) / 256);
    if (gx == 0) return GGRS_OK;
    void* params[] = {&a};
    HIPCHK(w, hipModuleLaunchKernel(c.fn, gx, 1, 1, 256, 1, 1, 0, w->stream, params, nullptr));
    return GGRS_OK;
}

// ---- AdvanceWorld
int do_advance(ggrs_world* w, uint32_t dt_bits, const uint8_t* inputs, uint32_t n_inputs,
               uint64_t spawn_count, const float* spawn_vx, const float* spawn_vy) {
    int rc = seal(w); if (rc) return rc;
    w->frame += 1;                                              // schedule_systems.rs:254-259
    if (dt_bits == 0) dt_bits = dt_bits_for_frame(w->fps, w->frame);
    rc = step_despawn_confirmed(w); if (rc) return rc;         // AdvanceWorldSystems::DespawnConfirmed, before Main
    const uint32_t g = tiles_for(w->len);
    const uint32_t n_cks = w->cks_args.n_cks;
    uint64_t* part_cnt = w->cks_args.part_cnt;
    w->pending_valid = false;

    auto step_args = [&](const ggrs_system_desc* upd, const ggrs_system_desc* ttl) {
        StepArgs a; memset(&a, 0, sizeof a);
        a.state = w->live.ptr; a.off_alive = w->off_alive; a.dt_bits = dt_bits; a.ts = w->ts;
        if (upd) {
            const Comp& T = w->comps[upd->comp[0]]; const Comp& V = w->comps[upd->comp[1]];
            a.off_pT = w->off_present[upd->comp[0]]; a.off_pV = w->off_present[upd->comp[1]];
            for (int k = 0; k < 3; ++k) {
                a.off_t[k] = w->col_off[T.col_base + upd->word[0] + k];
                a.off_v[k] = w->col_off[V.col_base + upd->word[1] + k];
                a.g[k] = upd->fparam[k];
            }
        }
        if (ttl) {
            const Comp& L = w->comps[ttl->comp[0]];
            a.off_pL = w->off_present[ttl->comp[0]];
            a.off_ttl = w->col_off[L.col_base + ttl->word[0]];
        }
        return a;
    };

    if (g > 0) {
        if (w->fused_ok) {
            const ggrs_system_desc *upd = nullptr, *ttl = nullptr;
            for (auto& s : w->systems) { if (s.kind == GGRS_SYS_PARTICLES_UPDATE) upd = &s; if (s.kind == GGRS_SYS_TTL_DESPAWN) ttl = &s; }
            StepArgs a = step_args(upd, ttl);
            // partial columns in checksum-spec order
            for (uint32_t k = 0; k < n_cks; ++k) {
                if ((int)w->cks_comp[k] == w->f_T) a.part_T = w->d_parts + (uint64_t)k * w->part_stride;
                if ((int)w->cks_comp[k] == w->f_V) a.part_V = w->d_parts + (uint64_t)k * w->part_stride;
            }
            a.part_cnt = part_cnt;
            ProfScope ps(w, GGRS_KERNEL_ADVANCE);
            const bool ck = w->fused_cks;
            if (ck && w->f_cksT && w->f_cksV) launch_step_fused<true, true>(w, a, g);
            else if (ck && w->f_cksT) launch_step_fused<true, false>(w, a, g);
            else if (ck && w->f_cksV) launch_step_fused<false, true>(w, a, g);
            else if (ck) {   // no component checksums at all: still produce the live count
                hipLaunchKernelGGL((k_particles_step<true, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
            } else hipLaunchKernelGGL((k_particles_step<true, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
            if (ck && (w->f_cksT || w->f_cksV)) { w->pending_valid = true; w->pending_parts = g; }
        } else {
            for (auto& s : w->systems) {
                ProfScope ps(w, GGRS_KERNEL_ADVANCE);
                switch (s.kind) {
                case GGRS_SYS_PARTICLES_UPDATE: {
                    StepArgs a = step_args(&s, nullptr);
                    hipLaunchKernelGGL((k_particles_step<true, false, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_TTL_DESPAWN: {
                    StepArgs a = step_args(nullptr, &s);
                    hipLaunchKernelGGL((k_particles_step<false, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_ADD_U32: {
                    const Comp& C = w->comps[s.comp[0]];
                    hipLaunchKernelGGL(k_add_u32, dim3((uint32_t)((w->len + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, w->live.ptr,
                                       w->off_alive, w->off_present[s.comp[0]], w->col_off[C.col_base + s.word[0]], w->col_ts[C.col_base + s.word[0]], (uint32_t)s.iparam[0], w->len);
                } break;
                case GGRS_SYS_SAT_SUB_DESPAWN: {
                    const Comp& C = w->comps[s.comp[0]];
                    const uint64_t lp = align_up(w->len, 64);
                    // despawn_rollback (despawn.rs:129-142): only an unconfirmed frame defers the despawn
                    const int defer = (s.iparam[1] == GGRS_DESPAWN_ROLLBACK && w->confirmed < w->frame) ? 1 : 0;
                    if (defer) w->marks_possible = true;
                    hipLaunchKernelGGL(k_sat_sub_despawn, dim3((uint32_t)((lp + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, w->live.ptr,
                                       w->off_alive, w->off_present[s.comp[0]], w->col_off[C.col_base + s.word[0]], w->col_ts[C.col_base + s.word[0]], (uint32_t)s.iparam[0], lp,
                                       defer, w->frame, w->marks);
                } break;
                case GGRS_SYS_BOX_MOVE: {
                    const Comp& T = w->comps[s.comp[0]]; const Comp& V = w->comps[s.comp[1]]; const Comp& P = w->comps[s.comp[2]];
                    BoxMoveArgs a; memset(&a, 0, sizeof a);
                    a.state = w->live.ptr; a.off_alive = w->off_alive;
                    a.off_pT = w->off_present[s.comp[0]]; a.off_pV = w->off_present[s.comp[1]]; a.off_pP = w->off_present[s.comp[2]];
                    for (int k = 0; k < 3; ++k) { a.off_t[k] = w->col_off[T.col_base + s.word[0] + k]; a.off_v[k] = w->col_off[V.col_base + s.word[1] + k]; }
                    a.off_handle = w->col_off[P.col_base + s.word[2]];
                    a.ts_t = w->col_ts[T.col_base + s.word[0]]; a.ts_v = w->col_ts[V.col_base + s.word[1]]; a.ts_handle = w->col_ts[P.col_base + s.word[2]];
                    a.len = w->len; a.dt_bits = dt_bits;
                    // FRICTION.powf(dt) (box_game.rs:189-195): Rust lowers f32::powf to the platform libm's powf
                    float dtf; memcpy(&dtf, &dt_bits, 4);
                    const float fp = powf(s.fparam[2], dtf);
                    memcpy(&a.friction_pow_bits, &fp, 4);
                    a.accel = s.fparam[0]; a.max_speed = s.fparam[1]; a.half_width = s.fparam[3];
                    a.n_inputs = std::min<uint32_t>(n_inputs, 16);
                    for (uint32_t k = 0; k < a.n_inputs; ++k) a.inputs[k] = inputs[k];
                    hipLaunchKernelGGL(k_box_move, dim3((uint32_t)((w->len + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_CUSTOM: { rc = launch_custom(w, s, dt_bits, inputs, n_inputs); if (rc) return rc; } break;
                default: break;
                }
            }
        }
        HIPCHK(w, hipGetLastError());
    }

    return run_spawn_systems(w, inputs, n_inputs, spawn_count, spawn_vx, spawn_vy);
}

// component_checksum.rs:92-95 (hash the XOR of the entity hashes once more), entity_checksum.rs:29-52, checksum.rs:88-99 (XOR of all
// parts; the upper 64 bits of the u128 are always 0) -- what k_gen_finalize does, over rows the device left in pinned memory
void run_host_folds(ggrs_world* w, uint32_t n) {
    for (; n && !w->folds.empty(); --n) {
        const ggrs_world::HostFold f = w->folds.front(); w->folds.pop_front();
        const uint32_t nc = f.n_cks + 1;
        for (uint32_t m = 0; m < f.members; ++m)
            for (uint32_t sv = 0; sv < f.n_saves; ++sv) {
                uint64_t total = 0;
                for (uint32_t c = 0; c < nc; ++c) {
                    const uint64_t* row = w->h_rows + f.rows_off + ((uint64_t)(m * f.n_saves + sv) * nc + c) * f.g;
                    uint64_t x = 0, sum = 0;
                    for (uint32_t t = 0; t < f.g; ++t) { x ^= row[t]; sum += row[t]; }
                    total ^= c == f.n_cks ? sea_pair(sum, f.total_len) : sea_one(x);
                }
                uint64_t* out = w->h_results + 2 * (uint64_t)(f.res_slot + m * f.n_saves + sv);
                out[0] = total; out[1] = 0;
            }
    }
    // the row buffer is a ring: everything before the oldest unfolded group is free again
    if (w->folds.empty()) { w->rows_used = 0; w->rows_tail = 0; } else w->rows_tail = w->folds.front().rows_off;
}
// room for the partial rows of a group in the pinned row buffer?  (no: the group is folded by k_gen_finalize on the device)
bool host_fold_rows(ggrs_world* w, uint32_t g, uint32_t n_saves, uint32_t n_cks, uint32_t members, uint64_t* off) {
    if (!w->h_rows || w->device_results_only || !n_saves || g > (uint32_t)w->knobs.host_fold_max_wgs) return false;
    const uint64_t need = (uint64_t)g * n_saves * (n_cks + 1) * members;
    if (w->folds.empty()) { w->rows_used = 0; w->rows_tail = 0; }
    uint64_t& head = w->rows_used;                                 // ring: rows of pending folds live in [tail, head) (mod wrap)
    if (head >= w->rows_tail) {
        if (head + need <= w->rows_cap) { *off = head; head += need; return true; }
        if (need < w->rows_tail) { *off = 0; head = need; return true; }          // wrap: the front of the buffer has been folded
        return false;
    }
    if (head + need < w->rows_tail) { *off = head; head += need; return true; }
    return false;
}

int read_back(ggrs_world* w, uint32_t n_results, uint64_t* out) {
    HIPCHK(w, hipStreamSynchronize(w->stream));
    run_host_folds(w, ~0u);
    w->stage_used = 0;
    if (n_results && out) memcpy(out, w->h_results, (size_t)n_results * 16);
    return GGRS_OK;
}

void apply_synctest_confirmed(ggrs_world* w) {
    // handle_requests, schedule_systems.rs:204-220: SyncTest => current_frame - check_distance, if >= 0
    if (w->synctest_cd < 0) return;
    const int32_t c = w->frame - w->synctest_cd;
    if (c >= 0) { w->has_confirmed = true; w->confirmed = c; }
}

// ---- fused request groups: [Load?] (Save | Advance)* as ONE k_tick launch + one finalize ----
// measured crossovers (profiles/README.md, run wpb1): the 1-slot-per-lane kernel has the shortest per-wave dependency
// chain and wins while the chip is under-filled; single-wave workgroups of the 4-slots-per-lane kernel win in between
constexpr uint64_t TICK_WAVE_WG_MAX_SLOTS = 800 * 1024;
bool advance_spawns(const ggrs_world* w, const ggrs_request& r) {
    if (r.spawn_count == 0) return false;
    for (auto& s : w->systems) {
        if (s.kind != GGRS_SYS_PARTICLES_SPAWN) continue;
        for (uint32_t k = 0; k < r.n_inputs; ++k) if (r.inputs[k] & (uint8_t)s.iparam[1]) return true;
    }
    return false;
}


// ---- tracing (the reference: tracing spans "HandleRequests" / "SaveWorld" / "LoadWorld" / "AdvanceWorld" and a
// debug! line per request, schedule_systems.rs:171,224-267).  GGRS_HIP_TRACE=1 prints one line per request to stderr;
// GGRS_HIP_ROCTX=1 opens roctx ranges with the same names (the roctx library is dlopen'ed: no link-time dependency), so a
// `rocprofv3 --marker-trace` timeline shows which requests every fused launch carries.
struct Tracer {
    bool log = false;
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Tracer() {
        if (const char* v = getenv("GGRS_HIP_TRACE")) log = atoi(v) != 0;
        if (const char* v = getenv("GGRS_HIP_ROCTX")) if (atoi(v)) {
            void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);    // what rocprofv3 --marker-trace intercepts
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (!push || !pop) { push = nullptr; pop = nullptr; }
            }
        }
    }
    bool on() const { return log || push; }
    void begin(const char* name) const { if (push) push(name); }
    void end() const { if (pop) pop(); }
};
const Tracer& tracer() { static Tracer t; return t; }
struct TraceRange {
    bool active;
    explicit TraceRange(const char* name) : active(tracer().push != nullptr) { if (active) tracer().begin(name); }
    ~TraceRange() { if (active) tracer().end(); }
};
void trace_request(const ggrs_world* w, const ggrs_request& r) {
    const Tracer& t = tracer();
    if (!t.on()) return;
    char buf[96];
    switch (r.kind) {
    case GGRS_REQ_SAVE: snprintf(buf, sizeof buf, "SaveWorld: saving snapshot for frame %d", r.frame); break;
    case GGRS_REQ_LOAD: snprintf(buf, sizeof buf, "LoadWorld: restoring snapshot for frame %d", r.frame); break;
    case GGRS_REQ_ADVANCE: snprintf(buf, sizeof buf, "AdvanceWorld: advancing to frame: %d", w->frame + 1); break;
    default: snprintf(buf, sizeof buf, "unknown request %u", r.kind);
    }
    if (t.log) fprintf(stderr, "[ggrs_hip] %s\n", buf);
    if (t.push) { t.begin(buf); t.end(); }            // a zero-length marker inside the enclosing HandleRequests range
}

// ---- bookkeeping shared by the two request-group runners (k_tick and k_tick_gen): everything the host does in
// request order while a group is assembled -- frame counters, ring push / confirm / rollback, dirty extents
struct GroupState {
    Block* src; uint64_t cover; uint32_t src_is_live;
    Block* dsts[MAX_TICK_SAVES];
};
// LoadGameState opens a group: the ring slot becomes the source (schedule_systems.rs:238-250)
int group_open(ggrs_world* w, const ggrs_request* reqs, uint32_t& i, GroupState& g) {
    g.src = &w->live; g.cover = w->live.dirty_len; g.src_is_live = 1;
    if (reqs[i].kind != GGRS_REQ_LOAD) return GGRS_OK;
    trace_request(w, reqs[i]);
    apply_synctest_confirmed(w);
    w->frame = reqs[i].frame;
    if (!ring_rollback(w, reqs[i].frame))
        return w->fail(GGRS_E_NO_SNAPSHOT, "Could not rollback to %d: no snapshot at that moment could be found.", reqs[i].frame);
    g.src = &w->slots[w->ring_slot.front()];
    int rc = launch_load_reconcile(w, *g.src); if (rc) return rc;        // EntityResurrect: before the group rewrites live liveness
    w->len = g.src->len;
    g.cover = std::max(g.cover, g.src->dirty_len);
    g.src_is_live = 0;
    ++i;
    return GGRS_OK;
}
// SaveGameState inside a group: discard_old_snapshots + GgrsSnapshots::push (mod.rs:147-202); the copy itself is an op of the kernel
int group_save(ggrs_world* w, GroupState& g, uint32_t k, uint8_t** save_dst, int32_t* save_frame) {
    if (tracer().on()) { ggrs_request r{}; r.kind = GGRS_REQ_SAVE; r.frame = w->frame; trace_request(w, r); }
    apply_synctest_confirmed(w);
    if (w->has_confirmed) ring_confirm(w, w->confirmed);
    int sl = -1;
    int rc = ring_push(w, w->frame, &sl); if (rc) return rc;
    Block* d = sl >= 0 ? &w->slots[sl] : nullptr;
    g.dsts[k] = d;
    save_dst[k] = d ? d->ptr : nullptr;
    save_frame[k] = w->frame;
    if (d) { g.cover = std::max(g.cover, d->dirty_len); d->len = w->len; }
    return GGRS_OK;
}
// AdvanceFrame inside a group: RollbackFrameCount += 1 (schedule_systems.rs:254-259), DespawnConfirmed, Time<GgrsTime>.
// DespawnConfirmed only touches the live-only marker mask, which no op inside a group reads or writes: queueing it
// ahead of the group's launch keeps request order.
// marks_flags != nullptr: the group kernel keeps the RollbackDespawned markers itself (k_tick_gen with a.marks);
// it receives bit 0 = DespawnConfirmed is due before this step (its Local<ConfirmedFrameCount> changed, despawn.rs:92-99),
// bit 1 = the step's frame is unconfirmed, i.e. despawn_rollback() defers (despawn.rs:129-137).
int group_step(ggrs_world* w, const ggrs_request& r, uint32_t* dt_bits_out, uint8_t* marks_flags = nullptr) {
    trace_request(w, r);
    apply_synctest_confirmed(w);
    w->frame += 1;
    if (marks_flags) {
        uint8_t f = 0;
        if (w->confirmed != w->dc_local) { w->dc_local = w->confirmed; f |= 1; }
        if (w->confirmed < w->frame) { f |= 2; w->marks_possible = true; }
        *marks_flags = f;
    } else {
        int rc = step_despawn_confirmed(w); if (rc) return rc;
    }
    *dt_bits_out = r.dt_bits ? r.dt_bits : dt_bits_for_frame(w->fps, w->frame);
    return GGRS_OK;
}
// dead: the group ran checksum-only (dead-snapshot elimination) -- neither its ring slots nor the live block were written, so
// their dirty extents still describe what they hold: lowering them here would leave mask bits beyond the new extent that no
// later pass cleans (ghost entities once len grows back into those words)
void group_close(ggrs_world* w, GroupState& g, uint32_t n_saves, bool dead = false) {
    w->pending_valid = false;
    if (dead) return;
    const uint64_t new_dirty = std::max(g.src->dirty_len, w->len);
    for (uint32_t k = 0; k < n_saves; ++k) if (g.dsts[k]) g.dsts[k]->dirty_len = new_dirty;
    w->live.dirty_len = new_dirty;
}

// Depth-parallel k_tick1 (kernels.hpp): the group's outputs (Saves + live world) are split over grid.z roles of dp_s outputs.
// Every role reads the source block while the others write theirs, so the source must be none of the destinations; below
// ~2 Saves there is no chain to split.  Returns dp_s (0: one workgroup walks the whole group).
uint32_t tick1_depth_parallel(const ggrs_world* w, const TickArgs& a, uint64_t cover) {
    if (!w->knobs.tick1_dp || a.n_saves < 2) return 0;
    const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
    if (writes_live && a.src == a.live) return 0;
    for (uint32_t k = 0; k < a.n_saves; ++k) if (a.save_dst[k] == a.src) return 0;
    if (w->knobs.tick1_dp > 1) return cover <= w->knobs.tick1_dp_max_slots2 ? (uint32_t)w->knobs.tick1_dp : 0;    // A/B: fixed split
    // measured crossovers, us per depth-8 tick (profiles/r02dp/ab2.txt):   10k    30k    50k    70k   100k   200k   300k
    //   whole group per workgroup                                           23.0   23.7   24.3   26.3   27.0   37.3   45.1
    //   1 output per role                                                   15.4   17.4   22.4   26.2   32.2   51.2   69.5
    //   2 outputs per role                                                  16.5   17.6   19.8   22.6   27.8   40.7   54.7
    //   3 outputs per role                                                  16.9   18.2   20.2   22.8   25.3   36.8   48.0
    if (cover <= w->knobs.tick1_dp_max_slots) return 1;
    if (cover <= 2 * w->knobs.tick1_dp_max_slots) return 2;
    if (cover <= 6 * w->knobs.tick1_dp_max_slots) return 3;
    return 0;
}
template <bool NT, bool DP>
void launch_tick1_dp(ggrs_world* w, const TickArgs& a, uint32_t g, uint32_t batch) {
    const dim3 grid(g, batch, DP ? (a.n_saves + 1 + a.dp_s - 1) / a.dp_s : 1);
    if (w->f_cksT && w->f_cksV) hipLaunchKernelGGL((k_tick1<true, true, NT, DP>), grid, dim3(TPB), 0, w->stream, a);
    else if (w->f_cksT) hipLaunchKernelGGL((k_tick1<true, false, NT, DP>), grid, dim3(TPB), 0, w->stream, a);
    else if (w->f_cksV) hipLaunchKernelGGL((k_tick1<false, true, NT, DP>), grid, dim3(TPB), 0, w->stream, a);
    else hipLaunchKernelGGL((k_tick1<false, false, NT, DP>), grid, dim3(TPB), 0, w->stream, a);
}
template <bool NT>
void launch_tick1(ggrs_world* w, TickArgs& a, uint32_t g, uint32_t batch = 1, uint32_t dp_s = 0) {
    a.dp_s = dp_s;
    if (dp_s) launch_tick1_dp<NT, true>(w, a, g, batch); else launch_tick1_dp<NT, false>(w, a, g, batch);
}

// Dead-snapshot elimination.  A request group whose NEXT request is a LoadGameState of a frame older than everything the
// group saved leaves nothing behind: that rollback pops every one of its snapshots from the ring (mod.rs:210-226) before
// anything could load them, and LoadWorld overwrites the live world.  Only the group's Checksum(u128)s are observable --
// exactly what a speculative branch of the fan-out is ([Load(C), Adv, Save, ...] x B in one list: every branch but the last).
// Such a group runs checksum-only: no snapshot stores, no live write.  The host ring bookkeeping is done as usual.
// Not applied when something else reads the live world in between (a firing spawn system, live-only components or
// RollbackDespawned markers, whose reconcile pass reads the live liveness mask).
bool group_is_dead(const ggrs_world* w, const ggrs_request* reqs, uint32_t i, uint32_t n, const int32_t* save_frame, uint32_t n_saves, bool spawn_pending) {
    if (!w->knobs.dead_groups || spawn_pending || n_saves == 0 || i >= n || reqs[i].kind != GGRS_REQ_LOAD || w->has_nr || w->marks_possible) return false;
    bool present = false;
    for (int32_t f : w->ring_frame) present |= f == reqs[i].frame;
    if (!present) return false;                                        // that Load is going to fail: change nothing
    for (uint32_t k = 0; k < n_saves; ++k) {
        const int64_t d = (int64_t)save_frame[k] - (int64_t)reqs[i].frame;
        if (d <= 0 || d > (1 << 30)) return false;                       // not newer (or i32 wrap-around in play): keep it
    }
    return true;
}

constexpr int TICK_RESTL = 8;          // rest rows the register-resident variant of k_tick can carry
constexpr int TICK2_RESTL = TICK2_RESTL_MAX;
// wpb: waves per workgroup (4: one 1024-slot tile per workgroup, g = tiles; 1: one 256-slot quarter per workgroup)
template <bool NT, int WPB>
void launch_tick(ggrs_world* w, const TickArgs& a, uint32_t g) {
    const uint32_t lds = w->knobs.tick_lds;
    const bool rl = w->knobs.tick_rest_loop && a.n_rest_rows <= (uint32_t)TICK_RESTL && a.n_rest_rows > 0 && a.n_saves > 0;
#define GGRS_LAUNCH_TICK(T_, V_) do { \
        if (rl) hipLaunchKernelGGL((k_tick<T_, V_, NT, TICK_RESTL, WPB>), dim3(g), dim3(WPB * 64), lds, w->stream, a); \
        else hipLaunchKernelGGL((k_tick<T_, V_, NT, 0, WPB>), dim3(g), dim3(WPB * 64), lds, w->stream, a); } while (0)
    if (w->f_cksT && w->f_cksV) GGRS_LAUNCH_TICK(true, true);
    else if (w->f_cksT) GGRS_LAUNCH_TICK(true, false);
    else if (w->f_cksV) GGRS_LAUNCH_TICK(false, true);
    else GGRS_LAUNCH_TICK(false, false);
#undef GGRS_LAUNCH_TICK
}

template <bool NT, int ILV>
void launch_tick2(ggrs_world* w, const Tick2Args& a, uint32_t g) {
    const uint32_t lds = w->knobs.tick_lds;
    if (w->f_cksT && w->f_cksV) hipLaunchKernelGGL((k_tick2<true, true, NT, TICK2_RESTL, ILV>), dim3(g), dim3(TPB), lds, w->stream, a);
    else if (w->f_cksT) hipLaunchKernelGGL((k_tick2<true, false, NT, TICK2_RESTL, ILV>), dim3(g), dim3(TPB), lds, w->stream, a);
    else if (w->f_cksV) hipLaunchKernelGGL((k_tick2<false, true, NT, TICK2_RESTL, ILV>), dim3(g), dim3(TPB), lds, w->stream, a);
    else hipLaunchKernelGGL((k_tick2<false, false, NT, TICK2_RESTL, ILV>), dim3(g), dim3(TPB), lds, w->stream, a);
}

template <bool NT, int PS>
void launch_tick3(ggrs_world* w, const Tick2Args& a, uint32_t g) {
    if (a.n_rest_rows != (uint32_t)TICK2_RESTL) {                      // not the stress_test's 7 rows: the general instantiation
        if (w->f_cksT && w->f_cksV) hipLaunchKernelGGL((k_tick3<true, true, NT, TICK3_RESTL_ANY, PS, false>), dim3(g), dim3(512), 0, w->stream, a);
        else if (w->f_cksT) hipLaunchKernelGGL((k_tick3<true, false, NT, TICK3_RESTL_ANY, PS, false>), dim3(g), dim3(512), 0, w->stream, a);
        else if (w->f_cksV) hipLaunchKernelGGL((k_tick3<false, true, NT, TICK3_RESTL_ANY, PS, false>), dim3(g), dim3(512), 0, w->stream, a);
        else hipLaunchKernelGGL((k_tick3<false, false, NT, TICK3_RESTL_ANY, PS, false>), dim3(g), dim3(512), 0, w->stream, a);
        return;
    }
    if (w->f_cksT && w->f_cksV) hipLaunchKernelGGL((k_tick3<true, true, NT, TICK2_RESTL, PS>), dim3(g), dim3(512), 0, w->stream, a);
    else if (w->f_cksT) hipLaunchKernelGGL((k_tick3<true, false, NT, TICK2_RESTL, PS>), dim3(g), dim3(512), 0, w->stream, a);
    else if (w->f_cksV) hipLaunchKernelGGL((k_tick3<false, true, NT, TICK2_RESTL, PS>), dim3(g), dim3(512), 0, w->stream, a);
    else hipLaunchKernelGGL((k_tick3<false, false, NT, TICK2_RESTL, PS>), dim3(g), dim3(512), 0, w->stream, a);
}

// One SyncTest-shaped group -- Load, (Advance, Save) x D, live write -- on a candidate arena, straight through the
// launcher, with the world's real layout and checksum configuration; average of 3 launches after a warm-up.
int probe_arena_placement(ggrs_world* w, uint8_t* base, uint64_t tick_parts_off, float* us_out) {
    TickArgs a = w->tick_proto;
    const uint32_t D = std::min<uint32_t>(8, w->max_depth - 1);
    a.src = base + w->state_bytes; a.live = base; a.src_is_live = 0;
    for (uint32_t k = 0; k < D; ++k) {
        a.save_dst[k] = base + (uint64_t)(2 + k) * w->state_bytes; a.save_frame[k] = (int32_t)k;
        a.dt_bits[k] = dt_bits_for_frame(w->fps, (int32_t)k + 1);
        a.op_bits |= 1ULL << (2 * k);                        // op 2k: Advance, op 2k + 1: Save
    }
    a.n_ops = 2 * D; a.n_saves = D; a.n_steps = D;
    a.len = w->capacity;
    a.parts = (uint64_t*)(base + tick_parts_off); a.part_stride = w->tick_part_stride;
    const uint32_t g = std::max(1u, tiles_for(w->capacity));
    hipEvent_t e0, e1;
    HIPCHK(w, hipEventCreate(&e0)); HIPCHK(w, hipEventCreate(&e1));
    launch_tick<false, 4>(w, a, g);
    HIPCHK(w, hipEventRecord(e0, w->stream));
    for (int i = 0; i < 3; ++i) launch_tick<false, 4>(w, a, g);
    HIPCHK(w, hipEventRecord(e1, w->stream));
    HIPCHK(w, hipEventSynchronize(e1));
    HIPCHK(w, hipGetLastError());
    float ms = 0; HIPCHK(w, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_out = ms / 3 * 1e3f;
    return GGRS_OK;
}

// res_base: first slot of the pinned result ring this list writes to.  wait == false only enqueues
// (ggrs_hip_enqueue_requests); the list then must hold fewer Saves than the ring can take.
// Identical checksum-only groups off the same source block (speculative branches: same ops, same frames, same length)
// are launched TOGETHER: one k_tick1 grid of tiles x K members and one finalize of saves x K, instead of K launch pairs.
struct TickBatch {
    bool active = false; TickArgs a; uint32_t g = 0, k = 0, res_first = 0;
    void start(const TickArgs& a_, uint32_t g_, uint32_t res) { active = true; a = a_; g = g_; k = 1; res_first = res; }
    bool try_add(const ggrs_world* w, const TickArgs& b, uint32_t g_, uint32_t res) {
        if (!active || g_ != g || b.src != a.src || b.len != a.len || b.op_bits != a.op_bits || b.n_ops != a.n_ops || b.n_saves != a.n_saves ||
            b.n_steps != a.n_steps || memcmp(b.dt_bits, a.dt_bits, sizeof(uint32_t) * a.n_steps) != 0) return false;
        if ((k + 1) * a.n_saves > w->tick_parts_saves || res != res_first + k * a.n_saves) return false;
        ++k;
        return true;
    }
    int flush(ggrs_world* w) {
        if (!active) return GGRS_OK;
        active = false;
        {
            ProfScope ps(w, GGRS_KERNEL_TICK);
            // a batch already fills the chip with its members: splitting the chain only adds replayed steps
            launch_tick1<false>(w, a, g, k, k == 1 ? tick1_depth_parallel(w, a, (uint64_t)g * TILE1) : 0u);
        }
        HIPCHK(w, hipGetLastError());
        TickFinArgs f; memset(&f, 0, sizeof f);
        f.parts = w->d_tick_parts; f.part_stride = w->tick_part_stride; f.n_parts = 4 * g;
        f.cks_T = w->f_cksT; f.cks_V = w->f_cksV; f.total_len = a.len;
        f.out = w->d_results + 2 * (uint64_t)res_first;
        {
            ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
            hipLaunchKernelGGL(k_tick_finalize, dim3(a.n_saves * k), dim3(FIN_TPB), 0, w->stream, f);
        }
        HIPCHK(w, hipGetLastError());
        return GGRS_OK;
    }
};

int run_request_groups(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint64_t* checksums_out,
                       uint32_t res_base = 0, bool wait = true, uint32_t* n_saves_out = nullptr) {
    uint32_t i = 0, ns = 0;                      // ns: results pending in d_results
    int rc = GGRS_OK;
    TickBatch batch;
    while (i < n) {
        TickArgs a = w->tick_proto;
        GroupState gs;
        const ggrs_request* spawn_req = nullptr;
        rc = group_open(w, reqs, i, gs); if (rc) return rc;
        a.src_is_live = gs.src_is_live;
        // ---- gather the ops that follow, doing the host-side bookkeeping in request order
        while (i < n && a.n_ops < (uint32_t)MAX_TICK_OPS) {
            const ggrs_request& r = reqs[i];
            if (r.kind == GGRS_REQ_LOAD) break;
            if (r.kind == GGRS_REQ_SAVE) {
                if (a.n_saves == (uint32_t)MAX_TICK_SAVES || (wait && ns + a.n_saves == w->max_results)) break;
                rc = group_save(w, gs, a.n_saves, a.save_dst, a.save_frame); if (rc) return rc;
                ++a.n_ops; ++a.n_saves;                                     // op bit stays 0: Save
            } else if (r.kind == GGRS_REQ_ADVANCE) {
                if (a.n_steps == (uint32_t)MAX_TICK_STEPS) break;
                rc = group_step(w, r, &a.dt_bits[a.n_steps]); if (rc) return rc;
                ++a.n_steps;
                a.op_bits |= 1ULL << a.n_ops; ++a.n_ops;                     // op bit 1: Advance
                if (advance_spawns(w, r)) { spawn_req = &r; ++i; break; }   // Commands flush ends the group
            } else {
                return w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
            }
            ++i;
        }
        const bool dead = group_is_dead(w, reqs, i, n, a.save_frame, a.n_saves, spawn_req != nullptr);
        if (dead) { for (uint32_t k = 0; k < a.n_saves; ++k) a.save_dst[k] = nullptr; a.skip_live = 1; }
        // ---- one pass over the tiles
        const uint64_t cover = std::max(gs.cover, w->len);
        // kernel shape by world size: k_tick1 (1 slot per lane, 256-slot workgroups) for small worlds, k_tick with
        // single-wave workgroups (256 slots, 16 B per lane) in between, k_tick with one 1024-slot tile per 4-wave
        // workgroup for big ones; GGRS_TICK_VEC = 1 / 41 / 4 forces one (A/B)
        const int vec = w->knobs.tick_vec ? w->knobs.tick_vec : (cover <= TICK_VEC1_MAX_SLOTS ? 1 : (cover <= TICK_WAVE_WG_MAX_SLOTS ? 41 : 4));
        const uint32_t n_waves = std::max(1u, (uint32_t)((cover + TILE1 - 1) / TILE1));      // 256-slot quarters
        const uint32_t g = vec == 4 ? std::max(1u, tiles_for(cover)) : n_waves;
        a.src = gs.src->ptr; a.live = w->live.ptr; a.len = w->len;
        a.parts = w->d_tick_parts; a.part_stride = w->tick_part_stride;
        const bool use2 = w->tick2_ok && !w->knobs.tick_vec && cover > w->knobs.tick2_min_slots;
        if (use2) { rc = batch.flush(w); if (rc) return rc; }
        if (use2) {
            // persistent grid, in-kernel fold: ONE launch per group, the Checksum(u128)s land in the pinned result ring
            Tick2Args b = w->tick2_proto;
            b.src = a.src; b.live = a.live; b.len = a.len;
            memcpy(b.save_dst, a.save_dst, sizeof b.save_dst); memcpy(b.save_frame, a.save_frame, sizeof b.save_frame);
            memcpy(b.dt_bits, a.dt_bits, sizeof b.dt_bits);
            b.op_bits = a.op_bits; b.n_ops = a.n_ops; b.n_saves = a.n_saves; b.n_steps = a.n_steps; b.src_is_live = a.src_is_live;
            b.n_units = n_waves; b.skip_live = a.skip_live;
            b.fold.wg_parts = w->d_wg_parts; b.fold.ticket = w->d_ticket;
            b.fold.out = w->d_results + 2 * (uint64_t)(res_base + ns);
            const uint32_t tiles = std::max(1u, tiles_for(cover));
            const uint32_t g2 = w->knobs.tick2_wgs_per_cu > 0 ? std::min<uint32_t>(tiles, (uint32_t)(w->n_cu * w->knobs.tick2_wgs_per_cu)) : tiles;
            if (b.n_ops || !b.src_is_live) {
                ProfScope ps(w, GGRS_KERNEL_TICK);
                const bool nt = w->nt_copy || w->knobs.tick2_nt;
                if (w->knobs.tick3 == 2) { if (nt) launch_tick3<true, 1>(w, b, g2); else launch_tick3<false, 1>(w, b, g2); }
                else if (w->knobs.tick3) { if (nt) launch_tick3<true, 0>(w, b, g2); else launch_tick3<false, 0>(w, b, g2); }
                else if (w->knobs.tick2_ilv) { if (nt) launch_tick2<true, 1>(w, b, g2); else launch_tick2<false, 1>(w, b, g2); }
                else { if (nt) launch_tick2<true, 0>(w, b, g2); else launch_tick2<false, 0>(w, b, g2); }
            }
            HIPCHK(w, hipGetLastError());
            group_close(w, gs, a.n_saves, dead);
            ns += a.n_saves;
        } else if (vec == 1 && dead && !w->nt_copy && batch.try_add(w, a, g, res_base + ns)) {
            // an identical checksum-only group already waits to be launched: this one rides along as blockIdx.y = K
            group_close(w, gs, a.n_saves, dead);
            ns += a.n_saves;
        } else {
        rc = batch.flush(w); if (rc) return rc;
        if (vec == 1 && dead && !w->nt_copy) {
            batch.start(a, g, res_base + ns);                          // launched when the batch is full or something else follows
            group_close(w, gs, a.n_saves, dead);
            ns += a.n_saves;
        } else {
        if (a.n_ops || !a.src_is_live) {
            ProfScope ps(w, GGRS_KERNEL_TICK);
            if (vec == 1) { const uint32_t dp = tick1_depth_parallel(w, a, cover); if (w->nt_copy) launch_tick1<true>(w, a, g, 1, dp); else launch_tick1<false>(w, a, g, 1, dp); }
            else if (vec == 41) { if (w->nt_copy) launch_tick<true, 1>(w, a, g); else launch_tick<false, 1>(w, a, g); }
            else { if (w->nt_copy) launch_tick<true, 4>(w, a, g); else launch_tick<false, 4>(w, a, g); }
        }
        HIPCHK(w, hipGetLastError());
        group_close(w, gs, a.n_saves, dead);
        if (a.n_saves) {
            TickFinArgs f; memset(&f, 0, sizeof f);
            f.parts = w->d_tick_parts; f.part_stride = w->tick_part_stride; f.n_parts = vec == 4 ? 4 * g : (vec == 41 ? g : 4 * g);
            f.cks_T = w->f_cksT; f.cks_V = w->f_cksV; f.total_len = w->len;
            f.out = w->d_results + 2 * (uint64_t)(res_base + ns);
            {
                ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
                hipLaunchKernelGGL(k_tick_finalize, dim3(a.n_saves), dim3(FIN_TPB), 0, w->stream, f);
            }
            HIPCHK(w, hipGetLastError());
            ns += a.n_saves;
        }
        }
        }
        if (spawn_req) {
            rc = run_spawn_systems(w, spawn_req->inputs, spawn_req->n_inputs, spawn_req->spawn_count, spawn_req->spawn_vx, spawn_req->spawn_vy);
            if (rc) return rc;
        }
        if (wait && ns == w->max_results) {                                // flush a full result page
            rc = batch.flush(w); if (rc) return rc;
            rc = read_back(w, ns, checksums_out); if (rc) return rc;
            checksums_out += 2 * (uint64_t)ns; ns = 0;
        }
    }
    rc = batch.flush(w); if (rc) return rc;
    if (n_saves_out) *n_saves_out = ns;
    if (!wait) return GGRS_OK;
    return read_back(w, ns, checksums_out);
}

// The same grouping for worlds served by the generic LDS-staged kernel (k_tick_gen + k_gen_finalize).
// slots per workgroup by world size, us per depth-8 tick (profiles/r02gd/sub.txt):   50k   100k  200k  300k  400k  600k   1M
//   256                                                                                 41.4  45.5  80.5 109.0 140.1 185.5 289.4
//   512                                                                                 51.3  51.9  58.8  97.9 105.5 142.1 202.5
//   1024                                                                                73.7  73.9  75.1  85.4  87.3 147.1 165.1
constexpr uint64_t GEN_SUB256_MAX_SLOTS = 144 * 1024, GEN_SUB512_MAX_SLOTS = 256 * 1024;
// The particles world has two fused paths: the hand-specialised kernels (k_tick3 above ~500 k slots) and the kernel generated
// for it like for any other world.  Below the k_tick3 range the generated kernel is the faster one (profiles/r02jit: 10 k 12.8
// vs 15.5 us per depth-8 tick for k_tick1, 100 k 22.7 vs 25.1, 300 k 40.5 vs 44.5), so a list goes to it while the world is small.
bool use_tick_runner(const ggrs_world* w) {
    if (!w->tick_ok) return false;
    if (!w->jit_fn || !w->gen_ok || w->knobs.tick_vec) return true;
    return std::max(w->len, w->live.dirty_len) > w->knobs.jit_particles_max_slots;
}

// TickBatch for the generated kernel: identical checksum-only groups off one source block ride in one launch (blockIdx.z).
struct JitBatch {
    bool active = false; GgrsJitArgs j; uint32_t g = 0, k = 0, res_first = 0, n_cks = 0;
    void start(const GgrsJitArgs& j_, uint32_t g_, uint32_t res, uint32_t n_cks_) { active = true; j = j_; g = g_; k = 1; res_first = res; n_cks = n_cks_; }
    bool try_add(const ggrs_world* w, const GgrsJitArgs& b, uint32_t g_, uint32_t res) {
        if (!active || g_ != g || b.src != j.src || b.len != j.len || b.op_bits != j.op_bits || b.n_ops != j.n_ops || b.n_saves != j.n_saves ||
            b.n_steps != j.n_steps || memcmp(b.dt_bits, j.dt_bits, sizeof b.dt_bits) != 0 || memcmp(b.aux_bits, j.aux_bits, sizeof b.aux_bits) != 0 ||
            memcmp(b.step_frame, j.step_frame, sizeof b.step_frame) != 0 || memcmp(b.step_confirmed, j.step_confirmed, sizeof b.step_confirmed) != 0 ||
            memcmp(b.step_flags, j.step_flags, sizeof b.step_flags) != 0) return false;
        if (w->jit_reads_inputs && (memcmp(b.inputs, j.inputs, sizeof b.inputs) != 0 || memcmp(b.n_inputs, j.n_inputs, sizeof b.n_inputs) != 0)) return false;
        if ((k + 1) * j.n_saves > w->gen_parts_saves || res != res_first + k * j.n_saves) return false;
        ++k;
        return true;
    }
    int flush(ggrs_world* w) {
        if (!active) return GGRS_OK;
        active = false;
        bool host_fold = false; uint64_t rows_off = 0;
        {
            ProfScope ps(w, GGRS_KERNEL_TICK);
            void* params[] = {&j};
            if (k > 1) j.dp_s = 0;
            host_fold = host_fold_rows(w, g, j.n_saves, n_cks, k, &rows_off);
            if (host_fold) { j.parts = reinterpret_cast<ggrs_u64*>(w->d_rows + rows_off); j.part_stride = g; }
            HIPCHK(w, hipModuleLaunchKernel(w->jit_fn, g, j.dp_s ? (j.n_saves + j.dp_s) / j.dp_s : 1u, k, TPB, 1, 1, 0, w->stream, params, nullptr));
        }
        if (host_fold) { w->folds.push_back({res_first, j.n_saves, g, n_cks, k, rows_off, j.len}); return GGRS_OK; }
        GenFinArgs f; memset(&f, 0, sizeof f);
        f.parts = reinterpret_cast<uint64_t*>(j.parts); f.part_stride = j.part_stride; f.n_parts = g; f.n_cks = n_cks; f.total_len = j.len;   // one row per workgroup
        f.out = w->d_results + 2 * (uint64_t)res_first;
        {
            ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
            hipLaunchKernelGGL(k_gen_finalize, dim3(j.n_saves * k), dim3(FIN_TPB), 0, w->stream, f);
        }
        HIPCHK(w, hipGetLastError());
        return GGRS_OK;
    }
};

int run_request_groups_gen(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint64_t* checksums_out,
                           uint32_t res_base = 0, bool wait = true, uint32_t* n_saves_out = nullptr) {
    uint32_t i = 0, ns = 0;
    int rc = GGRS_OK;
    JitBatch batch;
    while (i < n) {
        GenArgs a = w->gen_proto;
        GroupState gs;
        const ggrs_request* spawn_req = nullptr;
        rc = group_open(w, reqs, i, gs); if (rc) return rc;
        a.src_is_live = gs.src_is_live;
        while (i < n && a.n_ops < (uint32_t)MAX_TICK_OPS) {
            const ggrs_request& r = reqs[i];
            if (r.kind == GGRS_REQ_LOAD) break;
            if (r.kind == GGRS_REQ_SAVE) {
                if (a.n_saves == (uint32_t)MAX_TICK_SAVES || (wait && ns + a.n_saves == w->max_results)) break;
                rc = group_save(w, gs, a.n_saves, a.save_dst, a.save_frame); if (rc) return rc;
                ++a.n_ops; ++a.n_saves;
            } else if (r.kind == GGRS_REQ_ADVANCE) {
                if (a.n_steps == (uint32_t)MAX_TICK_STEPS) break;
                if (r.n_inputs > 16) return w->fail(GGRS_E_INVALID, "more than 16 player inputs");
                uint32_t dtb = 0;
                rc = group_step(w, r, &dtb, a.marks ? &a.step_flags[a.n_steps] : nullptr); if (rc) return rc;
                a.dt_bits[a.n_steps] = dtb;
                a.step_frame[a.n_steps] = w->frame; a.step_confirmed[a.n_steps] = w->confirmed;
                if (w->gen_box_sys >= 0) {                                     // FRICTION.powf(dt), platform libm (box_game.rs:189-195)
                    float dtf; memcpy(&dtf, &dtb, 4);
                    const float fp = powf(w->systems[w->gen_box_sys].fparam[2], dtf);
                    memcpy(&a.aux_bits[a.n_steps], &fp, 4);
                }
                a.n_inputs[a.n_steps] = (uint8_t)r.n_inputs;
                for (uint32_t k = 0; k < r.n_inputs; ++k) a.inputs[a.n_steps][k] = r.inputs[k];
                ++a.n_steps;
                a.op_bits |= 1ULL << a.n_ops; ++a.n_ops;
                if (advance_spawns(w, r)) { spawn_req = &r; ++i; break; }
            } else {
                return w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
            }
            ++i;
        }
        const bool dead = group_is_dead(w, reqs, i, n, a.save_frame, a.n_saves, spawn_req != nullptr);
        if (dead) { for (uint32_t k = 0; k < a.n_saves; ++k) a.save_dst[k] = nullptr; a.skip_live = 1; }
        const uint64_t cover = std::max(gs.cover, w->len);
        if (w->jit_fn) {
            // ---- the kernel generated for this world: 256-slot workgroups, one slot per lane, depth-parallel roles as k_tick1
            GgrsJitArgs j; memset(&j, 0, sizeof j);
            j.src = gs.src->ptr; j.live = w->live.ptr;
            memcpy(j.save_dst, a.save_dst, sizeof j.save_dst); memcpy(j.save_frame, a.save_frame, sizeof j.save_frame);
            memcpy(j.dt_bits, a.dt_bits, sizeof j.dt_bits); memcpy(j.aux_bits, a.aux_bits, sizeof j.aux_bits);
            memcpy(j.inputs, a.inputs, sizeof j.inputs); memcpy(j.n_inputs, a.n_inputs, sizeof j.n_inputs);
            memcpy(j.step_frame, a.step_frame, sizeof j.step_frame); memcpy(j.step_confirmed, a.step_confirmed, sizeof j.step_confirmed);
            memcpy(j.step_flags, a.step_flags, sizeof j.step_flags);
            j.op_bits = a.op_bits; j.n_ops = a.n_ops; j.n_saves = a.n_saves; j.n_steps = a.n_steps; j.src_is_live = a.src_is_live; j.skip_live = a.skip_live;
            j.len = w->len; j.parts = reinterpret_cast<ggrs_u64*>(a.parts); j.part_stride = a.part_stride;
            const bool v4 = w->jit_fn4 && w->knobs.jit_v == 4;
            const uint32_t slots_per_wg = v4 ? (uint32_t)TILE : (uint32_t)TILE1;
            const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + slots_per_wg - 1) / slots_per_wg));
            j.nt = (w->nt_copy || (w->knobs.tick2_nt && cover > w->knobs.tick2_min_slots)) ? 1u : 0u;
            if (w->knobs.tick1_dp && a.n_saves >= 2 && !w->jit_marks) {
                const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
                bool ok = !(writes_live && j.src == j.live);
                for (uint32_t k = 0; k < a.n_saves; ++k) ok = ok && a.save_dst[k] != j.src;
                if (ok) j.dp_s = w->knobs.tick1_dp > 1 ? (cover <= w->knobs.tick1_dp_max_slots2 ? (uint32_t)w->knobs.tick1_dp : 0u)
                               : (cover <= w->knobs.tick1_dp_max_slots ? 1u : (cover <= 2 * w->knobs.tick1_dp_max_slots ? 2u : (cover <= 6 * w->knobs.tick1_dp_max_slots ? 3u : 0u)));   // profiles/r02jit/jit_dp.txt
            }
            // identical checksum-only groups (speculative branches) ride in one launch; a batch already fills the chip, so no roles
            const bool batchable = dead && !v4 && a.n_saves > 0 && !w->jit_marks && cover <= TICK_VEC1_MAX_SLOTS;
            if (batchable && batch.active) {
                GgrsJitArgs jb = j; jb.dp_s = 0;
                if (batch.try_add(w, jb, g, res_base + ns)) { batch.j.dp_s = 0; group_close(w, gs, a.n_saves, dead); ns += a.n_saves; goto group_done; }
            }
            rc = batch.flush(w); if (rc) return rc;
            if (batchable) { batch.start(j, g, res_base + ns, a.n_cks); group_close(w, gs, a.n_saves, dead); ns += a.n_saves; goto group_done; }
            uint64_t rows_off = 0;
            const bool host_fold = (a.n_ops || !a.src_is_live) && host_fold_rows(w, g, a.n_saves, a.n_cks, 1, &rows_off);
            if (host_fold) { j.parts = reinterpret_cast<ggrs_u64*>(w->d_rows + rows_off); j.part_stride = g; }
            if (a.n_ops || !a.src_is_live) {
                ProfScope ps(w, GGRS_KERNEL_TICK);
                void* params[] = {&j};
                HIPCHK(w, hipModuleLaunchKernel(v4 ? w->jit_fn4 : w->jit_fn, g, j.dp_s ? (a.n_saves + j.dp_s) / j.dp_s : 1u, 1, TPB, 1, 1, 0, w->stream, params, nullptr));
            }
            group_close(w, gs, a.n_saves, dead);
            if (host_fold) { w->folds.push_back({res_base + ns, a.n_saves, g, a.n_cks, 1u, rows_off, w->len}); ns += a.n_saves; }
            else if (a.n_saves) {
                GenFinArgs f; memset(&f, 0, sizeof f);
                f.parts = a.parts; f.part_stride = a.part_stride; f.n_parts = g; f.n_cks = a.n_cks; f.total_len = w->len;   // one row per workgroup
                f.out = w->d_results + 2 * (uint64_t)(res_base + ns);
                {
                    ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
                    hipLaunchKernelGGL(k_gen_finalize, dim3(a.n_saves), dim3(FIN_TPB), 0, w->stream, f);
                }
                HIPCHK(w, hipGetLastError());
                ns += a.n_saves;
            }
        } else {
        uint32_t sub = w->gen_sub_max;
        if (w->knobs.gen_sub) sub = std::min<uint32_t>(sub, (uint32_t)w->knobs.gen_sub);
        else if (cover <= GEN_SUB256_MAX_SLOTS) sub = std::min<uint32_t>(sub, 256);
        else if (cover <= GEN_SUB512_MAX_SLOTS) sub = std::min<uint32_t>(sub, 512);
        const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + sub - 1) / sub));
        a.sub = sub; a.src = gs.src->ptr; a.live = w->live.ptr; a.len = w->len;
        // word image + masks + the staged row-offset and checksum-unit tables
        const uint32_t n_rows = a.ts >> (LT_SHIFT + 2);
        const uint32_t lds = n_rows * sub * 4 + a.n_masks * (sub / 8) + ((n_rows + 3u) & ~3u) * 4 + a.n_units * (uint32_t)sizeof(GenUnit) +
                             (a.marks ? sub / 8 + sub * 4 : 0);
        // depth-parallel roles (k_tick1's DP): same validity rule -- the source block is none of the destinations -- and the
        // RollbackDespawned markers stay with the one workgroup that walks the whole group
        a.dp_s = 0;
        // us per depth-8 tick (profiles/r02gd/ab.txt):      10k    50k   100k
        //   whole group per workgroup                        40.2   41.5   45.5
        //   1 / 5 outputs per role                           25.5   33.9   56.3 (5)
        if (w->knobs.gen_dp && a.n_saves >= 2 && !a.marks && cover <= (w->knobs.gen_dp > 1 ? w->knobs.gen_dp_max_slots : 72 * 1024)) {
            const bool writes_live = (!a.src_is_live || a.n_steps) && !a.skip_live;
            bool ok = !(writes_live && a.src == a.live);
            for (uint32_t k = 0; k < a.n_saves; ++k) ok = ok && a.save_dst[k] != a.src;
            if (ok) a.dp_s = w->knobs.gen_dp > 1 ? (uint32_t)w->knobs.gen_dp : (cover <= 32 * 1024 ? 1u : 5u);
        }
        if (a.n_ops || !a.src_is_live) {
            ProfScope ps(w, GGRS_KERNEL_TICK);
            hipLaunchKernelGGL(k_tick_gen, dim3(g, a.dp_s ? (a.n_saves + a.dp_s) / a.dp_s : 1u), dim3(GEN_TPB), lds, w->stream, a);
        }
        HIPCHK(w, hipGetLastError());
        group_close(w, gs, a.n_saves, dead);
        if (a.n_saves) {
            GenFinArgs f; memset(&f, 0, sizeof f);
            f.parts = a.parts; f.part_stride = a.part_stride; f.n_parts = 4 * g; f.n_cks = a.n_cks; f.total_len = w->len;
            f.out = w->d_results + 2 * (uint64_t)(res_base + ns);
            {
                ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
                hipLaunchKernelGGL(k_gen_finalize, dim3(a.n_saves), dim3(FIN_TPB), 0, w->stream, f);
            }
            HIPCHK(w, hipGetLastError());
            ns += a.n_saves;
        }
        }   // k_tick_gen
        group_done:
        if (spawn_req) {
            rc = batch.flush(w); if (rc) return rc;
            rc = run_spawn_systems(w, spawn_req->inputs, spawn_req->n_inputs, spawn_req->spawn_count, spawn_req->spawn_vx, spawn_req->spawn_vy);
            if (rc) return rc;
        }
        if (wait && ns == w->max_results) {
            rc = batch.flush(w); if (rc) return rc;
            rc = read_back(w, ns, checksums_out); if (rc) return rc;
            checksums_out += 2 * (uint64_t)ns; ns = 0;
        }
    }
    rc = batch.flush(w); if (rc) return rc;
    if (n_saves_out) *n_saves_out = ns;
    if (!wait) return GGRS_OK;
    return read_back(w, ns, checksums_out);
}

// Every entry point that touches the device runs with the world's device current on the calling thread and puts the
// caller's device back on return (two worlds on different GPUs in one process; a host thread torch switched elsewhere).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const ggrs_world* w) {
        if (hipGetDevice(&prev) == hipSuccess && prev != w->device) switched = hipSetDevice(w->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Requests are validated BEFORE any host bookkeeping (frame counters, ring) is touched: a malformed list fails with
// GGRS_E_INVALID and leaves the world exactly as it was.
int validate_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        const ggrs_request& r = reqs[i];
        if (r.kind != GGRS_REQ_SAVE && r.kind != GGRS_REQ_LOAD && r.kind != GGRS_REQ_ADVANCE)
            return w->fail(GGRS_E_INVALID, "request %u: unknown request kind %u", i, r.kind);
        if (r.kind != GGRS_REQ_ADVANCE) continue;
        if (r.n_inputs > GGRS_MAX_PLAYERS) return w->fail(GGRS_E_INVALID, "request %u: %u player inputs (at most %d)", i, r.n_inputs, GGRS_MAX_PLAYERS);
        if (r.n_inputs && !r.inputs) return w->fail(GGRS_E_INVALID, "request %u: n_inputs = %u but inputs is NULL", i, r.n_inputs);
        if (advance_spawns(w, r) && (!r.spawn_vx || !r.spawn_vy)) return w->fail(GGRS_E_INVALID, "request %u: a spawn of %llu fires but spawn_vx / spawn_vy is NULL", i, (unsigned long long)r.spawn_count);
    }
    return GGRS_OK;
}
inline bool range_ok(uint64_t first, uint64_t count, uint64_t capacity) { return first <= capacity && count <= capacity - first; }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int ggrs_hip_abi_version(void) { return GGRS_HIP_ABI_VERSION; }

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
    if (!d->arena) { w->block_pad = align_up(w->knobs.block_pad, ALIGN); w->col_pad = align_up(w->knobs.col_pad, ALIGN); }
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
    // each 4-byte word column is 256-B aligned; bytes_per_slot/4 bounds the column count
    // header + liveness/presence masks, 4 KiB aligned, then the tile-major word columns (bytes_per_slot x 1024 per tile)
    const uint64_t state = align_up(align_up(ALIGN + (1 + (uint64_t)n_components) * mask, 4096) + cap_pad * bytes_per_slot, 4096);
    const uint64_t parts = align_up((uint64_t)(n_components + 1) * (cap_pad / TILE + 4096) * 8, ALIGN) +
                           align_up((uint64_t)8 * MAX_TICK_SAVES * 3 * 4 * (cap_pad / TILE1) * 8, ALIGN);
    const uint64_t side = align_up(mask + align_up(cap_pad * 4, ALIGN) + (uint64_t)n_components * mask + cap_pad * bytes_per_slot + (uint64_t)(bytes_per_slot / 4 + 1) * ALIGN, 4096);
    return (uint64_t)(max_depth + 1) * state + side + parts + align_up((cap_pad / TILE) * 4 * MAX_TICK_SAVES * std::max<uint64_t>(3, n_components + 1) * 8, ALIGN) + ALIGN + 16384 * 16 + ALIGN + (uint64_t)(GGRS_MAX_COMPONENTS * GGRS_MAX_CKS_UNITS + 1) * sizeof(UnitDesc) + ALIGN * 4 + (4ULL << 20);
}
void ggrs_hip_world_destroy(ggrs_world* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    for (auto& e : w->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& b : w->pending) (void)hipEventDestroy(b.ev);
    for (auto& e : w->event_pool) (void)hipEventDestroy(e);
    for (auto& c : w->customs) if (c.mod) (void)hipModuleUnload(c.mod);
    if (w->d_gen_words) (void)hipFree(w->d_gen_words);
    if (w->d_gen_units) (void)hipFree(w->d_gen_units);
    if (w->d_gen_parts) (void)hipFree(w->d_gen_parts);
    if (w->h_results) (void)hipHostFree(w->h_results);
    if (w->h_stage) (void)hipHostFree(w->h_stage);
    if (w->h_rows) (void)hipHostFree(w->h_rows);
    if (w->own_arena && w->arena_alloc) { (void)hipFree(w->arena_alloc); if (!w->arena_contiguous) g_paged_arena_frees.fetch_add(1, std::memory_order_relaxed); }
    if (w->own_stream && w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}
const char* ggrs_hip_last_error(ggrs_world* w) { return w ? w->err.c_str() : "null world"; }

int ggrs_hip_register_component(ggrs_world* w, const char* name, uint32_t word_bytes, uint32_t n_words, uint32_t* comp_id) {
    if (!w || !name) return GGRS_E_INVALID;
    if (w->sealed) return w->fail(GGRS_E_INVALID, "register_component after the world was sealed");
    if (w->comps.size() >= GGRS_MAX_COMPONENTS || n_words == 0 || n_words > GGRS_MAX_WORDS || (word_bytes != 4 && word_bytes != 8))
        return w->fail(GGRS_E_INVALID, "bad component shape");
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
    cc.cks_words.clear();
    for (uint32_t k = 0; k < n; ++k) { if (word_idx[k] >= cc.n_words) return w->fail(GGRS_E_INVALID, "word index out of range"); cc.cks_words.push_back(word_idx[k]); }
    if (n * (cc.word_bytes / 4) > GGRS_MAX_CKS_UNITS) return w->fail(GGRS_E_INVALID, "checksum spec too long");
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
int ggrs_hip_generated_kernel_source(ggrs_world* w, uint32_t slots_per_lane, char* buf, uint64_t cap, uint64_t* needed, int compile) {
    if (!w || (slots_per_lane != 1 && slots_per_lane != 4)) return GGRS_E_INVALID;
    if (!w->sealed) {
        if (!w->layout_only) { DeviceGuard dg(w); const int rc = seal(w); if (rc) return rc; }
        else build_layout(w);                                      // host arithmetic only: offsets of every mask and column
    }
    std::string src;
    if (!jit_source(w, src, (int)slots_per_lane)) return w->fail(GGRS_E_INVALID, "the kernel generator does not cover this world (a system writes a live-only component, or more than %u four-byte words per entity)", JIT_MAX_UNITS);
    if (needed) *needed = src.size() + 1;
    if (buf && cap) { const uint64_t n = std::min<uint64_t>(cap, src.size() + 1); memcpy(buf, src.c_str(), n); buf[n - 1] = 0; }
    if (compile) { hipFunction_t fn = nullptr; return hiprtc_build(w, src, "generated request-group kernel", "ggrs_jit_tick", nullptr, &fn); }
    return GGRS_OK;
}
int ggrs_hip_set_frame_rate(ggrs_world* w, uint64_t fps) { if (!w || fps == 0) return GGRS_E_INVALID; w->fps = fps; return GGRS_OK; }

int ggrs_hip_spawn(ggrs_world* w, uint64_t count, uint64_t comp_mask, const void* const* cols, uint64_t* first_slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
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
        ci += cc.n_words;
    }
    rc = set_masks_for_range(w, first, count, comp_mask); if (rc) return rc;
    w->len += count;
    w->live.dirty_len = std::max(w->live.dirty_len, w->len);
    w->pending_valid = false;
    HIPCHK(w, hipStreamSynchronize(w->stream));     // host buffers may be freed on return
    return GGRS_OK;
}
int ggrs_hip_despawn(ggrs_world* w, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (slot >= w->len) return w->fail(GGRS_E_INVALID, "slot out of range");
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_alive, slot, 0);
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_despawn_rollback(ggrs_world* w, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
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
    int rc = seal(w); if (rc) return rc;
    if (c >= w->comps.size() || slot >= w->len || !words) return w->fail(GGRS_E_INVALID, "bad insert_component arguments");
    const Comp& cc = w->comps[c];
    for (uint32_t k = 0; k < cc.n_words; ++k)
        { rc = copy_column(w, cc.col_base + k, slot, 1, const_cast<uint8_t*>((const uint8_t*)words + (size_t)k * cc.word_bytes), true); if (rc) return rc; }
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_present[c], slot, 1);
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}
int ggrs_hip_remove_component(ggrs_world* w, uint32_t c, uint64_t slot) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (c >= w->comps.size() || slot >= w->len) return w->fail(GGRS_E_INVALID, "bad remove_component arguments");
    hipLaunchKernelGGL(k_edit_mask_bit, dim3(1), dim3(1), 0, w->stream, w->live.ptr, w->off_present[c], slot, 0);
    HIPCHK(w, hipGetLastError());
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_upload_word(ggrs_world* w, uint32_t c, uint32_t word, uint64_t first, uint64_t count, const void* src) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (c >= w->comps.size() || word >= w->comps[c].n_words || !range_ok(first, count, w->capacity) || !src) return w->fail(GGRS_E_INVALID, "bad upload_word arguments");
    const Comp& cc = w->comps[c];
    rc = copy_column(w, cc.col_base + word, first, count, const_cast<void*>(src), true); if (rc) return rc;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->pending_valid = false;
    return GGRS_OK;
}
int ggrs_hip_download_word(ggrs_world* w, uint32_t c, uint32_t word, uint64_t first, uint64_t count, void* dst) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
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
    int rc = seal(w); if (rc) return rc;
    return download_mask(w, w->off_alive, dst, n);
}
int ggrs_hip_download_present(ggrs_world* w, uint32_t c, uint64_t* dst, uint64_t n) {
    if (!w || !dst) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
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
    int rc = seal(w); if (rc) return rc;
    return download_mask(w, w->marks.off_disabled, dst, n);
}
int ggrs_hip_download_despawned_frames(ggrs_world* w, uint64_t first, uint64_t count, int32_t* frames) {
    if (!w || !frames) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (!range_ok(first, count, w->capacity)) return w->fail(GGRS_E_INVALID, "bad download_despawned_frames range");
    if (count) HIPCHK(w, hipMemcpyAsync(frames, w->live.ptr + w->marks.off_dframe + first * 4, (size_t)count * 4, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}
int ggrs_hip_column_device_ptr(ggrs_world* w, uint32_t c, uint32_t word, void** p, uint64_t* tile_stride) {
    if (!w || !p) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (c >= w->comps.size() || word >= w->comps[c].n_words) return w->fail(GGRS_E_INVALID, "bad column");
    *p = w->live.ptr + w->col_off[w->comps[c].col_base + word];
    if (tile_stride) *tile_stride = w->col_ts[w->comps[c].col_base + word];
    return GGRS_OK;
}
uint64_t ggrs_hip_len(ggrs_world* w) { return w ? w->len : 0; }
int ggrs_hip_active_count(ggrs_world* w, uint64_t* out) {
    if (!w || !out) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
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
    if (w->tick_ok) { ggrs_request r; memset(&r, 0, sizeof r); r.kind = GGRS_REQ_SAVE; r.frame = w->frame; return run_request_groups(w, &r, 1, out); }
    rc = do_save(w, 0); if (rc) return rc;
    return read_back(w, 1, out);
}
int ggrs_hip_load(ggrs_world* w, int32_t frame) {
    if (!w) return GGRS_E_INVALID;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "synchronous request while %zu enqueued batches are uncollected", w->pending.size());
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (w->tick_ok) { ggrs_request r; memset(&r, 0, sizeof r); r.kind = GGRS_REQ_LOAD; r.frame = frame; return run_request_groups(w, &r, 1, nullptr); }
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
    if (w->tick_ok) return run_request_groups(w, &r, 1, nullptr);
    return do_advance(w, dt_bits, inputs, n_inputs, spawn_count, vx, vy);
}

int ggrs_hip_handle_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint64_t* checksums_out) {
    if (!w || (!reqs && n)) return GGRS_E_INVALID;
    TraceRange tr("HandleRequests");
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (!w->pending.empty()) return w->fail(GGRS_E_INVALID, "handle_requests while %zu enqueued batches are uncollected", w->pending.size());
    rc = validate_requests(w, reqs, n); if (rc) return rc;
    if (w->tick_ok || w->gen_ok) {
        rc = use_tick_runner(w) ? run_request_groups(w, reqs, n, checksums_out) : run_request_groups_gen(w, reqs, n, checksums_out);
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
        case GGRS_REQ_ADVANCE: rc = do_advance(w, r.dt_bits, r.inputs, r.n_inputs, r.spawn_count, r.spawn_vx, r.spawn_vy); break;
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
int ggrs_hip_enqueue_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out) {
    if (!w || (!reqs && n)) return GGRS_E_INVALID;
    TraceRange tr("HandleRequests");
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    rc = validate_requests(w, reqs, n); if (rc) return rc;
    uint32_t n_save = 0;
    for (uint32_t i = 0; i < n; ++i) n_save += reqs[i].kind == GGRS_REQ_SAVE;
    if (n_save > w->max_results / 4 || w->pending_results + n_save > w->max_results / 2 || w->pending.size() >= 16)
        return w->fail(GGRS_E_INVALID, "too many uncollected checksums (%u pending + %u new): call ggrs_hip_collect_checksums", w->pending_results, n_save);
    ggrs_world::PendingBatch b;
    b.first = (w->res_head + n_save > w->max_results) ? 0u : w->res_head;
    b.count = n_save;
    const size_t folds_before = w->folds.size();
    if (w->tick_ok || w->gen_ok) {
        rc = use_tick_runner(w) ? run_request_groups(w, reqs, n, nullptr, b.first, false, nullptr) : run_request_groups_gen(w, reqs, n, nullptr, b.first, false, nullptr);
        if (rc) { (void)hipStreamSynchronize(w->stream); while (w->folds.size() > folds_before) w->folds.pop_back(); return rc; }
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
            case GGRS_REQ_ADVANCE: rc = do_advance(w, r.dt_bits, r.inputs, r.n_inputs, r.spawn_count, r.spawn_vx, r.spawn_vy); break;
            default: rc = w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
            }
        }
        if (rc) { (void)hipStreamSynchronize(w->stream); return rc; }
    }
    if (w->event_pool.empty()) { hipEvent_t e; HIPCHK(w, hipEventCreateWithFlags(&e, hipEventDisableTiming)); w->event_pool.push_back(e); }
    b.ev = w->event_pool.back(); w->event_pool.pop_back();
    HIPCHK(w, hipEventRecord(b.ev, w->stream));
    w->res_head = b.first + n_save; w->pending_results += n_save;
    b.n_folds = (uint32_t)(w->folds.size() - folds_before);
    if (n_saves_out) *n_saves_out = n_save;
    w->pending.push_back(std::move(b));
    return GGRS_OK;
}
int ggrs_hip_collect_checksums(ggrs_world* w, uint64_t* checksums_out, uint32_t max_saves, uint32_t* n_saves_out) {
    if (!w) return GGRS_E_INVALID;
    if (w->pending.empty()) return w->fail(GGRS_E_INVALID, "no enqueued batch to collect");
    DeviceGuard dg(w);
    ggrs_world::PendingBatch& b = w->pending.front();
    if (b.count > max_saves || (b.count && !checksums_out)) return w->fail(GGRS_E_INVALID, "oldest batch holds %u checksums, room for %u", b.count, max_saves);
    HIPCHK(w, hipEventSynchronize(b.ev));
    run_host_folds(w, b.n_folds);
    if (b.count) {
        if (!b.host.empty()) memcpy(checksums_out, b.host.data(), (size_t)b.count * 16);
        else memcpy(checksums_out, w->h_results + 2 * (size_t)b.first, (size_t)b.count * 16);
    }
    if (n_saves_out) *n_saves_out = b.count;
    w->pending_results -= b.count;
    w->event_pool.push_back(b.ev);
    w->pending.pop_front();
    if (w->pending.empty()) w->stage_used = 0;       // every staged spawn payload has been consumed
    return GGRS_OK;
}
uint32_t ggrs_hip_pending_batches(ggrs_world* w) { return w ? (uint32_t)w->pending.size() : 0; }

int ggrs_hip_synchronize(ggrs_world* w) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    HIPCHK(w, hipStreamSynchronize(w->stream));
    return GGRS_OK;
}

uint64_t ggrs_hip_state_bytes(ggrs_world* w) { if (!w) return 0; DeviceGuard dg(w); if (seal(w)) return 0; return w->state_bytes; }
int ggrs_hip_live_state_ptr(ggrs_world* w, void** p) {
    if (!w || !p) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    // keep the header current so an exported block is self-describing
    Header h = header_of(w);
    HIPCHK(w, hipMemcpyAsync(w->live.ptr, &h, 16, hipMemcpyHostToDevice, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    *p = w->live.ptr;
    return GGRS_OK;
}
int ggrs_hip_adopt_live_state(ggrs_world* w) {
    if (!w) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    Header h;
    HIPCHK(w, hipMemcpyAsync(&h, w->live.ptr, sizeof h, hipMemcpyDeviceToHost, w->stream));
    HIPCHK(w, hipStreamSynchronize(w->stream));
    if (h.len > w->capacity) return w->fail(GGRS_E_INVALID, "adopted state has len %llu > capacity", (unsigned long long)h.len);
    w->len = h.len; w->frame = h.frame;
    w->live.dirty_len = std::max(w->live.dirty_len, w->len);
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
    add("arena", !w->sealed ? "none" : (!w->own_arena ? "caller-provided" : (w->arena_contiguous ? "contiguous (hipExtMallocWithFlags, write-through)" : "paged (hipMalloc)")));
    add("arena_bytes", std::to_string(w->arena_bytes));
    {
        Hiprtc& r = hiprtc();
        add("hiprtc", r.lib ? "loaded" : ("missing: " + r.why));
    }
    add("generated_kernel", w->jit_fn ? "ok" : w->jit_status);
    std::string k;
    const uint64_t cover = std::max(w->len, w->live.dirty_len);
    if (!w->sealed) k = "unknown (not sealed)";
    else if (!(w->tick_ok || w->gen_ok)) k = "per-request kernels (k_copy_state, one launch per system)";
    else if (use_tick_runner(w)) {
        const bool use2 = w->tick2_ok && !w->knobs.tick_vec && cover > w->knobs.tick2_min_slots;
        k = use2 ? (w->knobs.tick3 ? "k_tick3 (wave-specialised, in-kernel checksum fold)" : "k_tick2") : "k_tick1 / k_tick + k_tick_finalize";
    } else k = w->jit_fn ? "ggrs_jit_tick (generated for this world)" : "k_tick_gen (LDS-staged interpreter)";
    add("request_group_kernel", k);
    add("slots_covered", std::to_string(cover));
    if (needed) *needed = s.size() + 1;
    if (buf && cap) { const uint64_t n = std::min<uint64_t>(cap, s.size() + 1); memcpy(buf, s.c_str(), n); buf[n - 1] = 0; }
    return GGRS_OK;
}

int ggrs_hip_profile_enable(ggrs_world* w, int on) {
    if (!w) return GGRS_E_INVALID;
    w->prof = on != 0;
    if (on) { for (auto& e : w->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); } w->prof_events.clear();
              for (int i = 0; i < (int)GGRS_KERNEL_CLASSES; ++i) { w->prof_ms[i] = 0; w->prof_n[i] = 0; w->prof_launch_us[i].clear(); } }
    return GGRS_OK;
}
// drains the recorded event pairs into the per-class totals and per-launch lists
static int profile_drain(ggrs_world* w) {
    HIPCHK(w, hipStreamSynchronize(w->stream));
    for (auto& e : w->prof_events) {
        float ms = 0; (void)hipEventElapsedTime(&ms, e.a, e.b);
        w->prof_ms[e.cls] += ms; w->prof_n[e.cls] += 1;
        if (w->prof_launch_us[e.cls].size() < 65536) w->prof_launch_us[e.cls].push_back(ms * 1e3f);
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    w->prof_events.clear();
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
int ggrs_hip_profile_read(ggrs_world* w, double* ms_out, uint64_t* launches_out) {
    if (!w || !ms_out || !launches_out) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = profile_drain(w); if (rc) return rc;
    for (int i = 0; i < (int)GGRS_KERNEL_CLASSES; ++i) { ms_out[i] = w->prof_ms[i]; launches_out[i] = w->prof_n[i]; }
    return GGRS_OK;
}


// =============================================================================================
// Speculative fan-out over RCCL (include/ggrs_hip.h, "Speculative fan-out ACROSS GPUs")
// =============================================================================================
}  // extern "C"

namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    std::string why;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && Broadcast && AllGather && GetErrorString && CommCount && CommUserRank; }
};
// ONE RCCL per process: a copy that is already mapped (a torch process ships its own librccl.so) wins over /opt/rocm's,
// or two collective runtimes would each initialise the device.
void rccl_load(Rccl& r) {
    // GGRS_RCCL_LIB=<path>: load THIS collective library instead (tests: a same-GPU transport double, tests/cpp/rccl_double.cpp,
    // so that the rank != 0 half of the fan-out runs on a one-GPU box where RCCL refuses two ranks per device)
    if (const char* forced = getenv("GGRS_RCCL_LIB")) { if (*forced) r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL); }
    else {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) { const char* e = dlerror(); r.why = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    if (!r.ok()) r.why = "librccl.so lacks an expected entry point";
}
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });                     // worlds may start their fan-out from several threads
    return r;
}
constexpr int FANOUT_MAX_INFLIGHT = 8;
}  // namespace

struct ggrs_fanout {
    ggrs_world* w = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    hipStream_t comm_stream = nullptr;
    // one slot = one all-gather: the checksums of `interval` consecutive steps
    struct Slot { uint64_t* d_send = nullptr; uint64_t* d_recv = nullptr; uint64_t* h_recv = nullptr; hipEvent_t ready = nullptr, done = nullptr;
                  uint32_t n_saves = 0, n_steps = 0; bool closed = false; uint32_t first[64]; };   // first[k]: step k's slot in the pinned result ring
    Slot slot[FANOUT_MAX_INFLIGHT];
    uint32_t head = 0, tail = 0;         // tail: slot being filled, head: oldest uncollected
    uint32_t cap_u128 = 4096;            // checksums per rank a slot can hold (steps x saves)
    uint32_t interval = 1;               // steps per all-gather
    std::string err;
    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};
#define FANCHK_HIP(f, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (f)->fail(GGRS_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define FANCHK_NCCL(f, call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return (f)->fail(GGRS_E_HIP, "%s failed: %s", #call, rccl().GetErrorString(e_)); } while (0)

namespace {
// closes the slot being filled: ONE all-gather of everything it holds, then device -> pinned, on the side stream
int fanout_close_slot(ggrs_fanout* f) {
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    if (s.n_steps == 0 || s.closed) return GGRS_OK;
    const size_t n = (size_t)s.n_steps * s.n_saves;
    ggrs_world* w = f->w;
    // everything of the group happens here, once per `interval` steps and on the side stream: wait for the group's last
    // tick, pinned result ring -> device (consecutive steps sit in consecutive ring slots unless the ring wrapped),
    // all-gather, device -> pinned.  A step itself adds nothing to the world's stream.
    FANCHK_HIP(f, hipEventRecord(s.ready, w->stream));
    FANCHK_HIP(f, hipStreamWaitEvent(f->comm_stream, s.ready, 0));
    for (uint32_t k = 0; k < s.n_steps && s.n_saves; ) {
        uint32_t run = 1;
        while (k + run < s.n_steps && s.first[k + run] == s.first[k] + run * s.n_saves) ++run;
        FANCHK_HIP(f, hipMemcpyAsync(s.d_send + (size_t)k * s.n_saves * 2, w->h_results + 2 * (size_t)s.first[k], (size_t)run * s.n_saves * 16, hipMemcpyHostToDevice, f->comm_stream));
        k += run;
    }
    if (n) {
        FANCHK_NCCL(f, rccl().AllGather(s.d_send, s.d_recv, n * 2, ncclUint64, f->comm, f->comm_stream));
        FANCHK_HIP(f, hipMemcpyAsync(s.h_recv, s.d_recv, n * 16 * f->size, hipMemcpyDeviceToHost, f->comm_stream));
    }
    FANCHK_HIP(f, hipEventRecord(s.done, f->comm_stream));
    s.closed = true;
    ++f->tail;
    return GGRS_OK;
}
}  // namespace

extern "C" {

int ggrs_hip_fanout_unique_id(uint8_t id_out[GGRS_FANOUT_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == GGRS_FANOUT_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return GGRS_E_INVALID;
    if (!rccl().ok()) return GGRS_E_HIP;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return GGRS_E_HIP;
    memcpy(id_out, &id, sizeof id);
    return GGRS_OK;
}
int ggrs_hip_fanout_init(ggrs_world* w, const uint8_t id[GGRS_FANOUT_ID_BYTES], int rank, int world_size, ggrs_fanout** out) {
    if (!w || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (!rccl().ok()) return w->fail(GGRS_E_HIP, "%s", rccl().why.c_str());
    ggrs_fanout* f = new ggrs_fanout();
    f->w = w; f->rank = rank; f->size = world_size;
    w->device_results_only = true;       // the all-gather reads the Checksum(u128)s from the result ring on the GPU's side of the stream: no host-side folds
    ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
    ncclResult_t e = rccl().CommInitRank(&f->comm, world_size, uid, rank);
    if (e != ncclSuccess) { rc = w->fail(GGRS_E_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(e)); delete f; return rc; }
    bool ok = hipStreamCreateWithFlags(&f->comm_stream, hipStreamNonBlocking) == hipSuccess;
    for (auto& s : f->slot) {
        ok = ok && hipMalloc((void**)&s.d_send, (size_t)f->cap_u128 * 16) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.d_recv, (size_t)f->cap_u128 * 16 * world_size) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&s.h_recv, (size_t)f->cap_u128 * 16 * world_size) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { rc = w->fail(GGRS_E_HIP, "fan-out staging buffers could not be allocated"); ggrs_hip_fanout_destroy(f); return rc; }
    *out = f;
    return GGRS_OK;
}
void ggrs_hip_fanout_destroy(ggrs_fanout* f) {
    if (!f) return;
    if (f->w) (void)hipSetDevice(f->w->device);
    if (f->comm_stream) (void)hipStreamSynchronize(f->comm_stream);
    for (auto& s : f->slot) {
        if (s.d_send) (void)hipFree(s.d_send);
        if (s.d_recv) (void)hipFree(s.d_recv);
        if (s.h_recv) (void)hipHostFree(s.h_recv);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (f->comm && rccl().ok()) (void)rccl().CommDestroy(f->comm);
    if (f->comm_stream) (void)hipStreamDestroy(f->comm_stream);
    delete f;
}
const char* ggrs_hip_fanout_last_error(ggrs_fanout* f) { return f ? f->err.c_str() : "null fan-out"; }
int ggrs_hip_fanout_comm_info(ggrs_fanout* f, int* rank_out, int* size_out, int* device_out) {
    if (!f || !f->comm) return GGRS_E_INVALID;
    int n = 0, r = 0;
    FANCHK_NCCL(f, rccl().CommCount(f->comm, &n));
    FANCHK_NCCL(f, rccl().CommUserRank(f->comm, &r));
    if (rank_out) *rank_out = r;
    if (size_out) *size_out = n;
    if (device_out) *device_out = f->w->device;
    return GGRS_OK;
}

int ggrs_hip_fanout_set_interval(ggrs_fanout* f, uint32_t steps_per_all_gather) {
    // a group's steps are all outstanding batches of the world until the group is collected: at most 16 (ggrs_hip_enqueue_requests)
    if (!f || steps_per_all_gather == 0 || steps_per_all_gather > 16) return GGRS_E_INVALID;
    if (f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps) return f->fail(GGRS_E_INVALID, "interval changed inside a partly filled group");
    f->interval = steps_per_all_gather;
    return GGRS_OK;
}

int ggrs_hip_fanout_sync_confirmed(ggrs_fanout* f, int root) {
    if (!f || root < 0 || root >= f->size) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->head != f->tail || f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps || !w->pending.empty())
        return f->fail(GGRS_E_INVALID, "sync_confirmed while steps are in flight: collect them first");
    void* live = nullptr;
    int rc = ggrs_hip_live_state_ptr(w, &live);          // refreshes the block's header (len, frame) on every rank
    if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    // xGMI is point-to-point: one flat broadcast of the packed block, in place in HBM, on the world's own stream
    FANCHK_NCCL(f, rccl().Broadcast(live, live, (size_t)w->state_bytes, ncclUint8, root, f->comm, w->stream));
    FANCHK_HIP(f, hipStreamSynchronize(w->stream));
    rc = ggrs_hip_adopt_live_state(w);
    if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    return GGRS_OK;
}

int ggrs_hip_fanout_step(ggrs_fanout* f, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out) {
    if (!f || (!reqs && n)) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->tail - f->head >= (uint32_t)FANOUT_MAX_INFLIGHT) return f->fail(GGRS_E_INVALID, "%d all-gathers in flight: call ggrs_hip_fanout_collect", FANOUT_MAX_INFLIGHT);
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    if (s.n_steps == 0) { s.closed = false; s.n_saves = 0; }
    // everything that can refuse the step is checked BEFORE the world advances: a batch enqueued here and not tracked by a slot
    // would shift every later collect by one
    uint32_t want = 0;
    for (uint32_t i = 0; i < n; ++i) want += reqs[i].kind == GGRS_REQ_SAVE;
    if (s.n_steps && want != s.n_saves) return f->fail(GGRS_E_INVALID, "steps of one all-gather group must hold the same number of SaveGameState requests (%u vs %u)", want, s.n_saves);
    if ((uint64_t)(s.n_steps + 1) * want > f->cap_u128) return f->fail(GGRS_E_INVALID, "%u checksums per rank in one all-gather (at most %u)", (s.n_steps + 1) * want, f->cap_u128);
    uint32_t ns = 0;
    int rc = ggrs_hip_enqueue_requests(w, reqs, n, &ns);
    if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    s.n_saves = ns;
    s.first[s.n_steps] = w->pending.back().first;        // where the kernels write this step's checksums (pinned result ring)
    ++s.n_steps;
    if (n_saves_out) *n_saves_out = ns;
    if (s.n_steps >= f->interval) return fanout_close_slot(f);
    return GGRS_OK;
}

int ggrs_hip_fanout_collect(ggrs_fanout* f, uint64_t* checksums_out, uint32_t max_u128_per_rank, uint32_t* n_steps_out, uint32_t* n_saves_out) {
    if (!f) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->head == f->tail) {                                   // only a partly filled group is left: close it now
        if (f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps == 0) return f->fail(GGRS_E_INVALID, "no step in flight");
        int rc = fanout_close_slot(f); if (rc) return rc;
    }
    ggrs_fanout::Slot& s = f->slot[f->head % FANOUT_MAX_INFLIGHT];
    const uint32_t per_rank = s.n_steps * s.n_saves;
    if (per_rank > max_u128_per_rank || (per_rank && !checksums_out)) return f->fail(GGRS_E_INVALID, "oldest group holds %u checksums per rank, room for %u", per_rank, max_u128_per_rank);
    FANCHK_HIP(f, hipEventSynchronize(s.done));
    // keep the world's own batch queue in step (its checksums are this rank's rows of the gathered table)
    std::vector<uint64_t> own(2 * (size_t)s.n_saves + 2);
    for (uint32_t k = 0; k < s.n_steps; ++k) {
        uint32_t got = 0;
        int rc = ggrs_hip_collect_checksums(w, own.data(), s.n_saves, &got);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    }
    if (per_rank) memcpy(checksums_out, s.h_recv, (size_t)per_rank * 16 * f->size);
    if (n_steps_out) *n_steps_out = s.n_steps;
    if (n_saves_out) *n_saves_out = s.n_saves;
    s.n_steps = 0; s.closed = false;
    ++f->head;
    return GGRS_OK;
}

}  // extern "C"

