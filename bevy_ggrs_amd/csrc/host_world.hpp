// host_world.hpp -- the ggrs_world object: configuration knobs, registered components and systems, the layout of a packed
// state block, row versions, device buffers and host mirrors.  Part of the single translation unit ggrs_hip.hip (included
// first; everything below the C ABI's opaque `struct ggrs_world` lives in one anonymous namespace).
#pragma once

namespace {

constexpr uint64_t ALIGN = 256;
inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

struct Comp {
    std::string name;
    uint32_t word_bytes = 4, n_words = 0;   // word_bytes: 1, 2, 4 or 8
    std::vector<uint32_t> cks_words;     // as registered
    bool checksummed = false;
    std::string cks_source;              // ggrs_hip_checksum_component_custom: the user's hasher (HIP C++), else empty
    std::vector<uint8_t> defaults;
    uint32_t col_base = 0;               // index of its first column
    bool no_rollback = false;            // GGRS_COMP_NO_ROLLBACK: lives in the side region, outside every snapshot
    // ggrs_hip_register_component_strategy (strategy.rs:22-40): snapshots hold Strategy::Stored -- s_n_words words of s_word_bytes, in columns of
    // their own (scol_base..) that only ring slots use -- and the user's ggrs_store / ggrs_load (HIP C++) convert at Save / Load
    uint32_t s_word_bytes = 0, s_n_words = 0, scol_base = 0;
    std::string strat_source;
};

// Row versions.  Every word column of a block carries the version of the bytes it holds; a version names one state of a
// column over all slots.  Whoever writes a live column -- a GgrsSchedule system (its write set), a spawn, an upload, an
// insert -- gives it a fresh version; SaveWorld copies a column only when the ring slot's version differs from the live
// one, LoadWorld likewise, and both make the destination's version equal to the source's.  A column no system writes (the
// rotation and scale of the stress_test's Transform) therefore reaches every ring slot once and is never stored again:
// the snapshot bytes are identical by construction, only the redundant store is gone.  GGRS_ROW_VERSIONS=0 turns the
// bookkeeping off (every copy moves every row).
// Row versions are 64-bit: a 32-bit counter started over after ~55 minutes of 1 M-entity ticks (~63 versions per depth-8 tick; minutes for small worlds), and a block
// column left alone that long -- the lazy live block, an idle branch block -- could then meet its own old number on other bytes
using ver_t = uint64_t;
static_assert(sizeof(ver_t) == 8, "row versions never start over");
constexpr ver_t VER_NONE = ~0ull;             // "nothing known": never equal to a live version

struct Block {                           // one packed state block in the arena
    uint8_t* ptr = nullptr;
    uint64_t dirty_len = 0;              // slots that may hold non-zero mask bits
    uint64_t len = 0;                    // host mirror of Header::len for ring slots
    std::vector<ver_t> ver;              // per column: the version of the bytes this block holds (VER_NONE: unknown)
    uint64_t tag_ok = 0;                 // bit c: the block's VALUE TAGS of column c (one per 64-slot unit, in the block's tag region) describe its bytes -- see ggrs_world::vtags
};

struct EventPair { hipEvent_t a, b; uint32_t cls; };
struct JitEntry;                         // kernel_gen.hpp: a cached generated module

// Every environment variable the library reads, in ONE place, read ONCE per world at creation (INTEGRATION.md lists them with the
// measurement that keeps each).  None of them changes a result; tests/test_gpu_knobs.py runs a bit-exact parity case under each.
// (GGRS_HIP_TRACE / GGRS_HIP_ROCTX, the two tracing switches, and GGRS_RCCL_LIB are process-wide.)
// Round 5 removed every A/B knob whose losing side is on record in profiles/ (persistent form, group fold, contiguous arena + parking, nt-load /
// first-Save-cache / lane-fold / depth-parallel / presence-version / dead-group / event-on-kernel switches): their winning side is the fixed policy.
struct Knobs {
    bool tick_jit = true;          // GGRS_TICK_JIT=0       no run-time generated kernel (what a deployment without libhiprtc.so AND without shipped code objects gets): one launch per request
    int fold_forward_min_wgs = 1024; // GGRS_FOLD_FORWARD_MIN_WGS=n  request groups of MORE than n workgroups leave their per-workgroup checksum rows in device memory and the NEXT launch on the
                                   //                       stream folds them (fold-forward, host_groups.hpp); up to n the rows go to pinned memory and the host XORs them at collect time
                                   //                       (0: always fold-forward; 1000000: never).  1 M: the host's 30 us fold per tick sat on the collect -> enqueue path; 100 k (391
                                   //                       workgroups): the tags arrive 6-9 us after the batch's event, the host's fold of 75 KB takes 4 -- 14.4 vs 16.5 us per P2P tick (profiles/r05f)
    uint64_t stage_bytes = 8u << 20; // GGRS_STAGE_BYTES=n   bytes of the spawn-payload staging ring (default 8 MiB: one spawn of 1 M particles = 2 x 1 M floats fits; tests shrink it to exercise the wrap)
    bool row_versions = true;      // GGRS_ROW_VERSIONS=0   every SaveWorld / LoadWorld moves every row (no version bookkeeping): the reference's clone-everything cost, measured as bench_fullcopy
    bool debug_poison = false;     // GGRS_DEBUG_POISON=1   fill fresh arenas / scratch with 0xA5 (uninitialised-read hunting)
    int jit_specialise_after = 16; // GGRS_JIT_SPECIALISE_AFTER=n  the n-th group of one shape starts the build of a kernel specialised for it (counted
                                   //                       per shape, 16 shapes per world; 0: never); GGRS_JIT_SPECIALISE_SYNC=1 builds on the calling thread (tests)
    bool jit_specialise_sync = false;
    int spin_wait_us = 200;        // GGRS_SPIN_WAIT_US=n        how long a waiting call polls completion tags in pinned memory (k_gen_finalize's for blocking calls, the fold-forward
                                   //                            role's for collect) before it falls back to hipStreamSynchronize (0: always the stream wait)
    int debug_jit = 0;             // GGRS_DEBUG_JIT=1      say why a generated kernel was rejected; =2 also print its source
    std::string jit_cache_dir;     // GGRS_JIT_CACHE_DIR    code objects of generated kernels on disk ("" = ~/.cache/ggrs_hip; "0": no disk cache)
    std::string aot_dir;           // GGRS_AOT_DIR          shipped code objects (`make -C bevy_ggrs_amd/csrc aot`): looked up by source hash before the run-time compiler is asked
                                   //                       ("" = <directory of libggrs_hip.so>/aot; "0": none)
    bool no_hiprtc = false;        // GGRS_NO_HIPRTC=1      behave as if libhiprtc.so were absent (tests of the shipped-code-object path)
    static Knobs from_env() {
        Knobs k;
        auto num = [](const char* n, long long dflt) { const char* v = getenv(n); return v ? atoll(v) : dflt; };
        k.tick_jit = num("GGRS_TICK_JIT", 1) != 0;
        k.fold_forward_min_wgs = (int)std::max<long long>(0, std::min<long long>(1 << 24, num("GGRS_FOLD_FORWARD_MIN_WGS", 1024)));
        k.stage_bytes = (uint64_t)std::max<long long>(4096, std::min<long long>(1ll << 30, num("GGRS_STAGE_BYTES", 8 << 20))) & ~15ull;
        k.row_versions = num("GGRS_ROW_VERSIONS", 1) != 0;
        k.jit_specialise_after = (int)num("GGRS_JIT_SPECIALISE_AFTER", 16);
        k.jit_specialise_sync = num("GGRS_JIT_SPECIALISE_SYNC", 0) != 0;
        k.spin_wait_us = (int)num("GGRS_SPIN_WAIT_US", 200);
        k.debug_poison = num("GGRS_DEBUG_POISON", 0) != 0;
        k.debug_jit = (int)num("GGRS_DEBUG_JIT", 0);
        k.no_hiprtc = num("GGRS_NO_HIPRTC", 0) != 0;
        if (const char* v = getenv("GGRS_JIT_CACHE_DIR")) k.jit_cache_dir = v;
        if (const char* v = getenv("GGRS_AOT_DIR")) k.aot_dir = v;
        return k;
    }
};
// fixed policies that used to be knobs (the measurements are cited where each is applied)
constexpr uint64_t JIT_DP_MAX_SLOTS = 40 * 1024;            // depth-parallel roles: one output per role up to here, two up to x2, three up to x6 (profiles/r02dp, r02jit)
constexpr uint64_t JIT_CACHED_SAVE_MAX_BYTES = 80ull << 20; // the group's first Save goes through the L2 while its rows are at most this (profiles/r03n, r04c)
constexpr int JIT_SPEC_SHAPES = 16;                         // group shapes counted (and specialised kernels kept) per world
constexpr uint64_t VTAGS_MIN_BYTES = 80ull << 20;           // value tags by default when one steady Save of the whole world moves at least this much: 3 M +16 %, 4 M +18 %, all-columns-hot 2 M +18 %; below (2 M -19 %, all-columns-hot 1 M -3 %) the launch is bound by its vector ALUs and the bookkeeping costs more than the bytes (profiles/r06h)
constexpr uint32_t SELF_FOLD_MAX_WGS = 512;                // self-fold (blocking calls): at most this many fold workgroups wait inside the launch for its tiles
constexpr uint32_t HOST_FOLD_MAX_WGS_BLOCKING = 1024;       // blocking calls: larger groups are folded by k_gen_finalize (the host's fold would be serial with the kernel)

}  // namespace

// What a specialised request-group kernel hard-codes: the op sequence and every wave-uniform mask of the group.
struct JitSig {
    uint64_t op_bits = 0, save_rows = 0, live_rows = 0, load_rows = 0;
    uint32_t n_ops = 0, n_saves = 0, n_steps = 0, src_is_live = 0, skip_live = 0, nt = 0, cached_saves = 0, save_pmask = 0, live_pmask = 0, dp_s = 0, nt_loads = 0;
    uint32_t vtags = 0;                   // value tags: Saves compare tags and skip columns whose bytes the destination already holds
    uint32_t members = 0;                 // a launch of batch members with records (ggrs_hip_fanout_step_branches): a.mtab stays an argument, the per-member fields are not literals
    bool operator==(const JitSig& o) const {
        return members == o.members && vtags == o.vtags && op_bits == o.op_bits && save_rows == o.save_rows && live_rows == o.live_rows && load_rows == o.load_rows && n_ops == o.n_ops && n_saves == o.n_saves &&
               n_steps == o.n_steps && src_is_live == o.src_is_live && skip_live == o.skip_live && nt == o.nt && cached_saves == o.cached_saves &&
               save_pmask == o.save_pmask && live_pmask == o.live_pmask && dp_s == o.dp_s && nt_loads == o.nt_loads;
    }
};
struct JitSpec {
    JitSig sig; std::atomic<int> state{0};                // 1 building (worker thread), 2 ready, 3 failed
    hipModule_t mod = nullptr; hipFunction_t fn = nullptr; std::thread th; std::string why;
};
// One group shape the session has sent: how often, when last, and its kernel once it earned one.  A SyncTest session has one steady
// shape; a P2P session has one per rollback length (0 .. max prediction) -- the table holds Knobs::jit_spec_shapes (16) of them, least recently
// used first out (shapes without a kernel before shapes with one).
struct JitSpecSlot { JitSig sig; uint32_t seen = 0; uint64_t last_use = 0; JitSpec* spec = nullptr; };
constexpr uint32_t JIT_SPEC_MAX_BUILDS = 64;               // per world: a session whose shapes never settle stops asking for kernels

namespace { struct JitLayout; }          // kernel_gen.hpp: the device-side layout of this world's argument block

// ggrs_hip_host_timeline: where the HOST spends a tick (microseconds, summed over the calls since it was enabled)
struct HostTimeline {
    bool on = false;
    uint64_t n_enqueue = 0, n_collect = 0, n_launches = 0;
    double enqueue_us = 0, validate_us = 0, launch_us = 0, collect_us = 0, wait_us = 0, tag_wait_us = 0, fold_us = 0;
};
inline double tl_now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Does `src` contain `tok` as a whole identifier?  (Used to decide whether a user-written system can defer a despawn: GgrsEntity::despawn_rollback() is the only
// way to do so besides writing the `kill` field itself.  Comments count -- a false positive costs speed, never correctness.)
inline bool source_has_token(const std::string& src, const char* tok) {
    const size_t n = strlen(tok);
    auto idc = [](char c) { return isalnum((unsigned char)c) || c == '_'; };
    for (size_t p = src.find(tok); p != std::string::npos; p = src.find(tok, p + 1))
        if ((p == 0 || !idc(src[p - 1])) && (p + n >= src.size() || !idc(src[p + n]))) return true;
    return false;
}
struct ggrs_world {
    // ---- configuration
    int device = 0;
    uint64_t capacity = 0, cap_pad = 0;
    uint32_t max_depth = 0, flags = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    uint8_t* arena = nullptr; uint64_t arena_bytes = 0; bool own_arena = false;

    std::vector<Comp> comps;
    std::vector<ggrs_system_desc> systems;
    struct Custom {                      // GGRS_SYS_CUSTOM: a hiprtc-compiled per-entity system (systems[i].comp[0] indexes this)
        std::string name, source;
        bool may_defer = true;           // the source names despawn_rollback() or the `kill` field: it can leave RollbackDespawned markers (source_has_token)
        hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
        uint32_t n_bind = 0, comp[GGRS_CUSTOM_MAX_BINDINGS] = {}, word[GGRS_CUSTOM_MAX_BINDINGS] = {};
        uint32_t n_pres = 0, pres_comp[GGRS_CUSTOM_MAX_BINDINGS] = {};
    };
    std::vector<Custom> customs;
    struct SpawnSys {                    // GGRS_SYS_SPAWN_CUSTOM: a user-written spawner (ggrs_hip_add_spawn_system; systems[i].comp[0] indexes this)
        std::string name, source;
        uint32_t n_bind = 0, comp[GGRS_CUSTOM_MAX_BINDINGS] = {}, word[GGRS_CUSTOM_MAX_BINDINGS] = {};
        uint64_t bundle_mask = 0; uint32_t payload_stride = 0;
    };
    std::vector<SpawnSys> spawn_customs;
    uint32_t input_bytes = 1, max_players = GGRS_MAX_PLAYERS;   // ggrs_hip_set_input_layout: bytes of one player's T::Input, players of the session
    // the request-group kernel generated for this world (kernel_gen.hpp): one workgroup per 256 slots (roles, batches, fold-forward)
    hipFunction_t jit_fn = nullptr;
    // the kernel specialised for the group shapes the session keeps sending (kernel_gen.hpp jit_specialise): its text, the
    // shapes being counted and their kernels (built on a worker thread, one at a time; used once `state` says ready)
    std::string jit_src; std::vector<JitSpecSlot> spec_tab; uint64_t spec_clock = 0; uint32_t spec_builds = 0; int spec_last_slot = -1;
    int spec_shapes = JIT_SPEC_SHAPES;   // places in spec_tab (a test hook shrinks it to exercise the eviction: ggrs_dbg_set_spec_shapes)
    JitEntry* jit_entry = nullptr;       // handed back to the module cache when the world is destroyed
    JitLayout* jl = nullptr;             // device-side layout of the argument block (owned)
    std::vector<unsigned char> jit_argbuf;   // the packed argument block of the launch being issued
    std::string jit_status = "not attempted";   // why the world has / has not a generated kernel (ggrs_hip_world_kernel_info)
    std::string jit_origin;              // where the generic kernel's code object came from: "hiprtc", "disk cache", "shipped (aot)"
    bool jit_marks = false;              // the generated kernel keeps the RollbackDespawned markers (a system may defer a despawn)
    bool jit_reads_inputs = false;       // a system reads PlayerInputs (BOX_MOVE, custom): branches with different inputs differ
    int jit_box_sys = -1;                // index of a BOX_MOVE system (its FRICTION.powf(dt) is evaluated per step on the host)
    int jit_spawn_sys = -1;              // index of the spawn system the generated kernel runs inside request groups (kernel_gen.hpp jit_fused_spawn_system); -1: a firing spawn ends the group
    uint32_t cap_saves = MAX_TICK_SAVES, cap_steps = MAX_TICK_STEPS;   // a request group of this world ends at this many Saves / steps (kernel_gen.hpp jit_layout)
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
    bool has_strategy = false;           // some component's snapshots hold Strategy::Stored (only the generated kernel can serve the world)
    bool marks_possible = false;         // a RollbackDespawned marker may exist in the live world
    int32_t dc_local = 0;                // Local<ConfirmedFrameCount> of despawn_confirmed_entities (despawn.rs:92)
    uint64_t state_bytes = 0, off_alive = 0;
    // VALUE TAGS.  Row versions know which columns a system MAY write; whether a column's values really changed only the kernel can see.  Every block carries,
    // per 64-slot unit and word column, a 32-bit tag naming the identity of those 64 values: the generated kernel loads the source's tags with the unit, gives a
    // column a fresh tag in the step that changes any of its 64 values (a wave-uniform __ballot of new != old), and at a Save compares with the tag the
    // destination holds -- equal (and non-zero) tags mean equal bytes, and the store is skipped.  The stress_test's translation.z and velocity.x / .z never change:
    // 12 of the 32 bytes a Save moves per entity.  Whoever writes a block's bytes WITHOUT the generated kernel (uploads, host-side spawns, collectives) clears
    // that block's Block::tag_ok bits; tag 0 never matches; ids are unique per launch, step and batch member (tag_counter; a wrap invalidates every block).
    uint64_t off_tags = 0; uint32_t tag_row_bytes = 0;     // the tag region of a block: [unit][column] u32
    uint64_t tag_cols = 0;                                 // columns that carry tags (plain rollback columns; a component under a Strategy is always stored)
    uint32_t tag_counter = 1, tag_wraps = 0;
    int vtags_mode = -1; bool vtags = false;               // ggrs_dbg_set_value_tags: 0 off, 1 on, -1 by size (seal decides: VTAGS_MIN_BYTES)
    uint64_t* d_skip = nullptr;                            // profiling: bytes the launches did NOT store thanks to the tags (ggrs_hip_profile_read_bytes stays honest)
    // SPAWNS DECIDED ON THE DEVICE (kernel_gen.hpp GgrsJitArgs::sp_*): a system called e.spawn(n).  RollbackOrdered::len then lives on the device -- the host's `len`
    // is what the last launch it waited for reported (len_sync) -- every launch covers the world's whole capacity and is COOPERATIVE (all workgroups resident)
    bool dev_spawn = false; bool len_stale = false;
    uint64_t* d_sp_sums = nullptr; uint8_t* d_sp_prec = nullptr; uint64_t* d_sp_link = nullptr;
    volatile uint64_t* h_sp_len = nullptr; uint64_t* d_sp_len = nullptr; uint32_t sp_tiles = 0, sp_epoch = 0; int sp_regs = 0, sp_sregs = 0, sp_per_cu = 0;
    std::vector<uint64_t> off_present, col_off;   // col_off: block-relative offset of the column's row in tile 0 (component words first, then the Stored words of strategy components)
    std::vector<uint32_t> col_wb, col_ts;          // word bytes / tile stride of every column (kernels.hpp col_at)
    std::vector<uint8_t> col_rb;                   // column is part of a rollback component (snapshotted)
    uint32_t n_tcols = 0;                          // columns of component words (the row-version masks cover exactly these)
    uint32_t ts = 0;                               // tile stride of the rollback word columns: bytes of all their words x 8192 slots
    CopyPlan plan{};                               // every mask + every row (rows carry their column index)
    std::vector<uint32_t> row_col;                 // plan.row[r] belongs to column row_col[r]

    // ---- row versions (see Block::ver)
    ver_t ver_counter = 0xFFFFFF00ull;   // (starts just below 2^32: every world's versions cross that line within its first ticks, so a 32-bit copy of one anywhere would show in the tests)
    std::vector<ver_t> cur_ver;                    // the LOGICAL live state's version per column (ahead of live.ver inside a fused group)
    std::vector<ver_t> group_save_ver;             // scratch of the group being assembled: [Save k][column] = the versions slot k will hold
    std::vector<uint8_t> col_ext;                  // a device pointer to this live column was handed out: assume it changes between any two calls
    std::vector<std::vector<uint32_t>> sys_writes; // per system: the columns it may write (one fresh version per AdvanceWorld)

    // ---- device buffers
    Block live;
    std::vector<Block> slots;            // ring slot pool
    std::vector<int> free_slots;
    uint64_t* d_parts = nullptr; uint32_t part_stride = 0;   // [(n_cks)+1][part_stride], last = counts
    uint64_t* d_results = nullptr; uint64_t* h_results = nullptr; uint32_t max_results = 0;
    // completion tags of the list's closing k_gen_finalize (one per workgroup, behind the results in the same pinned allocation): a blocking
    // call polls them instead of asking the runtime for the stream (read_back)
    static constexpr uint32_t SPIN_TAGS = 256;
    uint64_t* d_done = nullptr; volatile uint64_t* h_done = nullptr; uint64_t spin_seq = 0; uint32_t spin_n = 0; uint64_t spin_hits = 0, spin_misses = 0;
    UnitDesc* d_units = nullptr;
    uint64_t* d_maskoffs = nullptr;      // scratch for k_set_mask_range
    // spawn payloads: a ring of bytes in pinned, device-mapped memory (h_stage; d_hstage = the same bytes as the device sees them: fused spawns read
    // them zero-copy) with a device twin (d_stage) for the unfused spawn kernel.  [stage_tail, stage_used) (mod wrap) is what launches of
    // uncollected batches may still read; a collected batch frees everything up to its stage_end.  stage_gen counts resets: whoever remembers
    // an offset into the ring (a request list's payload dedup) forgets it when the generation changes
    uint8_t* d_stage = nullptr; uint8_t* h_stage = nullptr; uint8_t* d_hstage = nullptr; uint64_t stage_bytes = 0, stage_used = 0, stage_tail = 0, stage_gen = 0;

    // ---- checksum specs (device view)
    std::vector<uint32_t> cks_comp;      // checksummed component ids in id order
    CksArgs cks_args{};
    bool custom_hashers = false;         // some checksum spec is user source: only the generated kernel can compute it
    bool fused_ok = false;               // schedule == particles fast path
    bool fused_cks = false;              // ... and every checksum spec is covered by it
    int f_T = -1, f_V = -1, f_L = -1, f_spawn = -1; uint32_t f_tw = 0, f_vw = 0, f_lw = 0;
    bool f_cksT = false, f_cksV = false;
    float f_g[3] = {0, 0, 0};
    int n_cu = 256;
    uint64_t* d_gen_parts = nullptr; uint32_t gen_part_stride = 0;   // generated kernel: [saves][n_cks + 1][one row per 256-slot workgroup]
    bool gen_ok = false;                 // the generated kernel serves this world's request lists
    // fold-forward (host_groups.hpp): two row buffers in device memory, used alternately by consecutive launches; what the LAST launch left in
    // ff_rows[ff_cur] waits for the next launch (or k_ff_fold) to fold it into the pinned values + tags of the HostFold with id ff_pending_id
    uint64_t* d_ff_rows[2] = {nullptr, nullptr}; uint32_t ff_cur = 0;
    uint64_t self_fold_calls = 0;                           // blocking calls whose groups folded themselves (read_back: every 256th takes the stream wait)
    struct FfPending { bool valid = false; uint64_t id = 0, seq = 0; uint32_t buf = 0, nvals = 0, g = 0, stride = 0, istride = 1, split = 1; uint64_t out_off = 0; } ff_pending;   // nvals = rows x split
    uint64_t ff_next_id = 1, ff_done_id = 0, ff_seq = 0;    // ids are handed out in launch order; every id <= ff_done_id has a fold queued on the stream
    uint64_t ff_mark_id = 0;                                // the group whose fold the launch being issued carries (set by ff_attach, consumed by launch_jit)
    static constexpr uint32_t FF_EVENTS = 32;
    struct FfEvent { hipEvent_t ev = nullptr; uint64_t id = 0; } ff_events[FF_EVENTS];   // recorded behind the launch that folds group `id` (host_requests.hpp ff_mark_folded)

    // LAZY LIVE BLOCK (host_groups.hpp): the last group of the previous list ended  [.., Save(F), Advance]  and did not write the live block -- the
    // live world (frame F + 1) IS Advance(ring slot of F) until somebody needs its bytes: a list that opens with a LoadGameState never does (a SyncTest
    // session, a P2P session in steady rollback), everything else materialises it first (materialise_live: one small launch)
    struct LiveStale { bool valid = false; Block* src = nullptr; uint64_t len = 0; uint32_t dt_bits = 0, aux_bits = 0; int step_frame = 0, step_confirmed = 0;
                       unsigned char n_inputs = 0; unsigned char inputs[GGRS_MAX_PLAYERS * (GGRS_MAX_INPUT_BYTES + 1)] = {}; } live_stale;
    int lazy_live_on = 1;                // (ggrs_dbg_set_lazy_live: 0 = the A/B of profiles/r05h; 2 = every eligible list whatever its size and streak: the fuzzer)
    bool live_handed_out = false;        // ggrs_hip_live_state_ptr gave the block away: it is kept current from then on
    uint32_t load_open_streak = 0;       // consecutive request lists that opened with a LoadGameState
    uint64_t lazy_skips = 0, lazy_materialised = 0;

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
    struct PendingBatch { hipEvent_t ev; uint32_t first, count; std::vector<uint64_t> host; uint32_t n_folds = 0; uint64_t stage_end = 0; };
    // Host-side end of the checksum fold (generated kernel).  Small groups: the workgroups write their partial rows straight into pinned,
    // device-mapped host memory and the HOST finishes each Save (XOR of g rows + three hashes) when the batch is collected.  Fold-forward
    // groups (ff_id != 0): g == 1 -- the rows were folded on the device, the pinned ring holds one value per (Save, part) followed by one tag
    // each; collect waits for the tags (ff_seq) before it hashes.
    struct HostFold { uint32_t res_slot, n_saves, g, n_cks, members; uint64_t rows_off; uint64_t save_len[16]; uint64_t ff_id, ff_seq; };   // save_len[k]: RollbackOrdered::len at Save k (a fused spawn grows it inside a group)
    std::deque<HostFold> folds;          // in submission order; a PendingBatch owns the next n_folds of them
    uint64_t* h_rows = nullptr; uint64_t* d_rows = nullptr; uint64_t rows_cap = 0, rows_used = 0, rows_tail = 0;   // ring of partial rows
    hipEvent_t batch_ev = nullptr; bool batch_ev_attached = false;   // enqueue: the batch's event, offered to the list's last kernel launch (launch_jit)
    bool device_results_only = false;    // a consumer reads the result ring in stream order (ggrs_hip_fanout_*): every fold stays on the device
    // ... and wants the Checksum(u128)s of the list being enqueued in DEVICE memory too (k_gen_finalize's second copy): result slot dev_results_first + i
    // goes to dev_results_dst[2 i ..] -- the fan-out points this at the all-gather's send buffer for the duration of one step
    uint64_t* dev_results_dst = nullptr; uint32_t dev_results_first = 0;
    // SPECULATIVE BRANCH STATES (ggrs_hip_fanout_step_branches with GGRS_BRANCH_RETAIN_*): packed state blocks outside the ring, allocated on demand and kept
    // until the world goes -- an adopted branch state trades places with a ring slot (ggrs_hip_fanout_adopt), so both sets belong to the world
    std::vector<Block> spec_blocks; std::vector<void*> spec_allocs;
    uint64_t* d_branch_parts = nullptr; uint64_t branch_parts_cap = 0;      // partial rows of a member launch: [members x saves x (n_cks + 1)][workgroups]
    ggrs_world** fanout_backref = nullptr;   // the `w` field of the ggrs_fanout driving this world: cleared by world_destroy, so a fan-out object that outlives its world touches nothing
    std::deque<PendingBatch> pending; uint32_t res_head = 0; uint32_t pending_results = 0;
    std::vector<hipEvent_t> event_pool;

    // ---- profiling
    bool nt_copy = false;               // non-temporal loads/stores in k_copy_state (A/B knob)
    bool prof = false;
    std::vector<EventPair> prof_events;
    std::vector<hipEvent_t> prof_pool;   // timing events of drained pairs, reused: creating two events per launch kept the HOST behind the device in an instrumented pass
                                         // (idle gaps between ticks; the 1 M launch then read 49 us instead of 45: profiles/r05j)
    hipEvent_t prof_event() { if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; } hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e; }
    double prof_ms[GGRS_KERNEL_CLASSES] = {};
    uint64_t prof_n[GGRS_KERNEL_CLASSES] = {};
    uint64_t prof_skipped = 0;                                // bytes profiled launches did not store (value tags), already subtracted from prof_bytes
    uint64_t prof_bytes[GGRS_KERNEL_CLASSES] = {};            // algorithmic bytes the launches of a class were asked to move (rows x their extent)
    std::vector<float> prof_launch_us[GGRS_KERNEL_CLASSES];   // every launch since enable, in submission order (ggrs_hip_profile_read_launches)
    HostTimeline tl;

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
    ProfScope(ggrs_world* w_, uint32_t c, uint64_t bytes = 0) : w(w_), cls(c) {
        if (w->prof) { a = w->prof_event(); b = w->prof_event(); (void)hipEventRecord(a, w->stream); w->prof_bytes[c] += bytes; }
    }
    ~ProfScope() {
        if (w->prof) { (void)hipEventRecord(b, w->stream); w->prof_events.push_back({a, b, cls}); }
    }
};

// Every entry point that touches the device runs with the world's device current on the calling thread and puts the
// caller's device back on return (two worlds on different GPUs in one process; a host thread torch switched elsewhere).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(const ggrs_world* w) {
        if (hipGetDevice(&prev) == hipSuccess && prev != w->device) switched = hipSetDevice(w->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// ---- row versions ---------------------------------------------------------------------------------------------------
inline void ver_touch(ggrs_world* w, uint32_t col) { w->cur_ver[col] = ++w->ver_counter; }
// versions [n_columns, n_columns + n_components) belong to the presence masks (changed by the host only: spawn, insert, remove, load, adopt)
inline uint32_t ver_presence(const ggrs_world* w, uint32_t c) { return (uint32_t)w->col_off.size() + c; }
inline void ver_touch_comp(ggrs_world* w, uint32_t c) { for (uint32_t k = 0; k < w->comps[c].n_words; ++k) ver_touch(w, w->comps[c].col_base + k); ver_touch(w, ver_presence(w, c)); }
inline void ver_touch_all(ggrs_world* w) { for (uint32_t c = 0; c < w->cur_ver.size(); ++c) ver_touch(w, c); }
// AdvanceWorld: every registered system may have written its write set
inline void ver_step(ggrs_world* w) { for (auto& cols : w->sys_writes) for (uint32_t c : cols) ver_touch(w, c); }
// must `dst` receive column `col` to hold the state whose versions are `want`?  (yes also when nothing is known)
inline bool ver_differs(const ggrs_world* w, const Block& dst, const std::vector<ver_t>& want, uint32_t col) {
    return !w->knobs.row_versions || w->col_ext[col] || dst.ver[col] == VER_NONE || want[col] == VER_NONE || dst.ver[col] != want[col];
}
// the live block holds exactly the logical live state (no fused group is being assembled)
inline void ver_sync_live(ggrs_world* w) { w->live.ver = w->cur_ver; }
// the HOST (or a kernel that keeps no tags) wrote bytes of these live columns: their value tags no longer describe them
inline void live_tags_lost(ggrs_world* w, uint32_t col) { if (col < 64) w->live.tag_ok &= ~(1ull << col); }
inline void live_tags_lost_comp(ggrs_world* w, uint32_t c) { for (uint32_t k = 0; k < w->comps[c].n_words; ++k) live_tags_lost(w, w->comps[c].col_base + k); }
// presence masks of `dst` that differ from the ones `want` describes
inline uint32_t pmask_differs(const ggrs_world* w, const Block& dst, const std::vector<ver_t>& want) {
    uint32_t m = 0;
    for (uint32_t c = 0; c < w->comps.size(); ++c) {
        const uint32_t i = ver_presence(w, c);
        if (!w->knobs.row_versions || dst.ver[i] == VER_NONE || want[i] == VER_NONE || dst.ver[i] != want[i]) m |= 1u << c;
    }
    return m;
}

// Computes the packed state layout from the registered components.
void build_layout(ggrs_world* w) {
    const uint64_t mask_bytes = align_up(w->cap_pad / 8, ALIGN);
    uint64_t off = ALIGN;                          // header
    w->off_alive = off; off += mask_bytes;
    w->off_present.assign(w->comps.size(), 0); w->col_off.clear(); w->col_wb.clear();
    w->has_nr = false;
    uint32_t ncols = 0;
    w->has_strategy = false;
    for (auto& c : w->comps) { c.col_base = ncols; ncols += c.n_words; w->has_nr |= c.no_rollback; }
    w->n_tcols = ncols;
    // the Stored words of components under a Strategy: columns of their own behind the component words (ring slots use them, the live block does not)
    for (auto& c : w->comps) if (c.s_n_words && !c.no_rollback) { c.scol_base = ncols; ncols += c.s_n_words; w->has_strategy = true; }
    w->col_off.assign(ncols, 0); w->col_wb.assign(ncols, 4); w->col_rb.assign(ncols, 0);
    for (size_t c = 0; c < w->comps.size(); ++c) if (!w->comps[c].no_rollback) { w->off_present[c] = off; off += mask_bytes; }
    // rollback word columns, TILE-MAJOR: tile t of every column is contiguous (ts bytes per tile).  Inside a
    // tile the words that GgrsSchedule systems read or write come first, so a per-request AdvanceWorld kernel
    // streams one contiguous span per tile (particles: 32 of the 60 KiB) instead of 4 KiB pieces; then the other 4- and
    // 8-byte words, then the 2- and 1-byte ones (every column's tile stays 16-byte aligned).
    w->col_ts.assign(ncols, 0);
    std::vector<uint8_t> hot(ncols, 0);
    w->sys_writes.assign(w->systems.size(), {});
    for (size_t si = 0; si < w->systems.size(); ++si) {
        const ggrs_system_desc& sd = w->systems[si];
        auto mark = [&](uint32_t comp, uint32_t word, uint32_t span, bool writes) {
            if (comp >= w->comps.size()) return;
            for (uint32_t k = 0; k < span && word + k < w->comps[comp].n_words; ++k) {
                hot[w->comps[comp].col_base + word + k] = 1;
                if (writes) w->sys_writes[si].push_back(w->comps[comp].col_base + word + k);
            }
        };
        switch (sd.kind) {
        case GGRS_SYS_PARTICLES_UPDATE: mark(sd.comp[0], sd.word[0], 3, true); mark(sd.comp[1], sd.word[1], 3, true); break;
        case GGRS_SYS_TTL_DESPAWN: case GGRS_SYS_ADD_U32: case GGRS_SYS_SAT_SUB_DESPAWN: mark(sd.comp[0], sd.word[0], 1, true); break;
        case GGRS_SYS_BOX_MOVE: mark(sd.comp[0], sd.word[0], 3, true); mark(sd.comp[1], sd.word[1], 3, true); mark(sd.comp[2], sd.word[2], 1, false); break;
        case GGRS_SYS_CUSTOM: {
            const ggrs_world::Custom& c = w->customs[sd.comp[0]];
            for (uint32_t b = 0; b < c.n_bind; ++b) mark(c.comp[b], c.word[b], 1, true);      // a bound word may be written
        } break;
        default: break;     // spawn systems append rows: whoever runs them versions the bundle (run_spawn_systems / the fused path in host_groups.hpp)
        }
    }
    uint64_t tcol = 0;
    for (int pass = 0; pass < 4; ++pass)      // 0: hot words (any width >= 4), 1: other 4-/8-byte words, 2: 2-byte words, 3: 1-byte words
        for (auto& c : w->comps) {
            for (uint32_t k = 0; k < c.n_words + c.s_n_words; ++k) {
                const bool stored = k >= c.n_words;
                if (stored && c.no_rollback) continue;
                const uint32_t col = stored ? c.scol_base + (k - c.n_words) : c.col_base + k;
                const uint32_t wb = stored ? c.s_word_bytes : c.word_bytes;
                w->col_wb[col] = wb; w->col_rb[col] = !c.no_rollback && !stored;
                if (c.no_rollback) continue;
                const bool wide = wb >= 4;
                const int want = (wide && hot[col]) ? 0 : (wide ? 1 : (wb == 2 ? 2 : 3));
                if (want != pass) continue;
                w->col_off[col] = tcol;           // offset inside a tile for now
                tcol += (uint64_t)LAYOUT_TILE * wb;
            }
        }
    w->ts = (uint32_t)tcol;
    const uint64_t cols_base = align_up(off, 4096);
    for (auto& c : w->comps) if (!c.no_rollback) {
        for (uint32_t k = 0; k < c.n_words; ++k) { w->col_off[c.col_base + k] += cols_base; w->col_ts[c.col_base + k] = w->ts; }
        for (uint32_t k = 0; k < c.s_n_words; ++k) { w->col_off[c.scol_base + k] += cols_base; w->col_ts[c.scol_base + k] = w->ts; }
    }
    off = cols_base + (w->cap_pad / LAYOUT_TILE) * (uint64_t)w->ts;
    // value tags: one u32 per 64-slot unit and component word column, [unit][column]
    w->off_tags = align_up(off, ALIGN); w->tag_row_bytes = w->n_tcols * 4u;
    off = w->off_tags + (w->cap_pad / 64) * (uint64_t)w->tag_row_bytes;
    w->tag_cols = 0;
    for (auto& c : w->comps) if (!c.no_rollback && !c.s_n_words) for (uint32_t k = 0; k < c.n_words && c.col_base + k < 64; ++k) w->tag_cols |= 1ull << (c.col_base + k);
    w->state_bytes = align_up(off, 4096);
    // ---- live-only side region, placed right behind the ring blocks
    w->side_off = (uint64_t)(w->max_depth + 1) * w->state_bytes;
    uint64_t so = w->side_off;
    w->marks.off_disabled = so; so += mask_bytes;
    w->marks.off_dframe = so; so += align_up(w->cap_pad * 4, ALIGN);
    for (size_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback) { w->off_present[c] = so; so += mask_bytes; }
    for (auto& c : w->comps) {
        if (!c.no_rollback) continue;
        // live-only columns are plain arrays: the same addressing formula with tile stride = 8192 words
        for (uint32_t k = 0; k < c.n_words; ++k) { w->col_off[c.col_base + k] = so; w->col_ts[c.col_base + k] = LAYOUT_TILE * c.word_bytes; so += align_up(w->cap_pad * c.word_bytes, ALIGN); }
    }
    w->side_bytes = align_up(so - w->side_off, 4096);

    CopyPlan& p = w->plan;
    memset(&p, 0, sizeof p);
    w->row_col.clear();
    p.n_masks = 1;
    p.mask_off[0] = w->off_alive;
    uint32_t nr = 0;
    for (size_t c = 0; c < w->comps.size(); ++c) if (!w->comps[c].no_rollback) p.mask_off[p.n_masks++] = w->off_present[c];   // snapshots hold rollback components only
    // a row = up to 4 KiB of one workgroup tile (1024 slots) of one column: an 8-byte word has two, a 1- / 2-byte word a short
    // one.  The 4 KiB rows come first (k_copy_state moves them in straight-line batches), the short ones after them.
    for (int pass = 0; pass < 2; ++pass)
        for (size_t c = 0; c < w->comps.size(); ++c) {
            const Comp& cc = w->comps[c];
            if (cc.no_rollback || (cc.word_bytes >= 4) != (pass == 0)) continue;
            for (uint32_t k = 0; k < cc.n_words; ++k)
                for (uint32_t r = 0; r < std::max(1u, cc.word_bytes / 4); ++r) {
                    RowDesc& rd = p.row[nr++];
                    rd.col_off = w->col_off[cc.col_base + k]; rd.roff = r * 4096; rd.tile_stride = w->ts; rd.word_bytes = cc.word_bytes;
                    rd.bytes = std::min<uint32_t>(4096u, (uint32_t)TILE * cc.word_bytes);
                    w->row_col.push_back(cc.col_base + k);
                }
            if (pass == 0) p.n_wide = nr;
        }
    p.n_rows = nr;
    w->cur_ver.assign(ncols + w->comps.size(), 0); w->col_ext.assign(ncols, 0);
}

uint32_t total_rows(const ggrs_world* w) {
    uint32_t n = 0;
    for (auto& c : w->comps) if (!c.no_rollback) n += c.n_words * std::max(1u, c.word_bytes / 4);
    return n;
}
// bytes per slot of the rollback words
inline uint32_t bytes_per_slot(const ggrs_world* w) { return w->ts / LAYOUT_TILE; }

inline uint32_t tiles_for(uint64_t n) { return (uint32_t)((n + TILE - 1) / TILE); }

// bytes per slot of the columns some system may write: what a steady SaveWorld moves
inline uint64_t rows_bytes_hot(const ggrs_world* w) {
    uint64_t b = 0, seen = 0;
    for (size_t si = 0; si < w->sys_writes.size(); ++si) for (uint32_t c : w->sys_writes[si]) if (c < 64 && !((seen >> c) & 1ull)) { seen |= 1ull << c; b += w->col_wb[c]; }
    return b;
}
// does this world's kernel keep value tags?  (what a layout-only world -- `make aot` -- can tell as well)
inline bool vtags_policy(const ggrs_world* w) {
    return w->knobs.row_versions && w->tag_cols && (w->vtags_mode == 1 || (w->vtags_mode < 0 && rows_bytes_hot(w) * w->capacity >= VTAGS_MIN_BYTES));
}

}  // namespace
