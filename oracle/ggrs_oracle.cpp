// ggrs_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A CPU restatement of bevy_ggrs's snapshot-and-resimulate hot path (reference
// mounted at /root/reference, v0.22.0).  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load this library, and only as the
// checker / reported CPU baseline -- never as the thing shipped or measured as
// the product.  The product path is bevy_ggrs_amd/csrc (HIP, gfx950).
//
// PARITY PINNING: the reference is Rust and cannot be built here (no cargo, no
// vendored bevy/ggrs/seahash), and its own tests assert no absolute checksum or
// f32 value.  What IS pinned:
//   * SeaHash arithmetic: seahash 4.1's published vector
//       hash("to be or not to be") == 1988685042348123509
//     reproduced by BOTH the stream hasher and the 4-lane buffer hasher below,
//     which are two independent formulations that must agree for all lengths.
//   * Snapshot ring: the 11 known-answer unit tests of
//     src/snapshot/mod.rs:349-512 are mirrored 1:1 in tests/test_ring_kat.py.
//   * ggrs SyncTest request order and absolute checksum values: PARITY UNPINNED
//     by reference-supplied vectors (ggrs is an un-vendored git dependency,
//     Cargo.toml:23); restated from its published algorithm.
//   * The WHOLE tick (save / load / entity reconcile / ring / deferred despawn /
//     systems / checksums) is cross-checked bit for bit against an
//     independently written second restatement, oracle/twin_np.py (numpy,
//     dict-of-RollbackId snapshots, deque ring), on BASELINE configs 1-4:
//     tests/test_twin_oracle.py.  That is agreement of two restatements, not a
//     reference-produced vector: parity stays "unpinned by the reference".
//
// Two storage back-ends with identical observable behaviour:
//   mode 0 FLAT      : SoA columns + memcpy ring (best-case CPU layout)
//   mode 1 REFSHAPED : the reference's cost structure -- per save a fresh
//                      RollbackId->value hash map per registered component,
//                      per load one lookup per entity, order() lookups in the
//                      checksum, entity map + RollbackOrdered clone per save.
//                      This is the honest CPU baseline ("port") that bench.py times.
//
// Build: g++ -O3 -march=native -fno-fast-math -ffp-contract=off -shared -fPIC
//        (f32 results must be bit-identical to Rust semantics: no FMA contraction).

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <deque>
#include <string>
#include <map>
#include <vector>
#include <memory>
#include <chrono>

#if defined(_OPENMP)
#include <omp.h>
#endif

// ---------------------------------------------------------------------------
// SeaHash 4.1 (third-party crate, NOT under /root/reference; Cargo.toml:24).
// Restated from the crate's published algorithm.  Call sites in the reference:
// src/snapshot/mod.rs:318-320 (checksum_hasher), checksum.rs:38-44,
// component_checksum.rs:44-48,77-95, entity_checksum.rs:35-43.
// ---------------------------------------------------------------------------
namespace sea {

static const uint64_t P = 0x6eed0e9da4d94a4fULL;
static const uint64_t K0 = 0x16f11fe89b0d677cULL, K1 = 0xb480a793d8e6c86cULL,
                      K2 = 0x6fe2e5aaf078ebc9ULL, K3 = 0x14f994a4c5259381ULL;

static inline uint64_t diffuse(uint64_t x) {
    x *= P;
    x ^= (x >> 32) >> (x >> 60);
    x *= P;
    return x;
}

// Stream hasher (seahash::SeaHasher): 4-word rotating state, 8-byte tail.
struct Hasher {
    uint64_t s0 = K0, s1 = K1, s2 = K2, s3 = K3;
    uint64_t written = 0;
    uint64_t tail = 0;
    unsigned ntail = 0;

    inline void push(uint64_t x) {
        uint64_t a = diffuse(s0 ^ x);
        s0 = s1; s1 = s2; s2 = s3; s3 = a;
        written += 8;
    }
    void write(const uint8_t* b, size_t n) {
        // fill the tail first
        while (n > 0 && ntail > 0) {
            tail |= (uint64_t)(*b) << (8 * ntail);
            ++ntail; ++b; --n;
            if (ntail == 8) { push(tail); tail = 0; ntail = 0; }
        }
        while (n >= 8) {
            uint64_t w; memcpy(&w, b, 8);   // little-endian host
            push(w); b += 8; n -= 8;
        }
        while (n > 0) {
            tail |= (uint64_t)(*b) << (8 * ntail);
            ++ntail; ++b; --n;
            // ntail < 8 here by construction
        }
    }
    inline void write_u64(uint64_t v) { uint8_t b[8]; memcpy(b, &v, 8); write(b, 8); }
    inline void write_u32(uint32_t v) { uint8_t b[4]; memcpy(b, &v, 4); write(b, 4); }
    inline void write_u8(uint8_t v) { write(&v, 1); }
    inline uint64_t finish() const {
        uint64_t a = ntail > 0 ? diffuse(s0 ^ tail) : s0;
        // Rust precedence: `a ^ s1 ^ s2 ^ s3 ^ written + ntail` == a^s1^s2^s3^(written+ntail)
        return diffuse(a ^ s1 ^ s2 ^ s3 ^ (written + (uint64_t)ntail));
    }
};

// One-shot buffer hasher (seahash::hash): 4 independent lanes.  Used only to
// cross-validate the stream formulation (they must agree for every length).
static uint64_t hash_buffer(const uint8_t* buf, size_t len) {
    uint64_t a = K0, b = K1, c = K2, d = K3;
    size_t i = 0;
    auto rd = [&](size_t off, size_t n) { uint64_t w = 0; memcpy(&w, buf + off, n); return w; };
    while (len - i >= 32) {
        a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8));
        c = diffuse(c ^ rd(i + 16, 8)); d = diffuse(d ^ rd(i + 24, 8));
        i += 32;
    }
    size_t rem = len - i;
    if (rem >= 25)      { a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8)); c = diffuse(c ^ rd(i + 16, 8)); d = diffuse(d ^ rd(i + 24, rem - 24)); }
    else if (rem == 24) { a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8)); c = diffuse(c ^ rd(i + 16, 8)); }
    else if (rem >= 17) { a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8)); c = diffuse(c ^ rd(i + 16, rem - 16)); }
    else if (rem == 16) { a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, 8)); }
    else if (rem >= 9)  { a = diffuse(a ^ rd(i, 8)); b = diffuse(b ^ rd(i + 8, rem - 8)); }
    else if (rem == 8)  { a = diffuse(a ^ rd(i, 8)); }
    else if (rem >= 1)  { a = diffuse(a ^ rd(i, rem)); }
    a ^= b; c ^= d; a ^= c; a ^= (uint64_t)len;
    return diffuse(a);
}

}  // namespace sea

// ---------------------------------------------------------------------------
// GgrsSnapshots<For, As> ring -- src/snapshot/mod.rs:97-243.
// Newest at the front, oldest at the back; frames kept in a parallel deque.
// ---------------------------------------------------------------------------
template <class As>
struct Ring {
    std::deque<As> snapshots;
    std::deque<int32_t> frames;
    size_t depth = 60;   // DEFAULT_FPS, mod.rs:115

    void set_depth(size_t d) { depth = d; }   // mod.rs:123-138

    // mod.rs:147-181
    void push(int32_t frame, As snap) {
        while (!frames.empty()) {
            int32_t current = frames.front();
            // i32::abs_diff -> u32
            uint32_t ad = current >= frame ? (uint32_t)current - (uint32_t)frame
                                           : (uint32_t)frame - (uint32_t)current;
            bool wrapped = ad > (UINT32_MAX / 2);
            bool current_after_frame = current >= frame && !wrapped;
            bool current_after_frame_wrapped = frame >= current && wrapped;
            if (current_after_frame || current_after_frame_wrapped) {
                snapshots.pop_front(); frames.pop_front();
            } else break;
        }
        snapshots.push_front(std::move(snap));
        frames.push_front(frame);
        while (snapshots.size() > depth) { snapshots.pop_back(); frames.pop_back(); }
    }
    // mod.rs:185-202
    void confirm(int32_t confirmed) {
        while (!frames.empty() && frames.back() < confirmed) { snapshots.pop_back(); frames.pop_back(); }
    }
    // mod.rs:210-226; returns false where the reference panics (ring is left empty, as there)
    bool rollback(int32_t frame) {
        for (;;) {
            if (frames.empty()) return false;
            if (frames.front() != frame) { snapshots.pop_front(); frames.pop_front(); }
            else return true;
        }
    }
    As* get() { return snapshots.empty() ? nullptr : &snapshots.front(); }   // mod.rs:229-233
    As* peek(int32_t frame) {                                                 // mod.rs:236-243
        for (size_t i = 0; i < frames.size(); ++i) if (frames[i] == frame) return &snapshots[i];
        return nullptr;
    }
};

// ---------------------------------------------------------------------------
// World model
// ---------------------------------------------------------------------------
enum { MAX_COMPS = 32, MAX_WORDS = 16, MAX_UNITS = 32, MAX_SYSTEMS = 16 };

enum SystemKind : uint32_t {
    SYS_PARTICLES_UPDATE = 1,   // examples/stress_tests/particles.rs:272-280
    SYS_TTL_DESPAWN = 2,        // particles.rs:282-289
    SYS_PARTICLES_SPAWN = 3,    // particles.rs:254-270
    SYS_ADD_U32 = 4,            // benches/bench.rs:30-46, tests/component_rollback.rs:24-28
    SYS_SAT_SUB_DESPAWN = 5,    // tests/synctest.rs:37-44
    SYS_BOX_MOVE = 6,           // examples/box_game/box_game.rs:154-206
    SYS_CUSTOM = 7,             // any per-entity GgrsSchedule system the test hands in as a C callback (lib.rs:76, 247-251)
    SYS_SPAWN_CUSTOM = 8,       // a system that spawns Rollback entities (rollback.rs:45-59), as a C callback
};

// What a user-written system sees of the frame: Time<GgrsTime>::delta_secs, RollbackFrameCount, PlayerInputs<T> (src/lib.rs:98:
// (T::Input, InputStatus) per player -- `inputs` = n_inputs x input_bytes bytes, `status` = n_inputs bytes, never NULL here), the system's constants
struct FrameView { float dt; int32_t frame; uint32_t n_inputs, input_bytes; const uint8_t* inputs; const uint8_t* status; float fparam[4]; int64_t iparam[2]; };
// words: the bound words of ONE entity widened to u64 (written back narrowed to the word's width); *kill = 1: despawn(), 2: despawn_rollback()
typedef void (*CustomSysFn)(uint64_t* words, uint64_t slot, const FrameView* f, int32_t* kill, void* user);
// the k-th entity of this frame's spawn: words = the bound words, holding the bundle's registered defaults; payload = the request's blob (+ k * stride)
typedef void (*SpawnSysFn)(uint64_t* words, uint64_t slot, uint64_t k, const FrameView* f, const uint8_t* payload, void* user);
// Strategy::store / Strategy::load (strategy.rs:22-40) over words widened to u64; `target` of load arrives zeroed
typedef void (*StoreFn)(const uint64_t* target, uint64_t* stored, void* user);
typedef void (*LoadFn)(const uint64_t* stored, uint64_t* target, void* user);
struct CustomSys { CustomSysFn fn = nullptr; void* user = nullptr; uint32_t n_bind = 0, comp[8] = {}, word[8] = {}; };
struct SpawnSys { SpawnSysFn fn = nullptr; void* user = nullptr; uint64_t bundle_mask = 0; uint32_t payload_stride = 0, n_bind = 0, comp[8] = {}, word[8] = {}; };

struct SystemDesc {            // must match include/ggrs_hip.h ggrs_system_desc
    uint32_t kind;
    uint32_t comp[4];
    uint32_t word[4];
    int64_t iparam[2];
    float fparam[4];
};

struct Comp {
    std::string name;
    uint32_t word_bytes = 4, n_words = 0;
    std::vector<uint32_t> cks_units;          // words fed to the inner SeaHash, in order (each written as word_bytes little-endian bytes)
    // checksum_component::<T>(fn(&T) -> u64) with an arbitrary hasher (rollback_app.rs:119-121): the test hands in a C callback
    uint64_t (*cks_fn)(const uint8_t* words, uint64_t slot, void* user) = nullptr;
    void* cks_user = nullptr;
    bool checksummed = false;
    std::vector<uint8_t> defaults;            // n_words*word_bytes default value (zeros unless set)
    bool no_rollback = false;                 // not registered for rollback: outside every snapshot (despawn.rs:3-6)
    // ComponentSnapshotPlugin<S: Strategy> with a user-written S (strategy.rs:22-40): snapshots hold Stored = s_n_words words of s_word_bytes
    uint32_t s_word_bytes = 0, s_n_words = 0;
    StoreFn store_fn = nullptr; LoadFn load_fn = nullptr; void* strat_user = nullptr;
};

static inline bool bit(const std::vector<uint64_t>& m, uint64_t i) { return (m[i >> 6] >> (i & 63)) & 1ULL; }
static inline void setbit(std::vector<uint64_t>& m, uint64_t i, bool v) {
    if (v) m[i >> 6] |= 1ULL << (i & 63); else m[i >> 6] &= ~(1ULL << (i & 63));
}

// ---- flat snapshot: whole-world copy (all per-type rings of the reference move in
// lock-step -- same push/rollback/confirm/depth -- so one ring of world snapshots is
// equivalent; mod.rs:340-345, component_snapshot.rs:133-146, entity.rs:103-118) ----
struct FlatSnap {
    std::vector<std::vector<uint64_t>> stored;  // per component under a Strategy: len * s_n_words Stored words (narrowed to s_word_bytes), else empty
    std::vector<std::vector<uint8_t>> cols;   // per (comp,word) column, len*word_bytes bytes
    std::vector<uint64_t> alive;
    std::vector<std::vector<uint64_t>> present;
    uint64_t len = 0;
};

// ---- reference-shaped storage ------------------------------------------------
// Open-addressing table standing in for hashbrown::HashMap<RollbackId, V> with a
// cheap fold-multiply hash (bevy_platform FixedHasher = foldhash).  Pre-sized like
// `collect()` from an exact-size iterator (mod.rs:293-298): one allocation, no rehash.
struct RefTable {
    uint32_t stride = 0;            // value bytes
    uint64_t mask = 0, count = 0;
    std::vector<uint64_t> keys;     // 0 == empty (ids are stored +1)
    std::vector<uint8_t> vals;
    static inline uint64_t h(uint64_t k) {
        __uint128_t m = (__uint128_t)(k ^ 0x243f6a8885a308d3ULL) * 0x9e3779b97f4a7c15ULL;
        return (uint64_t)m ^ (uint64_t)(m >> 64);
    }
    void init(uint64_t n, uint32_t stride_) {
        stride = stride_;
        uint64_t cap = 16;
        while (cap * 7 / 8 < n) cap <<= 1;
        mask = cap - 1; count = 0;
        keys.assign(cap, 0);
        vals.resize(cap * (size_t)stride);
    }
    inline void insert(uint64_t id, const void* v) {
        uint64_t i = h(id) & mask;
        while (keys[i] != 0 && keys[i] != id + 1) i = (i + 1) & mask;
        if (keys[i] == 0) ++count;
        keys[i] = id + 1;
        if (stride) memcpy(&vals[i * (size_t)stride], v, stride);
    }
    inline const uint8_t* get(uint64_t id) const {
        if (keys.empty()) return nullptr;
        uint64_t i = h(id) & mask;
        while (keys[i] != 0) {
            if (keys[i] == id + 1) return stride ? &vals[i * (size_t)stride] : (const uint8_t*)&keys[i];
            i = (i + 1) & mask;
        }
        return nullptr;
    }
};

struct RefSnap {
    std::vector<RefTable> comp;       // GgrsComponentSnapshot<C> per registered component
    RefTable entities;                // GgrsComponentSnapshot<Entity> (entity.rs:39-51)
    RefTable order_clone;             // RollbackOrdered.order clone (mod.rs:342, rollback.rs:62-66)
    std::vector<uint64_t> sorted_clone;
    uint64_t len = 0;
};

struct World {
    int mode = 0;
    uint64_t capacity = 0;
    std::vector<Comp> comps;
    std::vector<SystemDesc> systems;
    std::vector<CustomSys> customs;            // SYS_CUSTOM: systems[i].comp[0] indexes this
    std::vector<SpawnSys> spawn_customs;       // SYS_SPAWN_CUSTOM: likewise
    // `commands.spawn((.., Rollback))` from inside a user-written system, as many as the entity's data says (rollback.rs:45-59): the requests of the frame being
    // advanced -- parent slot -> {children, the parent's bound words as its system call left them} -- applied after the frame's systems, parents in RollbackOrdered order
    struct SpawnReq { uint32_t n; uint64_t words[8]; };
    std::map<uint64_t, SpawnReq> spawn_reqs;
    uint32_t input_bytes = 1, max_players = 16;   // PlayerInputs<T>: size_of::<T::Input>()
    bool sealed = false;

    // live state (FLAT layout is authoritative in both modes for download/compare;
    // REFSHAPED keeps AoS "archetype columns" + id maps and mirrors into these on demand)
    std::vector<std::vector<uint8_t>> cols;        // flat index = col_base[c] + w
    std::vector<uint32_t> col_base;
    std::vector<uint64_t> alive;
    std::vector<std::vector<uint64_t>> present;
    uint64_t len = 0;                               // RollbackOrdered.len(): ids ever spawned
    // RollbackDespawned(frame) markers (despawn.rs:45-46): a disabled entity still exists (keeps its
    // non-rollback components) but no default query, snapshot or checksum sees it: alive bit 0.
    std::vector<uint64_t> disabled;
    std::vector<int32_t> dframe;
    int32_t dc_local = 0;                           // Local<ConfirmedFrameCount> of despawn_confirmed_entities (despawn.rs:92)

    int32_t frame = 0;                              // RollbackFrameCount (mod.rs:70)
    bool has_confirmed = false; int32_t confirmed = 0;  // ConfirmedFrameCount (mod.rs:80)
    uint64_t fps = 60;                              // RollbackFrameRate (time.rs:20)

    Ring<FlatSnap> ring;
    Ring<RefSnap> rring;

    // REFSHAPED live mirrors
    std::vector<std::vector<uint8_t>> aos;          // per comp: capacity * stride bytes (AoS rows)
    RefTable order_map;                             // RollbackOrdered.order: id -> index
    std::vector<uint64_t> sorted;                   // RollbackOrdered.sorted

    std::string err;

    uint32_t stride(uint32_t c) const { return comps[c].word_bytes * comps[c].n_words; }
    void seal() {
        if (sealed) return;
        sealed = true;
        col_base.clear(); cols.clear();
        for (auto& c : comps) {
            col_base.push_back((uint32_t)cols.size());
            for (uint32_t w = 0; w < c.n_words; ++w) cols.emplace_back((size_t)capacity * c.word_bytes, 0);
        }
        alive.assign((capacity + 63) / 64, 0);
        disabled.assign((capacity + 63) / 64, 0);
        dframe.assign(capacity, 0);
        present.assign(comps.size(), std::vector<uint64_t>((capacity + 63) / 64, 0));
        if (mode == 1) {
            aos.clear();
            for (uint32_t c = 0; c < comps.size(); ++c) aos.emplace_back((size_t)capacity * stride(c), 0);
            order_map.init(capacity, 8);
            sorted.reserve(capacity);
        }
    }
    // RollbackId of slot i.  REFSHAPED scrambles it so that table probe order is not
    // the slot order (Bevy Entity bits are not dense indices either).
    static inline uint64_t rid(uint64_t slot) { return slot * 0x9E3779B97F4A7C15ULL + 0x7f4a7c15ULL; }
};

// time.rs:63-87 + Duration::as_secs_f32: delta of Time<GgrsTime> when advancing INTO `frame`.
static uint32_t dt_bits_for_frame(uint64_t fps, int32_t frame) {
    uint64_t f = (uint64_t)(int64_t)frame;                 // `frame.0 as u64`
    uint64_t rt1 = f * 1000000000ULL / fps;
    uint64_t rt0 = (f - 1) * 1000000000ULL / fps;          // elapsed restored/left at frame-1
    uint64_t d = rt1 - rt0;
    uint64_t secs = d / 1000000000ULL; uint32_t nanos = (uint32_t)(d % 1000000000ULL);
    volatile float a = (float)secs;
    volatile float b = (float)nanos / (float)1000000000u;  // as_secs_f32 = secs as f32 + nanos as f32 / 1e9 as f32
    float r = a + b;
    uint32_t bits; memcpy(&bits, &r, 4);
    return bits;
}

// custom_hasher(component) -- SeaHash stream over the selected u32 units
// (particles.rs:107-120 Velocity Hash impl, particles.rs:207-222 Transform closure,
// component_checksum.rs:44-48 default_hasher).  Units are 4-byte writes; a u64 field is
// two consecutive units (little-endian), which is byte-identical to one write_u64.
static inline uint64_t inner_hash_units(const uint32_t* u, uint32_t n) {
    sea::Hasher h;
    for (uint32_t k = 0; k < n; ++k) h.write_u32(u[k]);
    return h.finish();
}

// component_checksum.rs:77-95: per entity hash(order, custom) XOR-folded, then hashed once more.
static inline uint64_t entity_part(uint64_t order, uint64_t inner) {
    sea::Hasher h;
    h.write_u64(order);
    h.write_u64(inner);
    return h.finish();
}
static inline uint64_t finalize_part(uint64_t x) {
    sea::Hasher h;
    h.write_u64(x);
    return h.finish();
}

// custom_hasher(&component): the registered word list through SeaHasher, each word as word_bytes little-endian bytes -- what
// derive(Hash) / the particles closure write (a bool or u8 enum 1 byte, f32::to_bits 4, usize 8) -- or the user's function.
// `row`: the component's words of one entity, back to back (n_words * word_bytes bytes).
static inline uint64_t custom_hash_of(const Comp& cc, const uint8_t* row, uint64_t slot) {
    if (cc.cks_fn) return cc.cks_fn(row, slot, cc.cks_user);
    sea::Hasher h;
    for (uint32_t wi : cc.cks_units) h.write(row + (size_t)wi * cc.word_bytes, cc.word_bytes);
    return h.finish();
}

static uint64_t component_checksum_flat(const World& w, uint32_t c) {
    const Comp& cc = w.comps[c];
    uint64_t result = 0;
    const int64_t L = (int64_t)w.len;
#pragma omp parallel for reduction(^ : result) schedule(static) if (L > 65536 && !cc.cks_fn)
    for (int64_t ii = 0; ii < L; ++ii) {
        uint64_t i = (uint64_t)ii;
        if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
        uint8_t row[MAX_WORDS * 8];
        for (uint32_t k = 0; k < cc.n_words; ++k) memcpy(row + (size_t)k * cc.word_bytes, &w.cols[w.col_base[c] + k][i * cc.word_bytes], cc.word_bytes);
        result ^= entity_part(i /* RollbackOrdered.order == slot */, custom_hash_of(cc, row, i));
    }
    return finalize_part(result);
}

static uint64_t active_count_flat(const World& w) {
    uint64_t n = 0;
    for (uint64_t k = 0; k < w.alive.size(); ++k) n += (uint64_t)__builtin_popcountll(w.alive[k]);
    return n;
}

// entity_checksum.rs:29-52
static uint64_t entity_checksum(uint64_t active, uint64_t total) {
    sea::Hasher h;
    h.write_u64(active);
    h.write_u64(total);
    return h.finish();
}

// ---------------- REFSHAPED mirrors ----------------
static void ref_sync_from_flat(World& w, uint64_t first, uint64_t count) {
    // copy flat columns -> AoS rows for [first, first+count)
    for (uint32_t c = 0; c < w.comps.size(); ++c) {
        const Comp& cc = w.comps[c];
        uint32_t st = w.stride(c);
        for (uint64_t i = first; i < first + count; ++i)
            for (uint32_t k = 0; k < cc.n_words; ++k)
                memcpy(&w.aos[c][i * st + k * cc.word_bytes], &w.cols[w.col_base[c] + k][i * cc.word_bytes], cc.word_bytes);
    }
}
static void ref_sync_to_flat(World& w) {
    for (uint32_t c = 0; c < w.comps.size(); ++c) {
        const Comp& cc = w.comps[c];
        uint32_t st = w.stride(c);
        for (uint64_t i = 0; i < w.len; ++i)
            for (uint32_t k = 0; k < cc.n_words; ++k)
                memcpy(&w.cols[w.col_base[c] + k][i * cc.word_bytes], &w.aos[c][i * st + k * cc.word_bytes], cc.word_bytes);
    }
}

static uint64_t component_checksum_ref(const World& w, uint32_t c) {
    const Comp& cc = w.comps[c];
    uint64_t result = 0;
    const uint32_t st = w.stride(c);
    for (uint64_t i = 0; i < w.len; ++i) {          // Query<(&RollbackId,&C)>::iter()
        if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
        uint64_t id = World::rid(i);
        const uint8_t* op = w.order_map.get(id);    // rollback_ordered.order(rollback): HashMap lookup
        uint64_t order; memcpy(&order, op, 8);
        const uint8_t* row = &w.aos[c][i * st];
        result ^= entity_part(order, custom_hash_of(cc, row, i));
    }
    return finalize_part(result);
}

// ---------------- RollbackDespawned (snapshot/despawn.rs) ----------------
// EntityCommands::despawn_rollback, despawn.rs:114-143 (a queued command: RollbackFrameCount is the
// frame being simulated).  Unconfirmed frame -> insert RollbackDespawned(frame): the entity is
// disabled; otherwise a plain despawn.  (Children recursion: this path has no hierarchies.)
static inline void despawn_rollback_one(World& w, uint64_t i) {
    if (!bit(w.alive, i)) return;
    setbit(w.alive, i, false);
    if (w.confirmed < w.frame) { setbit(w.disabled, i, true); w.dframe[i] = w.frame; }
}
// LoadWorldSystems::EntityResurrect (resurrect_entities, despawn.rs:69-87) followed by the effect
// EntitySnapshotPlugin::load's reconcile (entity.rs:55-99) has on NON-rollback components: an
// entity that survives (live, in the snapshot) or stays disabled keeps them; one that is despawned
// or re-created with a fresh Entity does not have them any more.  `snap_alive(i)` = the snapshot
// holds RollbackId i.  Must run BEFORE the mode-specific load touches w.alive.
template <class SnapAlive>
static void resurrect_and_reconcile(World& w, uint64_t snap_len, SnapAlive snap_alive) {
    const uint64_t hi = snap_len > w.len ? snap_len : w.len;
    for (uint64_t i = 0; i < hi; ++i) {
        bool exists = bit(w.alive, i);
        if (bit(w.disabled, i) && w.dframe[i] > w.frame) {          // despawned_frame > rollback_frame
            setbit(w.disabled, i, false); setbit(w.alive, i, true); exists = true;
        }
        const bool keep = (exists && i < snap_len && snap_alive(i)) || bit(w.disabled, i);
        if (!keep) for (uint32_t c = 0; c < w.comps.size(); ++c) if (w.comps[c].no_rollback) setbit(w.present[c], i, false);
    }
}
// AdvanceWorldSystems::DespawnConfirmed (despawn_confirmed_entities, despawn.rs:89-112)
static void despawn_confirmed(World& w) {
    if (w.confirmed == w.dc_local) return;                          // "No work necessary"
    w.dc_local = w.confirmed;
    for (uint64_t i = 0; i < w.len; ++i)
        if (bit(w.disabled, i) && w.dframe[i] <= w.confirmed) setbit(w.disabled, i, false);   // world.despawn(entity)
}

// ---------------- systems ----------------
static inline float f32_of(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t bits_of(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

struct AdvanceArgs {
    uint32_t dt_bits;
    const uint8_t* inputs; uint32_t n_inputs;
    uint64_t spawn_count; const float* spawn_vx; const float* spawn_vy;
    const uint8_t* status = nullptr;                       // InputStatus per player; NULL: all Confirmed
    const uint8_t* spawn_payload = nullptr; uint64_t spawn_payload_bytes = 0;
};
static inline uint64_t word_mask(uint32_t wb) { return wb >= 8 ? ~0ULL : ((1ULL << (8 * wb)) - 1ULL); }
static inline uint64_t load_word(const World& w, uint32_t c, uint32_t k, uint64_t i) {
    uint64_t v = 0; memcpy(&v, &w.cols[w.col_base[c] + k][i * w.comps[c].word_bytes], w.comps[c].word_bytes); return v;
}
static inline void store_word(World& w, uint32_t c, uint32_t k, uint64_t i, uint64_t v) {
    memcpy(&w.cols[w.col_base[c] + k][i * w.comps[c].word_bytes], &v, w.comps[c].word_bytes);
}
static void ref_sync_to_flat(World& w);
static void ref_sync_from_flat(World& w, uint64_t first, uint64_t count);
static inline void despawn_rollback_one(World& w, uint64_t i);
static FrameView frame_view(const World& w, const AdvanceArgs& a, const SystemDesc& s, const uint8_t* zero_status) {
    FrameView f; memset(&f, 0, sizeof f);
    memcpy(&f.dt, &a.dt_bits, 4); f.frame = w.frame; f.n_inputs = a.n_inputs; f.input_bytes = w.input_bytes;
    f.inputs = a.inputs; f.status = a.status ? a.status : zero_status;
    for (int k = 0; k < 4; ++k) f.fparam[k] = s.fparam[k];
    f.iparam[0] = s.iparam[0]; f.iparam[1] = s.iparam[1];
    return f;
}
// a user-written per-entity system: Query<(&mut A, &mut B, ..), With<Rollback>> + Commands, one entity at a time, in slot order
static void run_custom_system(World& w, const AdvanceArgs& a, const SystemDesc& s) {
    static const uint8_t zero_status[16] = {0};
    const CustomSys& cs = w.customs[s.comp[0]];
    const FrameView f = frame_view(w, a, s, zero_status);
    if (w.mode == 1) ref_sync_to_flat(w);
    for (uint64_t i = 0; i < w.len; ++i) {
        if (!bit(w.alive, i)) continue;
        bool has = true;
        for (uint32_t b = 0; b < cs.n_bind; ++b) has = has && bit(w.present[cs.comp[b]], i);
        if (!has) continue;
        uint64_t words[8]; int32_t kill = 0;
        for (uint32_t b = 0; b < cs.n_bind; ++b) words[b] = load_word(w, cs.comp[b], cs.word[b], i);
        cs.fn(words, i, &f, &kill, cs.user);
        for (uint32_t b = 0; b < cs.n_bind; ++b) store_word(w, cs.comp[b], cs.word[b], i, words[b]);
        // (the callback's int carries two things: bits 0..7 the despawn request -- 0 none, 1 despawn(), 2 despawn_rollback() --, bits 8..15 how many Rollback entities
        // this call spawns)
        const uint32_t n_spawn = ((uint32_t)kill >> 8) & 0xFFu; kill &= 0xFF;
        if (n_spawn) { World::SpawnReq rq; rq.n = n_spawn; memset(rq.words, 0, sizeof rq.words); for (uint32_t b = 0; b < cs.n_bind; ++b) rq.words[b] = words[b]; w.spawn_reqs[i] = rq; }
        if (kill == 2) despawn_rollback_one(w, i); else if (kill) setbit(w.alive, i, false);
    }
    if (w.mode == 1) ref_sync_from_flat(w, 0, w.len);
}

template <class GetW, class SetW>
static inline void particles_update_one(float dt, const float g[3], GetW get, SetW set) {
    // particles.rs:272-280:  **velocity += gravity * time_step; translation += **velocity * time_step;
    // glam Vec3 ops are component-wise mul then add, no FMA.
    for (int k = 0; k < 3; ++k) {
        float gd = g[k] * dt;          // built with -ffp-contract=off: never fused
        float v = get(1, k) + gd;
        float vd = v * dt;
        float x = get(0, k) + vd;
        set(1, k, v);
        set(0, k, x);
    }
}


// move_cube_system, examples/box_game/box_game.rs:154-206.  fparam = {ACCELERATION, MAX_SPEED, FRICTION,
// half_width}; INPUT_* bits box_game.rs:13-16.  Written in the reference's statement order; the build has
// -ffp-contract=off so `a*b+c` is never fused (Rust semantics).  FRICTION.powf(dt) -> libm powf (what
// f32::powf lowers to).  Vec3::clamp_length_max (glam, un-vendored): length_squared = x*x + y*y + z*z;
// `if length_sq > max*max { max * (self / sqrt(length_sq)) }`.  f32::clamp: `if x < min {min}; if x > max {max}`.
template <class GetW, class SetW>
static inline void box_move_one(float dt, const float p[4], uint8_t input, GetW get, SetW set) {
    const float ACCELERATION = p[0], MAX_SPEED = p[1], FRICTION = p[2], half_width = p[3];
    float vx = get(1, 0), vy = get(1, 1), vz = get(1, 2);
    const bool up = input & 1, down = input & 2, left = input & 4, right = input & 8;
    if (up && !down) vz -= ACCELERATION * dt;
    if (!up && down) vz += ACCELERATION * dt;
    if (left && !right) vx -= ACCELERATION * dt;
    if (!left && right) vx += ACCELERATION * dt;
    if (!up && !down) vz *= powf(FRICTION, dt);
    if (!left && !right) vx *= powf(FRICTION, dt);
    vy *= powf(FRICTION, dt);
    const float length_sq = (vx * vx + vy * vy) + vz * vz;
    if (length_sq > MAX_SPEED * MAX_SPEED) {
        const float l = sqrtf(length_sq);
        vx = MAX_SPEED * (vx / l); vy = MAX_SPEED * (vy / l); vz = MAX_SPEED * (vz / l);
    }
    float x = get(0, 0) + vx * dt, y = get(0, 1) + vy * dt, z = get(0, 2) + vz * dt;
    if (x < -half_width) x = -half_width;
    if (x > half_width) x = half_width;
    if (z < -half_width) z = -half_width;
    if (z > half_width) z = half_width;
    set(1, 0, vx); set(1, 1, vy); set(1, 2, vz);
    set(0, 0, x); set(0, 1, y); set(0, 2, z);
}

static int world_spawn(World& w, uint64_t count, uint64_t comp_mask, const void* const* cols_in, uint64_t* first_out);

static void advance_flat(World& w, const AdvanceArgs& a) {
    const int64_t L = (int64_t)w.len;
    for (const SystemDesc& s : w.systems) {
        switch (s.kind) {
        case SYS_PARTICLES_UPDATE: {
            uint32_t ct = s.comp[0], cv = s.comp[1];
            float dt = f32_of(a.dt_bits);
            float g[3] = {s.fparam[0], s.fparam[1], s.fparam[2]};
            float* tx[3]; float* vv[3];
            for (int k = 0; k < 3; ++k) {
                tx[k] = (float*)w.cols[w.col_base[ct] + s.word[0] + k].data();
                vv[k] = (float*)w.cols[w.col_base[cv] + s.word[1] + k].data();
            }
#pragma omp parallel for schedule(static) if (L > 65536)
            for (int64_t ii = 0; ii < L; ++ii) {
                uint64_t i = (uint64_t)ii;
                if (!bit(w.alive, i) || !bit(w.present[ct], i) || !bit(w.present[cv], i)) continue;
                particles_update_one(dt, g,
                    [&](int which, int k) { return which ? vv[k][i] : tx[k][i]; },
                    [&](int which, int k, float v) { (which ? vv[k][i] : tx[k][i]) = v; });
            }
        } break;
        case SYS_TTL_DESPAWN: {
            uint32_t c = s.comp[0];
            uint64_t* ttl = (uint64_t*)w.cols[w.col_base[c] + s.word[0]].data();
            const int64_t NW = (L + 63) / 64;      // one 64-slot mask word per iteration: no write races
#pragma omp parallel for schedule(static) if (L > 65536)
            for (int64_t wi = 0; wi < NW; ++wi) {
                uint64_t m = w.alive[wi] & w.present[c][wi], kill = 0;
                while (m) {
                    int b = __builtin_ctzll(m); m &= m - 1;
                    uint64_t i = (uint64_t)wi * 64 + (uint64_t)b;
                    ttl[i] -= 1;                    // usize, wrapping (release semantics)
                    if (ttl[i] == 0) kill |= 1ULL << b;
                }
                w.alive[wi] &= ~kill;
            }
        } break;
        case SYS_ADD_U32: {
            uint32_t c = s.comp[0];
            uint32_t* p = (uint32_t*)w.cols[w.col_base[c] + s.word[0]].data();
            uint32_t d = (uint32_t)s.iparam[0];
            for (int64_t ii = 0; ii < L; ++ii) {
                uint64_t i = (uint64_t)ii;
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                p[i] += d;
            }
        } break;
        case SYS_SAT_SUB_DESPAWN: {
            uint32_t c = s.comp[0];
            uint32_t* p = (uint32_t*)w.cols[w.col_base[c] + s.word[0]].data();
            uint32_t d = (uint32_t)s.iparam[0];
            for (int64_t ii = 0; ii < L; ++ii) {
                uint64_t i = (uint64_t)ii;
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                p[i] = p[i] >= d ? p[i] - d : 0;    // saturating_sub
                if (p[i] == 0) { if (s.iparam[1] == 1) despawn_rollback_one(w, i); else setbit(w.alive, i, false); }
            }
        } break;
        case SYS_CUSTOM: run_custom_system(w, a, s); break;
        case SYS_BOX_MOVE: {
            uint32_t ct = s.comp[0], cv = s.comp[1], cp = s.comp[2];
            float dt = f32_of(a.dt_bits);
            float* tx[3]; float* vv[3];
            for (int k = 0; k < 3; ++k) {
                tx[k] = (float*)w.cols[w.col_base[ct] + s.word[0] + k].data();
                vv[k] = (float*)w.cols[w.col_base[cv] + s.word[1] + k].data();
            }
            const uint64_t* handle = (const uint64_t*)w.cols[w.col_base[cp] + s.word[2]].data();
            for (int64_t ii = 0; ii < L; ++ii) {
                uint64_t i = (uint64_t)ii;
                if (!bit(w.alive, i) || !bit(w.present[ct], i) || !bit(w.present[cv], i) || !bit(w.present[cp], i)) continue;
                if (handle[i] >= a.n_inputs) continue;          // inputs[p.handle] out of range: the reference panics
                box_move_one(dt, s.fparam, a.inputs[handle[i] * w.input_bytes],
                    [&](int which, int k) { return which ? vv[k][i] : tx[k][i]; },
                    [&](int which, int k, float v) { (which ? vv[k][i] : tx[k][i]) = v; });
            }
        } break;
        default: break;
        }
    }
    // Commands are deferred: spawns materialise after every system of the frame ran
    // (set.rs:118-134 ApplyDeferred after AdvanceWorldSystems::Main).
    for (const SystemDesc& s : w.systems) {
        if (s.kind != SYS_PARTICLES_SPAWN) continue;
        bool pressed = false;                       // spawn_pressed, particles.rs:254-256
        for (uint32_t k = 0; k < a.n_inputs; ++k) pressed |= (a.inputs[(size_t)k * w.input_bytes] & (uint8_t)s.iparam[1]) != 0;
        if (!pressed || a.spawn_count == 0) continue;
        uint32_t ct = s.comp[0], cv = s.comp[1], cl = s.comp[2];
        uint64_t first = 0;
        uint64_t mask = (1ULL << ct) | (1ULL << cv) | (1ULL << cl);
        if (world_spawn(w, a.spawn_count, mask, nullptr, &first) != 0) continue;
        float* vx = (float*)w.cols[w.col_base[cv] + 0].data();
        float* vy = (float*)w.cols[w.col_base[cv] + 1].data();
        float* vz = (float*)w.cols[w.col_base[cv] + 2].data();
        uint64_t* ttl = (uint64_t*)w.cols[w.col_base[cl] + 0].data();
        for (uint64_t j = 0; j < a.spawn_count; ++j) {
            vx[first + j] = a.spawn_vx[j]; vy[first + j] = a.spawn_vy[j]; vz[first + j] = 0.0f;
            ttl[first + j] = (uint64_t)s.iparam[0];
        }
    }
}

// REFSHAPED advance: same arithmetic over AoS rows (Bevy archetype-table iteration).
static void advance_ref(World& w, const AdvanceArgs& a) {
    for (const SystemDesc& s : w.systems) {
        switch (s.kind) {
        case SYS_PARTICLES_UPDATE: {
            uint32_t ct = s.comp[0], cv = s.comp[1];
            float dt = f32_of(a.dt_bits);
            float g[3] = {s.fparam[0], s.fparam[1], s.fparam[2]};
            uint32_t st_t = w.stride(ct), st_v = w.stride(cv);
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[ct], i) || !bit(w.present[cv], i)) continue;
                float* t = (float*)&w.aos[ct][i * st_t] + s.word[0];
                float* v = (float*)&w.aos[cv][i * st_v] + s.word[1];
                particles_update_one(dt, g,
                    [&](int which, int k) { return which ? v[k] : t[k]; },
                    [&](int which, int k, float x) { (which ? v[k] : t[k]) = x; });
            }
        } break;
        case SYS_TTL_DESPAWN: {
            uint32_t c = s.comp[0]; uint32_t st = w.stride(c);
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                uint64_t* ttl = (uint64_t*)&w.aos[c][i * st] + s.word[0];
                *ttl -= 1;
                if (*ttl == 0) setbit(w.alive, i, false);
            }
        } break;
        case SYS_ADD_U32: {
            uint32_t c = s.comp[0]; uint32_t st = w.stride(c);
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                uint32_t* p = (uint32_t*)&w.aos[c][i * st] + s.word[0];
                *p += (uint32_t)s.iparam[0];
            }
        } break;
        case SYS_SAT_SUB_DESPAWN: {
            uint32_t c = s.comp[0]; uint32_t st = w.stride(c); uint32_t d = (uint32_t)s.iparam[0];
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                uint32_t* p = (uint32_t*)&w.aos[c][i * st] + s.word[0];
                *p = *p >= d ? *p - d : 0;
                if (*p == 0) { if (s.iparam[1] == 1) despawn_rollback_one(w, i); else setbit(w.alive, i, false); }
            }
        } break;
        case SYS_CUSTOM: run_custom_system(w, a, s); break;
        case SYS_BOX_MOVE: {
            uint32_t ct = s.comp[0], cv = s.comp[1], cp = s.comp[2];
            float dt = f32_of(a.dt_bits);
            uint32_t st_t = w.stride(ct), st_v = w.stride(cv), st_p = w.stride(cp);
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[ct], i) || !bit(w.present[cv], i) || !bit(w.present[cp], i)) continue;
                float* t = (float*)&w.aos[ct][i * st_t] + s.word[0];
                float* v = (float*)&w.aos[cv][i * st_v] + s.word[1];
                const uint64_t handle = *((const uint64_t*)&w.aos[cp][i * st_p] + s.word[2]);
                if (handle >= a.n_inputs) continue;
                box_move_one(dt, s.fparam, a.inputs[handle * w.input_bytes],
                    [&](int which, int k) { return which ? v[k] : t[k]; },
                    [&](int which, int k, float x) { (which ? v[k] : t[k]) = x; });
            }
        } break;
        default: break;
        }
    }
    for (const SystemDesc& s : w.systems) {
        if (s.kind != SYS_PARTICLES_SPAWN) continue;
        bool pressed = false;
        for (uint32_t k = 0; k < a.n_inputs; ++k) pressed |= (a.inputs[(size_t)k * w.input_bytes] & (uint8_t)s.iparam[1]) != 0;
        if (!pressed || a.spawn_count == 0) continue;
        uint32_t ct = s.comp[0], cv = s.comp[1], cl = s.comp[2];
        uint64_t first = 0;
        uint64_t mask = (1ULL << ct) | (1ULL << cv) | (1ULL << cl);
        if (world_spawn(w, a.spawn_count, mask, nullptr, &first) != 0) continue;
        uint32_t st_v = w.stride(cv), st_l = w.stride(cl);
        for (uint64_t j = 0; j < a.spawn_count; ++j) {
            float* v = (float*)&w.aos[cv][(first + j) * st_v];
            v[0] = a.spawn_vx[j]; v[1] = a.spawn_vy[j]; v[2] = 0.0f;
            uint64_t* ttl = (uint64_t*)&w.aos[cl][(first + j) * st_l];
            *ttl = (uint64_t)s.iparam[0];
        }
    }
}

// Rollback on_add hook + RollbackOrdered::push (rollback.rs:45-59,69-74): slot == order index.
static int world_spawn(World& w, uint64_t count, uint64_t comp_mask, const void* const* cols_in, uint64_t* first_out) {
    w.seal();
    if (w.len + count > w.capacity) { w.err = "capacity exceeded"; return -3; }
    uint64_t first = w.len;
    uint32_t ci = 0;
    for (uint32_t c = 0; c < w.comps.size(); ++c) {
        const Comp& cc = w.comps[c];
        bool has = (comp_mask >> c) & 1ULL;
        for (uint32_t k = 0; k < cc.n_words; ++k) {
            uint8_t* dst = &w.cols[w.col_base[c] + k][first * cc.word_bytes];
            const void* src = (has && cols_in) ? cols_in[ci] : nullptr;
            if (has) ++ci;
            if (src) memcpy(dst, src, (size_t)count * cc.word_bytes);
            else for (uint64_t j = 0; j < count; ++j) memcpy(dst + j * cc.word_bytes, &cc.defaults[k * cc.word_bytes], cc.word_bytes);
        }
        for (uint64_t j = 0; j < count; ++j) setbit(w.present[c], first + j, has);
    }
    for (uint64_t j = 0; j < count; ++j) { setbit(w.alive, first + j, true); setbit(w.disabled, first + j, false); }
    w.len += count;
    if (w.mode == 1) {
        ref_sync_from_flat(w, first, count);
        for (uint64_t j = 0; j < count; ++j) {
            uint64_t id = World::rid(first + j), idx = w.sorted.size();
            w.sorted.push_back(id);
            w.order_map.insert(id, &idx);
        }
    }
    if (first_out) *first_out = first;
    return 0;
}

// REFSHAPED only: the reference's per-component systems (ComponentChecksumPlugin<C>::update, ComponentSnapshotPlugin<S>::
// save / load) are separate Bevy systems that the multi-threaded executor MAY run on different worker threads
// ("also report a 3-thread (one per component) figure").  > 1: one OpenMP thread per registered component.
static int g_ref_comp_threads = 1;

// ---------------- SaveWorld ----------------
static void world_save(World& w, uint64_t out[2]) {
    w.seal();
    // SaveWorldSystems::Checksum (set.rs:104-107): component parts, entity part; then fold (checksum.rs:88-99)
    uint64_t total = 0;
    if (w.mode == 1 && g_ref_comp_threads > 1) {
        const int nc = (int)w.comps.size();
#pragma omp parallel for num_threads(g_ref_comp_threads) schedule(static, 1) reduction(^ : total)
        for (int c = 0; c < nc; ++c) if (w.comps[c].checksummed) total ^= component_checksum_ref(w, (uint32_t)c);
    } else {
        for (uint32_t c = 0; c < w.comps.size(); ++c)
            if (w.comps[c].checksummed) total ^= (w.mode == 1 ? component_checksum_ref(w, c) : component_checksum_flat(w, c));
    }
    total ^= entity_checksum(active_count_flat(w), w.len);
    out[0] = total; out[1] = 0;   // `as u128` of a u64: upper half always 0 (component_checksum.rs:95)

    // SaveWorldSystems::Snapshot: sync_depth -> discard_old_snapshots -> save (component_snapshot.rs:137-143)
    if (w.mode == 0) {
        if (w.has_confirmed) w.ring.confirm(w.confirmed);
        FlatSnap s;
        s.len = w.len;
        s.cols.resize(w.cols.size());
        s.stored.resize(w.comps.size());
        for (uint32_t c = 0; c < w.comps.size(); ++c) {
            if (w.comps[c].no_rollback) continue;
            if (w.comps[c].store_fn) {
                // ComponentSnapshotPlugin<S>::save with a user-written Strategy (component_snapshot.rs:66-84): Stored = S::store(component)
                const Comp& cc = w.comps[c];
                s.stored[c].assign((size_t)w.len * cc.s_n_words, 0);
                for (uint64_t i = 0; i < w.len; ++i) {
                    if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                    uint64_t tg[MAX_WORDS], st[MAX_WORDS] = {0};
                    for (uint32_t k = 0; k < cc.n_words; ++k) tg[k] = load_word(w, c, k, i);
                    cc.store_fn(tg, st, cc.strat_user);
                    for (uint32_t k = 0; k < cc.s_n_words; ++k) s.stored[c][i * cc.s_n_words + k] = st[k] & word_mask(cc.s_word_bytes);
                }
                continue;
            }
            for (uint32_t k = 0; k < w.comps[c].n_words; ++k) {
                auto& src = w.cols[w.col_base[c] + k];
                s.cols[w.col_base[c] + k].assign(src.begin(), src.begin() + (size_t)w.len * w.comps[c].word_bytes);
            }
        }
        size_t nw = (w.len + 63) / 64;
        s.alive.assign(w.alive.begin(), w.alive.begin() + nw);
        s.present.resize(w.comps.size());
        for (uint32_t c = 0; c < w.comps.size(); ++c)
            if (!w.comps[c].no_rollback) s.present[c].assign(w.present[c].begin(), w.present[c].begin() + nw);
        w.ring.push(w.frame, std::move(s));
    } else {
        if (w.has_confirmed) w.rring.confirm(w.confirmed);
        RefSnap s;
        s.len = w.len;
        uint64_t n_alive = active_count_flat(w);
        s.comp.resize(w.comps.size());
        const int nc_save = (int)w.comps.size();
#pragma omp parallel for num_threads(g_ref_comp_threads) schedule(static, 1) if (g_ref_comp_threads > 1)
        for (int c = 0; c < nc_save; ++c) {                       // ComponentSnapshotPlugin::save, component_snapshot.rs:66-84
            if (w.comps[c].no_rollback) continue;
            uint32_t st = w.stride(c);
            s.comp[c].init(n_alive, st);
            for (uint64_t i = 0; i < w.len; ++i) {
                if (!bit(w.alive, i) || !bit(w.present[c], i)) continue;
                s.comp[c].insert(World::rid(i), &w.aos[c][i * st]);
            }
        }
        s.entities.init(n_alive, 8);                              // EntitySnapshotPlugin::save, entity.rs:39-51
        for (uint64_t i = 0; i < w.len; ++i) {
            if (!bit(w.alive, i)) continue;
            uint64_t ent = i; s.entities.insert(World::rid(i), &ent);
        }
        s.order_clone = w.order_map;                              // ResourceSnapshotPlugin<CloneStrategy<RollbackOrdered>>, mod.rs:342
        s.sorted_clone = w.sorted;
        w.rring.push(w.frame, std::move(s));
    }
}

// ---------------- LoadWorld ----------------
static int world_load(World& w, int32_t frame) {
    w.seal();
    w.frame = frame;                                              // schedule_systems.rs:244-247
    if (w.mode == 0) {
        if (!w.ring.rollback(frame)) { w.err = "Could not rollback: no snapshot at that frame"; return -2; }
        FlatSnap& s = *w.ring.get();
        resurrect_and_reconcile(w, s.len, [&](uint64_t i) { return bit(s.alive, i); });
        // entity.rs:55-99 + component_snapshot.rs:95-123 collapse to: masks and columns := snapshot
        for (size_t k = 0; k < s.cols.size(); ++k) if (!s.cols[k].empty()) memcpy(w.cols[k].data(), s.cols[k].data(), s.cols[k].size());
        for (uint32_t c = 0; c < w.comps.size(); ++c) {
            // ComponentSnapshotPlugin<S>::load with a user-written Strategy (component_snapshot.rs:95-123): component = S::load(Stored)
            const Comp& cc = w.comps[c];
            if (!cc.load_fn || cc.no_rollback || s.stored.size() <= c) continue;
            for (uint64_t i = 0; i < s.len; ++i) {
                if (!bit(s.alive, i) || !bit(s.present[c], i)) continue;
                uint64_t tg[MAX_WORDS] = {0}, st[MAX_WORDS];
                for (uint32_t k = 0; k < cc.s_n_words; ++k) st[k] = s.stored[c][i * cc.s_n_words + k];
                cc.load_fn(st, tg, cc.strat_user);
                for (uint32_t k = 0; k < cc.n_words; ++k) store_word(w, c, k, i, tg[k]);
            }
        }
        std::fill(w.alive.begin(), w.alive.end(), 0);
        memcpy(w.alive.data(), s.alive.data(), s.alive.size() * 8);
        for (uint32_t c = 0; c < w.comps.size(); ++c) {
            if (w.comps[c].no_rollback) continue;
            std::fill(w.present[c].begin(), w.present[c].end(), 0);
            memcpy(w.present[c].data(), s.present[c].data(), s.present[c].size() * 8);
        }
        w.len = s.len;                                            // RollbackOrdered restored (mod.rs:342)
    } else {
        if (!w.rring.rollback(frame)) { w.err = "Could not rollback: no snapshot at that frame"; return -2; }
        RefSnap& s = *w.rring.get();
        resurrect_and_reconcile(w, s.len, [&](uint64_t i) { return s.entities.get(World::rid(i)) != nullptr; });
        // EntitySnapshotPlugin::load (entity.rs:55-99): rollback_mapping built from snapshot + live query,
        // then entity_map collected twice (HashMap then EntityHashMap).
        uint64_t hi = s.len > w.len ? s.len : w.len;
        RefTable mapping; mapping.init(s.entities.count + active_count_flat(w), 16);
        for (uint64_t i = 0; i < s.len; ++i) {
            const uint8_t* e = s.entities.get(World::rid(i));
            if (!e) continue;
            uint64_t pair[2] = {0, i + 1}; mapping.insert(World::rid(i), pair);
        }
        for (uint64_t i = 0; i < w.len; ++i) {
            if (!bit(w.alive, i)) continue;
            const uint8_t* e = mapping.get(World::rid(i));
            uint64_t pair[2] = {i + 1, 0};
            if (e) memcpy(&pair[1], e + 8, 8);
            mapping.insert(World::rid(i), pair);
        }
        RefTable entity_map; entity_map.init(mapping.count, 8);
        for (uint64_t i = 0; i < hi; ++i) {
            const uint8_t* e = mapping.get(World::rid(i));
            if (!e) continue;
            uint64_t pair[2]; memcpy(pair, e, 16);
            if (pair[0] && pair[1]) { uint64_t v = pair[1] - 1; entity_map.insert(i, &v); }
            else if (pair[0]) setbit(w.alive, i, false);                    // despawn
            else { setbit(w.alive, i, true); uint64_t v = i; entity_map.insert(i, &v); }   // respawn with same RollbackId
        }
        RefTable entity_hash_map; entity_hash_map.init(entity_map.count, 8);   // .collect::<EntityHashMap<Entity>>()
        for (uint64_t i = 0; i < hi; ++i) { const uint8_t* e = entity_map.get(i); if (e) entity_hash_map.insert(i, e); }
        // slots beyond the snapshot's len never existed then
        for (uint64_t i = s.len; i < w.len; ++i) setbit(w.alive, i, false);
        // ComponentSnapshotPlugin::load (component_snapshot.rs:95-123): per entity lookup
        const int nc_load = (int)w.comps.size();
#pragma omp parallel for num_threads(g_ref_comp_threads) schedule(static, 1) if (g_ref_comp_threads > 1)
        for (int c = 0; c < nc_load; ++c) {
            if (w.comps[c].no_rollback) continue;
            uint32_t st = w.stride(c);
            for (uint64_t i = 0; i < s.len; ++i) {
                if (!bit(w.alive, i)) { setbit(w.present[c], i, false); continue; }
                const uint8_t* v = s.comp[c].get(World::rid(i));
                if (v) { memcpy(&w.aos[c][i * st], v, st); setbit(w.present[c], i, true); }   // update / insert
                else setbit(w.present[c], i, false);                                          // remove
            }
            for (uint64_t i = s.len; i < w.len; ++i) setbit(w.present[c], i, false);
        }
        w.order_map = s.order_clone;                              // RollbackOrdered resource restore
        w.sorted = s.sorted_clone;
        w.len = s.len;
    }
    return 0;
}

// ---------------- AdvanceWorld ----------------
static void world_advance(World& w, const AdvanceArgs& a_in) {
    w.seal();
    w.frame += 1;                                                 // schedule_systems.rs:254-259
    AdvanceArgs a = a_in;
    if (a.dt_bits == 0) a.dt_bits = dt_bits_for_frame(w.fps, w.frame);   // GgrsTimePlugin::update, time.rs:63-87
    despawn_confirmed(w);                                         // AdvanceWorldSystems::DespawnConfirmed, set.rs:68-70
    w.spawn_reqs.clear();
    if (w.mode == 0) advance_flat(w, a); else advance_ref(w, a);
    // a user-written spawn system: its Commands are applied with the others, after every system of the frame ran (set.rs:118-134); the
    // entities take RollbackOrdered's next indices (rollback.rs:69-74), the bundle's components their registered defaults, then what the spawner writes
    for (const SystemDesc& s : w.systems) {
        if (s.kind != SYS_SPAWN_CUSTOM) continue;
        static const uint8_t zero_status[16] = {0};
        if (w.spawn_customs[s.comp[0]].payload_stride == 0xFFFFFFFFu) {
            // spawns the systems decided (GGRS_SPAWN_PAYLOAD_PARENT): Commands are applied in the order they were queued -- entities are visited in RollbackOrdered
            // (slot) order, so parents in slot order, each parent's children k = 0 .. n-1; the child's payload is its parent's record
            const SpawnSys& sp = w.spawn_customs[s.comp[0]];
            const FrameView f = frame_view(w, a, s, zero_status);
            const std::map<uint64_t, World::SpawnReq> reqs = w.spawn_reqs;
            w.spawn_reqs.clear();
            uint64_t total = 0;
            for (auto& kv : reqs) total += kv.second.n;
            if (w.len + total > w.capacity) continue;                  // (the device drops the frame's spawns and reports: tests stay within capacity)
            for (auto& kv : reqs) {
                uint64_t first = 0;
                if (world_spawn(w, kv.second.n, sp.bundle_mask, nullptr, &first) != 0) break;
                if (w.mode == 1) ref_sync_to_flat(w);
                for (uint64_t k = 0; k < kv.second.n; ++k) {
                    uint64_t words[8];
                    for (uint32_t b = 0; b < sp.n_bind; ++b) words[b] = load_word(w, sp.comp[b], sp.word[b], first + k);
                    sp.fn(words, first + k, k, &f, (const uint8_t*)kv.second.words, sp.user);
                    for (uint32_t b = 0; b < sp.n_bind; ++b) store_word(w, sp.comp[b], sp.word[b], first + k, words[b]);
                }
                if (w.mode == 1) ref_sync_from_flat(w, first, kv.second.n);
            }
            continue;
        }
        if (a.spawn_count == 0) continue;
        const SpawnSys& sp = w.spawn_customs[s.comp[0]];
        const FrameView f = frame_view(w, a, s, zero_status);
        uint64_t first = 0;
        if (world_spawn(w, a.spawn_count, sp.bundle_mask, nullptr, &first) != 0) continue;
        if (w.mode == 1) ref_sync_to_flat(w);
        for (uint64_t k = 0; k < a.spawn_count; ++k) {
            uint64_t words[8];
            for (uint32_t b = 0; b < sp.n_bind; ++b) words[b] = load_word(w, sp.comp[b], sp.word[b], first + k);
            sp.fn(words, first + k, k, &f, a.spawn_payload ? a.spawn_payload + (size_t)sp.payload_stride * k : nullptr, sp.user);
            for (uint32_t b = 0; b < sp.n_bind; ++b) store_word(w, sp.comp[b], sp.word[b], first + k, words[b]);
        }
        if (w.mode == 1) ref_sync_from_flat(w, first, a.spawn_count);
    }
}

// ===========================================================================
// C ABI (ctypes)
// ===========================================================================
extern "C" {

uint64_t gor_seahash_buffer(const uint8_t* p, uint64_t n) { return sea::hash_buffer(p, n); }
uint64_t gor_seahash_stream(const uint8_t* p, uint64_t n, const uint32_t* chunk_sizes, uint32_t n_chunks) {
    // writes `p` in the given chunking (sum must equal n); chunk_sizes==NULL -> one write
    sea::Hasher h;
    if (!chunk_sizes) { h.write(p, n); return h.finish(); }
    uint64_t off = 0;
    for (uint32_t k = 0; k < n_chunks; ++k) { h.write(p + off, chunk_sizes[k]); off += chunk_sizes[k]; }
    return h.finish();
}
uint64_t gor_diffuse(uint64_t x) { return sea::diffuse(x); }
uint64_t gor_inner_hash_units(const uint32_t* u, uint32_t n) { return inner_hash_units(u, n); }
uint64_t gor_entity_part(uint64_t order, uint64_t inner) { return entity_part(order, inner); }
uint64_t gor_finalize_part(uint64_t x) { return finalize_part(x); }
uint64_t gor_entity_checksum(uint64_t active, uint64_t total) { return entity_checksum(active, total); }
uint32_t gor_dt_bits(uint64_t fps, int32_t frame) { return dt_bits_for_frame(fps, frame); }

// ---- ring KAT surface: GgrsSnapshots<u32,u32> as in mod.rs:357 ----
void* gor_ring_create(uint64_t depth) { auto* r = new Ring<uint32_t>(); r->set_depth(depth); return r; }
void gor_ring_destroy(void* r) { delete (Ring<uint32_t>*)r; }
void gor_ring_set_depth(void* r, uint64_t d) { ((Ring<uint32_t>*)r)->set_depth(d); }
void gor_ring_push(void* r, int32_t f, uint32_t v) { ((Ring<uint32_t>*)r)->push(f, v); }
void gor_ring_confirm(void* r, int32_t f) { ((Ring<uint32_t>*)r)->confirm(f); }
int gor_ring_rollback(void* r, int32_t f) { return ((Ring<uint32_t>*)r)->rollback(f) ? 0 : -2; }
int gor_ring_get(void* r, uint32_t* out) { auto* p = ((Ring<uint32_t>*)r)->get(); if (!p) return -2; *out = *p; return 0; }
int gor_ring_peek(void* r, int32_t f, uint32_t* out) { auto* p = ((Ring<uint32_t>*)r)->peek(f); if (!p) return -2; *out = *p; return 0; }
uint64_t gor_ring_len(void* r) { return ((Ring<uint32_t>*)r)->frames.size(); }

// ---- world ----
void* gor_world_create(uint64_t capacity, uint32_t depth, int mode) {
    auto* w = new World();
    w->capacity = capacity; w->mode = mode;
    w->ring.set_depth(depth); w->rring.set_depth(depth);
    return w;
}
void gor_world_destroy(void* w) { delete (World*)w; }
const char* gor_last_error(void* w) { return ((World*)w)->err.c_str(); }

int gor_register_component(void* wp, const char* name, uint32_t word_bytes, uint32_t n_words, uint32_t* id) {
    World& w = *(World*)wp;
    if (w.sealed || w.comps.size() >= MAX_COMPS || n_words == 0 || n_words > MAX_WORDS || (word_bytes != 1 && word_bytes != 2 && word_bytes != 4 && word_bytes != 8)) return -1;
    Comp c; c.name = name; c.word_bytes = word_bytes; c.n_words = n_words;
    c.defaults.assign((size_t)word_bytes * n_words, 0);
    w.comps.push_back(c);
    if (id) *id = (uint32_t)w.comps.size() - 1;
    return 0;
}
int gor_register_component_ex(void* wp, const char* name, uint32_t word_bytes, uint32_t n_words, uint32_t flags, uint32_t* id) {
    uint32_t c = 0;
    int rc = gor_register_component(wp, name, word_bytes, n_words, &c);
    if (rc) return rc;
    ((World*)wp)->comps[c].no_rollback = (flags & 1u) != 0;
    if (id) *id = c;
    return 0;
}
int gor_set_component_default(void* wp, uint32_t c, const void* words) {
    World& w = *(World*)wp;
    if (c >= w.comps.size()) return -1;
    memcpy(w.comps[c].defaults.data(), words, w.comps[c].defaults.size());
    return 0;
}
// checksum_component / checksum_component_with_hash (rollback_app.rs:99-101,119-121)
int gor_checksum_component(void* wp, uint32_t c, const uint32_t* word_idx, uint32_t n) {
    World& w = *(World*)wp;
    if (c >= w.comps.size() || w.sealed) return -1;                  // registration ends with the first spawn, as in the library (and in an App: plugins are added at build time)
    Comp& cc = w.comps[c];
    cc.cks_units.clear(); cc.cks_fn = nullptr;
    for (uint32_t k = 0; k < n; ++k) {
        if (word_idx[k] >= cc.n_words) return -1;
        cc.cks_units.push_back(word_idx[k]);
    }
    if (cc.cks_units.size() > MAX_UNITS) return -1;
    cc.checksummed = true;
    return 0;
}
// checksum_component::<T>(hasher) with an arbitrary fn(&T) -> u64 (rollback_app.rs:119-121); `words` = the component's
// n_words * word_bytes bytes of one entity
int gor_checksum_component_custom(void* wp, uint32_t c, uint64_t (*fn)(const uint8_t* words, uint64_t slot, void* user), void* user) {
    World& w = *(World*)wp;
    if (c >= w.comps.size() || !fn) return -1;
    Comp& cc = w.comps[c];
    cc.cks_units.clear(); cc.cks_fn = fn; cc.cks_user = user;
    cc.checksummed = true;
    return 0;
}
int gor_add_system(void* wp, const SystemDesc* d) {
    World& w = *(World*)wp;
    if (w.systems.size() >= MAX_SYSTEMS) return -1;
    w.systems.push_back(*d);
    return 0;
}
int gor_spawn(void* wp, uint64_t count, uint64_t comp_mask, const void* const* cols, uint64_t* first) {
    return world_spawn(*(World*)wp, count, comp_mask, cols, first);
}
int gor_despawn(void* wp, uint64_t slot) {
    World& w = *(World*)wp; w.seal();
    if (slot >= w.len) return -1;
    setbit(w.alive, slot, false);
    return 0;
}
int gor_despawn_rollback(void* wp, uint64_t slot) {
    World& w = *(World*)wp; w.seal();
    if (slot >= w.len) return -1;
    despawn_rollback_one(w, slot);
    return 0;
}
int gor_download_disabled(void* wp, uint64_t* dst, uint64_t n_words64) {
    World& w = *(World*)wp; w.seal();
    for (uint64_t k = 0; k < n_words64; ++k) dst[k] = k < w.disabled.size() ? w.disabled[k] : 0;
    return 0;
}
int gor_download_despawned_frames(void* wp, uint64_t first, uint64_t count, int32_t* frames) {
    World& w = *(World*)wp; w.seal();
    if (first + count > w.capacity) return -1;
    memcpy(frames, w.dframe.data() + first, (size_t)count * 4);
    return 0;
}
int gor_insert_component(void* wp, uint32_t c, uint64_t slot, const void* words) {
    World& w = *(World*)wp; w.seal();
    if (c >= w.comps.size() || slot >= w.len) return -1;
    const Comp& cc = w.comps[c];
    for (uint32_t k = 0; k < cc.n_words; ++k)
        memcpy(&w.cols[w.col_base[c] + k][slot * cc.word_bytes], (const uint8_t*)words + k * cc.word_bytes, cc.word_bytes);
    setbit(w.present[c], slot, true);
    if (w.mode == 1) memcpy(&w.aos[c][slot * w.stride(c)], words, w.stride(c));
    return 0;
}
int gor_remove_component(void* wp, uint32_t c, uint64_t slot) {
    World& w = *(World*)wp; w.seal();
    if (c >= w.comps.size() || slot >= w.len) return -1;
    setbit(w.present[c], slot, false);
    return 0;
}
int gor_upload_word(void* wp, uint32_t c, uint32_t word, uint64_t first, uint64_t count, const void* src) {
    World& w = *(World*)wp; w.seal();
    if (c >= w.comps.size() || word >= w.comps[c].n_words || first + count > w.capacity) return -1;
    uint32_t wb = w.comps[c].word_bytes;
    memcpy(&w.cols[w.col_base[c] + word][first * wb], src, (size_t)count * wb);
    if (w.mode == 1) {
        uint32_t st = w.stride(c);
        for (uint64_t i = 0; i < count; ++i)
            memcpy(&w.aos[c][(first + i) * st + word * wb], (const uint8_t*)src + i * wb, wb);
    }
    return 0;
}
int gor_download_word(void* wp, uint32_t c, uint32_t word, uint64_t first, uint64_t count, void* dst) {
    World& w = *(World*)wp; w.seal();
    if (c >= w.comps.size() || word >= w.comps[c].n_words || first + count > w.capacity) return -1;
    if (w.mode == 1) ref_sync_to_flat(w);
    uint32_t wb = w.comps[c].word_bytes;
    memcpy(dst, &w.cols[w.col_base[c] + word][first * wb], (size_t)count * wb);
    return 0;
}
int gor_download_alive(void* wp, uint64_t* dst, uint64_t n_words64) {
    World& w = *(World*)wp; w.seal();
    for (uint64_t k = 0; k < n_words64; ++k) dst[k] = k < w.alive.size() ? w.alive[k] : 0;
    return 0;
}
int gor_download_present(void* wp, uint32_t c, uint64_t* dst, uint64_t n_words64) {
    World& w = *(World*)wp; w.seal();
    if (c >= w.comps.size()) return -1;
    for (uint64_t k = 0; k < n_words64; ++k) dst[k] = k < w.present[c].size() ? w.present[c][k] : 0;
    // a non-rollback component dies with its entity (it exists while the entity is alive or disabled)
    if (w.comps[c].no_rollback)
        for (uint64_t k = 0; k < n_words64 && k < w.alive.size(); ++k) dst[k] &= (w.alive[k] | w.disabled[k]);
    return 0;
}
uint64_t gor_len(void* wp) { return ((World*)wp)->len; }
uint64_t gor_active_count(void* wp) { World& w = *(World*)wp; w.seal(); return active_count_flat(w); }
int32_t gor_frame(void* wp) { return ((World*)wp)->frame; }
void gor_set_frame(void* wp, int32_t f) { ((World*)wp)->frame = f; }
void gor_set_frame_rate(void* wp, uint64_t fps) { ((World*)wp)->fps = fps; }
void gor_set_depth(void* wp, uint32_t d) { World& w = *(World*)wp; w.ring.set_depth(d); w.rring.set_depth(d); }
void gor_set_confirmed(void* wp, int has, int32_t f) { World& w = *(World*)wp; w.has_confirmed = has != 0; w.confirmed = f; }
int gor_has_snapshot(void* wp, int32_t f) { World& w = *(World*)wp; return w.mode == 0 ? (w.ring.peek(f) != nullptr) : (w.rring.peek(f) != nullptr); }
uint64_t gor_snapshot_count(void* wp) { World& w = *(World*)wp; return w.mode == 0 ? w.ring.frames.size() : w.rring.frames.size(); }

int gor_save(void* wp, uint64_t out[2]) { world_save(*(World*)wp, out); return 0; }
int gor_load(void* wp, int32_t frame) { return world_load(*(World*)wp, frame); }
int gor_advance(void* wp, uint32_t dt_bits, const uint8_t* inputs, uint32_t n_inputs,
                uint64_t spawn_count, const float* vx, const float* vy) {
    AdvanceArgs a{dt_bits, inputs, n_inputs, spawn_count, vx, vy};
    world_advance(*(World*)wp, a);
    return 0;
}
int gor_set_input_layout(void* wp, uint32_t input_bytes, uint32_t max_players) {
    World& w = *(World*)wp;
    if (input_bytes == 0 || input_bytes > 16 || max_players == 0 || max_players > 16) { w.err = "bad input layout"; return -1; }
    w.input_bytes = input_bytes; w.max_players = max_players;
    return 0;
}
int gor_add_custom_system(void* wp, CustomSysFn fn, void* user, uint32_t n_bind, const uint32_t* comp, const uint32_t* word, const int64_t* iparam, const float* fparam) {
    World& w = *(World*)wp;
    if (w.sealed || !fn || n_bind == 0 || n_bind > 8) { w.err = "bad custom system"; return -1; }
    CustomSys cs; cs.fn = fn; cs.user = user; cs.n_bind = n_bind;
    for (uint32_t b = 0; b < n_bind; ++b) { if (comp[b] >= w.comps.size() || word[b] >= w.comps[comp[b]].n_words) { w.err = "bad binding"; return -1; } cs.comp[b] = comp[b]; cs.word[b] = word[b]; }
    SystemDesc d; memset(&d, 0, sizeof d);
    d.kind = SYS_CUSTOM; d.comp[0] = (uint32_t)w.customs.size();
    if (iparam) { d.iparam[0] = iparam[0]; d.iparam[1] = iparam[1]; }
    if (fparam) for (int k = 0; k < 4; ++k) d.fparam[k] = fparam[k];
    w.customs.push_back(cs); w.systems.push_back(d);
    return 0;
}
int gor_add_spawn_system(void* wp, SpawnSysFn fn, void* user, uint64_t bundle_mask, uint32_t payload_stride, uint32_t n_bind, const uint32_t* comp, const uint32_t* word,
                         const int64_t* iparam, const float* fparam) {
    World& w = *(World*)wp;
    if (w.sealed || !fn || n_bind > 8 || bundle_mask == 0) { w.err = "bad spawn system"; return -1; }
    SpawnSys sp; sp.fn = fn; sp.user = user; sp.bundle_mask = bundle_mask; sp.payload_stride = payload_stride; sp.n_bind = n_bind;
    for (uint32_t b = 0; b < n_bind; ++b) { if (comp[b] >= w.comps.size() || word[b] >= w.comps[comp[b]].n_words) { w.err = "bad binding"; return -1; } sp.comp[b] = comp[b]; sp.word[b] = word[b]; }
    SystemDesc d; memset(&d, 0, sizeof d);
    d.kind = SYS_SPAWN_CUSTOM; d.comp[0] = (uint32_t)w.spawn_customs.size();
    if (iparam) { d.iparam[0] = iparam[0]; d.iparam[1] = iparam[1]; }
    if (fparam) for (int k = 0; k < 4; ++k) d.fparam[k] = fparam[k];
    w.spawn_customs.push_back(sp); w.systems.push_back(d);
    return 0;
}
int gor_register_component_strategy(void* wp, uint32_t c, uint32_t stored_word_bytes, uint32_t stored_n_words, StoreFn store, LoadFn load, void* user) {
    World& w = *(World*)wp;
    if (w.sealed || c >= w.comps.size() || !store || !load || stored_n_words == 0 || stored_n_words > MAX_WORDS) { w.err = "bad strategy"; return -1; }
    if (w.mode != 0) { w.err = "strategies: FLAT oracle only"; return -1; }
    Comp& cc = w.comps[c];
    cc.s_word_bytes = stored_word_bytes; cc.s_n_words = stored_n_words; cc.store_fn = store; cc.load_fn = load; cc.strat_user = user;
    return 0;
}

// Batched request list (handle_requests, schedule_systems.rs:170-289).  Layout must match
// include/ggrs_hip.h ggrs_request.
struct Request {
    uint32_t kind;        // 1 save, 2 load, 3 advance
    int32_t frame;
    uint32_t dt_bits;
    uint32_t n_inputs;
    const uint8_t* inputs;
    const uint8_t* status;
    uint64_t spawn_count;
    const float* spawn_vx;
    const float* spawn_vy;
    const void* spawn_payload;
    uint64_t spawn_payload_bytes;
};
int gor_handle_requests(void* wp, const Request* reqs, uint32_t n, uint64_t* checksums_out) {
    World& w = *(World*)wp;
    uint32_t ns = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const Request& r = reqs[i];
        if (r.kind == 1) { world_save(w, checksums_out + 2 * ns); ++ns; }
        else if (r.kind == 2) { int rc = world_load(w, r.frame); if (rc) return rc; }
        else if (r.kind == 3) {
            AdvanceArgs a{r.dt_bits, r.inputs, r.n_inputs, r.spawn_count, r.spawn_vx, r.spawn_vy};
            a.status = r.status; a.spawn_payload = (const uint8_t*)r.spawn_payload; a.spawn_payload_bytes = r.spawn_payload_bytes;
            world_advance(w, a);
        }
        else return -1;
    }
    return 0;
}

// Timed SyncTest loop for bench.py's cpu_baseline: `ticks` steady-state ticks of
// [Load(F-d), Adv, (Save,Adv)x(d-1), Save(F), Adv]; returns seconds.
double gor_bench_synctest(void* wp, uint32_t d, uint32_t warm_ticks, uint32_t ticks) {
    World& w = *(World*)wp;
    uint64_t cs[2];
    auto tick = [&]() {
        int32_t F = w.frame;
        if (F > (int32_t)d) {
            w.has_confirmed = (F - (int32_t)d) >= 0; w.confirmed = F - (int32_t)d;
            world_load(w, F - (int32_t)d);
            for (uint32_t i = 0; i < d; ++i) {
                if (i > 0) { w.has_confirmed = (w.frame - (int32_t)d) >= 0; w.confirmed = w.frame - (int32_t)d; world_save(w, cs); }
                AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a);
            }
        }
        w.has_confirmed = (w.frame - (int32_t)d) >= 0; w.confirmed = w.frame - (int32_t)d;
        world_save(w, cs);
        AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a);
    };
    for (uint32_t t = 0; t < warm_ticks; ++t) tick();
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t t = 0; t < ticks; ++t) tick();
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// The same SyncTest ticks as gor_bench_synctest, but every SaveWorld's Checksum(u128) is handed back in request order
// ({lo, hi} per Save): bench.py replays the frames its timed GPU ticks covered and compares (SURVEY 8d "parity check in
// the same run").  Returns the seconds the LAST `timed_ticks` of the `ticks` took; *n_saves_out = checksums written.
double gor_replay_synctest(void* wp, uint32_t d, uint32_t ticks, uint32_t timed_ticks, uint64_t* cs_out, uint64_t cs_cap, uint64_t* n_saves_out) {
    World& w = *(World*)wp;
    uint64_t n = 0;
    auto save = [&]() {
        uint64_t cs[2];
        w.has_confirmed = (w.frame - (int32_t)d) >= 0; w.confirmed = w.frame - (int32_t)d;
        world_save(w, cs);
        if (cs_out && n < cs_cap) { cs_out[2 * n] = cs[0]; cs_out[2 * n + 1] = cs[1]; }
        ++n;
    };
    auto tick = [&]() {
        int32_t F = w.frame;
        if (F > (int32_t)d) {
            w.has_confirmed = (F - (int32_t)d) >= 0; w.confirmed = F - (int32_t)d;
            world_load(w, F - (int32_t)d);
            for (uint32_t i = 0; i < d; ++i) {
                if (i > 0) save();
                AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a);
            }
        }
        save();
        AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a);
    };
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t t = 0; t < ticks; ++t) {
        if (t + timed_ticks == ticks) t0 = std::chrono::steady_clock::now();
        tick();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (n_saves_out) *n_saves_out = n;
    return std::chrono::duration<double>(t1 - t0).count();
}
// bench.py's pre-heat runs hundreds of SyncTest ticks before the timed region; by determinism (what SyncTest itself asserts,
// tests/synctest.rs) the state they leave at frame F is the state F plain AdvanceWorlds leave, so the checker gets there with
// `skip_frames` advance-only frames, then d + 1 plain [SaveWorld, AdvanceWorld] ticks (the ring the first timed tick loads
// from), then `ticks` real SyncTest ticks whose checksums are handed back.  Returns the seconds the real ticks took.
double gor_replay_synctest_from(void* wp, uint32_t d, uint32_t skip_frames, uint32_t ticks, uint64_t* cs_out, uint64_t cs_cap, uint64_t* n_saves_out) {
    World& w = *(World*)wp;
    for (uint32_t f = 0; f < skip_frames; ++f) { AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a); }
    uint64_t cs[2];
    for (uint32_t t = 0; t < d + 1; ++t) {
        w.has_confirmed = (w.frame - (int32_t)d) >= 0; w.confirmed = w.frame - (int32_t)d;
        world_save(w, cs);
        AdvanceArgs a{0, nullptr, 0, 0, nullptr, nullptr}; world_advance(w, a);
    }
    return gor_replay_synctest(wp, d, ticks, ticks, cs_out, cs_cap, n_saves_out);
}
void gor_set_ref_component_threads(int n) { g_ref_comp_threads = n < 1 ? 1 : n; }

int gor_num_threads() {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void gor_set_num_threads(int n) {
#if defined(_OPENMP)
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

}  // extern "C"
