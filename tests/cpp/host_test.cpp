// host_test.cpp -- the reference's integration tests, re-expressed against the C++ host mirror
// (include/bevy_ggrs_hip.hpp).  Built twice by tests/test_cpp_host.py:
//   -DBACKEND_ORACLE : host logic on the CPU oracle (tests only; runs without a GPU)
//   (default)        : the product path, libggrs_hip.so on a gfx950 device
// Every test prints "ok <name>"; `particles` additionally prints every checksum so the two
// builds can be compared bit for bit.
#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "bevy_ggrs_hip.hpp"

using namespace bevy_ggrs;

#ifdef BACKEND_ORACLE
extern "C" {
void* gor_world_create(uint64_t, uint32_t, int);
void gor_world_destroy(void*);
const char* gor_last_error(void*);
int gor_register_component(void*, const char*, uint32_t, uint32_t, uint32_t*);
int gor_register_component_ex(void*, const char*, uint32_t, uint32_t, uint32_t, uint32_t*);
int gor_set_component_default(void*, uint32_t, const void*);
int gor_checksum_component(void*, uint32_t, const uint32_t*, uint32_t);
int gor_add_system(void*, const ggrs_system_desc*);
int gor_spawn(void*, uint64_t, uint64_t, const void* const*, uint64_t*);
int gor_download_word(void*, uint32_t, uint32_t, uint64_t, uint64_t, void*);
int gor_download_alive(void*, uint64_t*, uint64_t);
uint64_t gor_len(void*);
uint64_t gor_active_count(void*);
int32_t gor_frame(void*);
void gor_set_frame(void*, int32_t);
void gor_set_frame_rate(void*, uint64_t);
void gor_set_depth(void*, uint32_t);
void gor_set_confirmed(void*, int, int32_t);
int gor_has_snapshot(void*, int32_t);
uint64_t gor_snapshot_count(void*);
int gor_handle_requests(void*, const ggrs_request*, uint32_t, uint64_t*);
int gor_set_input_layout(void*, uint32_t, uint32_t);
}
struct OracleBackend {       // same surface as bevy_ggrs::HipBackend, bound to oracle/_build/libggrs_oracle.so
    void* w; int32_t cd = -1;
    OracleBackend(uint64_t capacity, uint32_t max_depth, int) : w(gor_world_create(capacity, max_depth, 0)) {}
    ~OracleBackend() { gor_world_destroy(w); }
    const char* last_error() { return gor_last_error(w); }
    int register_component(const char* n, uint32_t wb, uint32_t nw, uint32_t* id) { return gor_register_component(w, n, wb, nw, id); }
    int register_component_ex(const char* n, uint32_t wb, uint32_t nw, uint32_t flags, uint32_t* id) { return gor_register_component_ex(w, n, wb, nw, flags, id); }
    int set_component_default(uint32_t c, const void* p) { return gor_set_component_default(w, c, p); }
    int checksum_component(uint32_t c, const uint32_t* idx, uint32_t n) { return gor_checksum_component(w, c, idx, n); }
    int add_system(const ggrs_system_desc* d) { return gor_add_system(w, d); }
    int set_frame_rate(uint64_t fps) { gor_set_frame_rate(w, fps); return 0; }
    int spawn(uint64_t count, uint64_t mask, const void* const* cols, uint64_t* first) { return gor_spawn(w, count, mask, cols, first); }
    int set_depth(uint32_t d) { gor_set_depth(w, d); return 0; }
    int set_confirmed(int has, int32_t f) { gor_set_confirmed(w, has, f); return 0; }
    int set_synctest_check_distance(int32_t c) { cd = c; return 0; }
    int set_input_layout(uint32_t ib, uint32_t mp) { return gor_set_input_layout(w, ib, mp); }
    int handle_requests(const ggrs_request* r, uint32_t n, uint64_t* out) {
        uint32_t ns = 0;
        for (uint32_t i = 0; i < n; ++i) {        // schedule_systems.rs:204-220, applied per request
            if (cd >= 0 && gor_frame(w) - cd >= 0) gor_set_confirmed(w, 1, gor_frame(w) - cd);
            int rc = gor_handle_requests(w, r + i, 1, out + 2 * ns);
            if (rc) return rc;
            if (r[i].kind == GGRS_REQ_SAVE) ++ns;
        }
        return 0;
    }
    std::vector<std::vector<uint64_t>> stash;          // "enqueue" on the CPU oracle = run now, hand back on collect
    int enqueue_requests(const ggrs_request* r, uint32_t n) {
        uint32_t ns = 0; for (uint32_t i = 0; i < n; ++i) ns += r[i].kind == GGRS_REQ_SAVE;
        std::vector<uint64_t> out(2 * ns + 2);
        int rc = handle_requests(r, n, out.data());
        out.resize(2 * ns); stash.push_back(out);
        return rc;
    }
    int collect_checksums(uint64_t* out, uint32_t max_saves) {
        if (stash.empty() || stash.front().size() > 2 * (size_t)max_saves) return -1;
        std::copy(stash.front().begin(), stash.front().end(), out); stash.erase(stash.begin());
        return 0;
    }
    int32_t frame() { return gor_frame(w); }
    int set_frame(int32_t f) { gor_set_frame(w, f); return 0; }
    uint64_t len() { return gor_len(w); }
    int active_count(uint64_t* out) { *out = gor_active_count(w); return 0; }
    int download_word(uint32_t c, uint32_t word, uint64_t first, uint64_t count, void* dst) { return gor_download_word(w, c, word, first, count, dst); }
    int download_alive(uint64_t* dst, uint64_t n) { return gor_download_alive(w, dst, n); }
    int has_snapshot(int32_t f) { return gor_has_snapshot(w, f); }
    uint64_t snapshot_count() { return gor_snapshot_count(w); }
};
using Backend = OracleBackend;
#else
using Backend = HipBackend;
#endif

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

// ---- tests/common/mod.rs:7-55
using Config = GgrsConfig<uint8_t, size_t>;
using TestApp = App<Config, Backend>;

static void input_system(const LocalPlayers& players, LocalInputs<Config>& inputs) {
    for (auto h : players.handles) inputs[h] = 0;
}
static SyncTestSession<Config> synctest_session(size_t check_distance) {
    return SessionBuilder<Config>().with_num_players(1).with_check_distance(check_distance).add_player(PlayerType::Local, 0).start_synctest_session();
}
static void base_synctest_app(TestApp& app, size_t check_distance) {
    app.add_plugins(GgrsPlugin<Config>{});
    app.insert_resource(synctest_session(check_distance));
    app.add_systems(ReadInputs{}, input_system);
}

struct Health {};
struct Counter {};
struct Transform {};
struct Velocity {};
struct Ttl {};
namespace bevy_ggrs {
template <> struct HipComponent<Health> { static constexpr const char* name = "Health"; static constexpr uint32_t word_bytes = 4, n_words = 1; };
template <> struct HipComponent<Counter> { static constexpr const char* name = "Counter"; static constexpr uint32_t word_bytes = 4, n_words = 1; };
template <> struct HipComponent<Transform> { static constexpr const char* name = "Transform"; static constexpr uint32_t word_bytes = 4, n_words = 10; };
template <> struct HipComponent<Velocity> { static constexpr const char* name = "Velocity"; static constexpr uint32_t word_bytes = 4, n_words = 3; };
struct Player {};
template <> struct HipComponent<Player> { static constexpr const char* name = "Player"; static constexpr uint32_t word_bytes = 8, n_words = 1; };
template <> struct HipComponent<Ttl> { static constexpr const char* name = "Ttl"; static constexpr uint32_t word_bytes = 8, n_words = 1; };
}

// ggrs SyncTestSession::advance_frame request order
static void synctest_request_shape() {
    auto s = synctest_session(2);
    std::string shape;
    for (int t = 0; t < 5; ++t) {
        s.add_local_input(0, 0);
        for (auto& r : s.advance_frame()) {
            shape += r.kind == GgrsRequest<Config>::SaveGameState ? 'S' : r.kind == GgrsRequest<Config>::LoadGameState ? 'L' : 'A';
            if (r.kind == GgrsRequest<Config>::SaveGameState) r.cell->save(r.frame, nullptr, u128{1, 0});
        }
        shape += '|';
    }
    CHECK(shape == "SA|SA|SA|LASASA|LASASA|");
    bool threw = false;
    try { SessionBuilder<Config>().with_check_distance(8).start_synctest_session(); } catch (const GgrsError&) { threw = true; }
    CHECK(threw);                                         // check_distance >= max_prediction is rejected
    std::puts("ok synctest_request_shape");
}

// tests/synctest.rs:60-75
static void despawn_and_rollback_does_not_panic() {
    TestApp app(16);
    base_synctest_app(app, 5);
    app.rollback_component_with_copy<Health>();
    app.add_systems(GgrsSchedule{}, systems::saturating_sub_despawn<Health>(1));
    const uint32_t ten = 10;
    app.set_component_default<Health>(&ten);
    app.spawn(1, {"Health"});                             // Startup: commands.spawn((Health::default(), Rollback))
    for (int i = 0; i < 60; ++i) app.update();
    CHECK(app.active_count() == 0);                       // despawned and confirmed gone
    std::puts("ok despawn_and_rollback_does_not_panic");
}

// add_systems(GgrsSchedule, <the user's own system>): tests/synctest.rs:37-44's decrease_health written as HIP C++ source
// (CustomKernelSystem -> ggrs_hip_add_custom_system, compiled with hiprtc and inlined into the world's generated kernel) must
// do exactly what the built-in restatement does.  Product build only: the CPU oracle has no run-time compiler.
static void custom_system_equals_builtin() {
#ifndef BACKEND_ORACLE
    std::vector<uint32_t> got[2]; uint64_t active[2] = {0, 0};
    for (int custom = 0; custom < 2; ++custom) {
        TestApp app(512);
        base_synctest_app(app, 5);
        app.rollback_component_with_copy<Health>().checksum_component_with_hash<Health>();
        if (custom) {
            CustomKernelSystem sys("decrease_health",
                "__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {\n"
                "    const unsigned a = (unsigned)f.iparam[0];\n"
                "    e.u32(0) = e.u32(0) >= a ? e.u32(0) - a : 0u;\n"
                "    if (e.u32(0) == 0) e.despawn();\n"
                "}\n");
            sys.bind<Health>(0);
            sys.iparam[0] = 1;
            app.add_systems(GgrsSchedule{}, sys);
        } else {
            app.add_systems(GgrsSchedule{}, systems::saturating_sub_despawn<Health>(1));
        }
        std::vector<uint32_t> h(300);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 5 + (uint32_t)(i % 40);
        app.spawn(h.size(), {"Health"}, {h.data()});
        int fired = 0;
        app.add_observer([&](const SyncTestMismatch&) { ++fired; });
        for (int i = 0; i < 30; ++i) app.update();
        CHECK(fired == 0);
        got[custom] = app.download<Health, uint32_t>(0);
        active[custom] = app.active_count();
    }
    CHECK(got[0] == got[1]);
    CHECK(active[0] == active[1] && active[0] > 0 && active[0] < 300);
    bool threw = false;
    try {
        TestApp app(16);
        app.rollback_component_with_copy<Health>();
        CustomKernelSystem bad("broken", "__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) { e.u32(0) += nope; }");
        bad.bind<Health>(0);
        app.add_systems(GgrsSchedule{}, bad);
    } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("nope") != std::string::npos; }
    CHECK(threw);                                         // the compiler log reaches the caller
#endif
    std::puts("ok custom_system_equals_builtin");
}

// A system of the GgrsSchedule that SPAWNS Rollback entities (snapshot/rollback.rs:45-59) and a Strategy whose Stored is not the component (strategy.rs:22-40),
// through the plugin mirror: bullets fired every third frame, aged and despawned by a user-written system; an f32 x 3 component snapshotted as f16 x 3
struct Bullet {};
struct Accel {};
namespace bevy_ggrs {
template <> struct HipComponent<Bullet> { static constexpr const char* name = "Bullet"; static constexpr uint32_t word_bytes = 4, n_words = 3; };   // x, vx, life
template <> struct HipComponent<Accel> { static constexpr const char* name = "Accel"; static constexpr uint32_t word_bytes = 4, n_words = 3; };
}
static void user_written_spawner_and_strategy() {
#ifndef BACKEND_ORACLE
    constexpr uint32_t LIFE = 6;
    {
        TestApp app(4096);
        base_synctest_app(app, 5);
        app.rollback_component_with_copy<Bullet>().checksum_component_with_hash<Bullet>();
        CustomKernelSystem age("age_bullets",
            "__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f) {\n"
            "    e.f32(0) = e.f32(0) + e.f32(1) * f.dt;\n"
            "    if (e.u32(2) <= 1u) e.despawn(); else e.u32(2) -= 1u;\n"
            "}\n");
        age.bind<Bullet>(0).bind<Bullet>(1).bind<Bullet>(2);
        app.add_systems(GgrsSchedule{}, age);
        SpawnKernelSystem fire("fire",
            "__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload) {\n"
            "    float vx; __builtin_memcpy(&vx, payload, 4);\n"
            "    e.f32(0) = 0.0f; e.f32(1) = vx; e.u32(2) = (ggrs_u32)f.iparam[0];\n"
            "}\n");
        fire.with<Bullet>().bind<Bullet>(0).bind<Bullet>(1).bind<Bullet>(2).stride(4);
        fire.iparam[0] = LIFE;
        app.add_systems(GgrsSchedule{}, fire);
        uint64_t asked = 0;
        app.set_spawn_payload_source([&](Frame f, const PlayerInputs<Config>&, std::vector<uint8_t>& blob) -> uint64_t {
            ++asked;
            if (f % 3 != 0) return 0;                                           // a pure function of the frame: a resimulated frame fires the same shots
            const float v[2] = {0.5f * (float)f, 0.5f * (float)f + 1.0f};
            blob.resize(sizeof v); std::memcpy(blob.data(), v, sizeof v);
            return 2;
        });
        int fired = 0;
        app.add_observer([&](const SyncTestMismatch&) { ++fired; });
        const int updates = 30;
        for (int i = 0; i < updates; ++i) app.update();
        CHECK(fired == 0);
        CHECK(asked > (uint64_t)updates);                                       // resimulated frames asked again
        // frames 0 .. updates-1 ran once the session is done: 2 bullets per frame f with f % 3 == 0; a bullet spawned at the end of frame f is aged by
        // frames f+1 .. and despawned by the LIFE-th of them
        uint64_t spawned = 0, alive = 0;
        for (int f = 0; f < updates; ++f) if (f % 3 == 0) { spawned += 2; if (updates - 1 - f < (int)LIFE) alive += 2; }
        CHECK(app.len() == spawned);
        CHECK(app.active_count() == alive);
    }
    std::vector<uint32_t> got[2];
    for (int strategy = 0; strategy < 2; ++strategy) {
        TestApp app(256);
        base_synctest_app(app, 4);
        if (strategy)
            app.rollback_component_with_strategy<Accel>(2, 3,
                "__device__ unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short b; __builtin_memcpy(&b, &h, 2); return b; }\n"
                "__device__ float h2f(unsigned short b) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }\n"
                "__device__ void ggrs_store(const GgrsWords& t, GgrsWords& s) { for (int k = 0; k < 3; ++k) s.u16(k) = f2h(t.f32(k)); }\n"
                "__device__ void ggrs_load(const GgrsWords& s, GgrsWords& t) { for (int k = 0; k < 3; ++k) t.f32(k) = h2f(s.u16(k)); }\n");
        else
            app.rollback_component_with_copy<Accel>();
        app.checksum_component_with_hash<Accel>();
        CustomKernelSystem drift("drift", "__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame&) { e.f32(0) = e.f32(0) + 0.5f; }\n");
        drift.bind<Accel>(0);
        app.add_systems(GgrsSchedule{}, drift);
        std::vector<float> x(200), y(200), z(200);
        for (size_t i = 0; i < x.size(); ++i) { x[i] = 0.5f * (float)(i % 64); y[i] = -2.0f; z[i] = 0.25f * (float)(i % 7); }      // exact in f16: store / load are a bijection here
        app.spawn(x.size(), {"Accel"}, {x.data(), y.data(), z.data()});
        int fired = 0;
        app.add_observer([&](const SyncTestMismatch&) { ++fired; });
        for (int i = 0; i < 20; ++i) app.update();
        CHECK(fired == 0);
        got[strategy] = app.download<Accel, uint32_t>(0);
    }
    CHECK(got[0] == got[1] && !got[0].empty());
#endif
    std::puts("ok user_written_spawner_and_strategy");
}

// tests/synctest.rs:84-125: something that is NOT rolled back leaks into a checksummed component
static void mismatch_fires_on_non_determinism() {
    TestApp app(4096);
    base_synctest_app(app, 2);
    app.rollback_component_with_copy<Transform>().rollback_component_with_copy<Velocity>().rollback_component_with_copy<Ttl>();
    app.checksum_component_with_hash<Velocity>();
    app.add_systems(GgrsSchedule{}, systems::update_particles<Transform, Velocity>(0, -200, 0));
    app.add_systems(GgrsSchedule{}, systems::despawn_particles<Ttl>());
    app.add_systems(GgrsSchedule{}, systems::spawn_particles<Transform, Velocity, Ttl>(300, 1 << 4));
    app.add_systems(ReadInputs{}, [](const LocalPlayers& p, LocalInputs<Config>& in) { for (auto h : p.handles) in[h] = 1 << 4; });
    static std::atomic<uint32_t> global{0};               // never rolled back
    app.set_spawn_source([](Frame, std::vector<float>& vx, std::vector<float>& vy) {
        const float v = (float)global.fetch_add(1);
        vx.assign(4, v); vy.assign(4, -v);
    });
    int fired = 0;
    app.add_observer([&](const SyncTestMismatch& m) { ++fired; CHECK(!m.mismatched_frames.empty()); });
    for (int i = 0; i < 10; ++i) app.update();
    CHECK(fired > 0);
    std::puts("ok mismatch_fires_on_non_determinism");
}

// tests/synctest.rs:130-153
static void confirmed_frame_pruning() {
    TestApp app(16);
    base_synctest_app(app, 2);
    app.rollback_component_with_copy<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.spawn(1, {"Counter"});
    for (int i = 0; i < 20; ++i) app.update();
    CHECK(app.confirmed_frame_count() > 0);
    CHECK(!app.backend().has_snapshot(0));                // frame-0 snapshot pruned after confirmation
    CHECK(app.backend().has_snapshot(app.rollback_frame_count() - 1));
    std::puts("ok confirmed_frame_pruning");
}

// tests/component_rollback.rs:90-119: 20 ticks at cd = 2, observer panics on mismatch, value == RollbackFrameCount
static void component_rollback_copy() {
    TestApp app(16);
    base_synctest_app(app, 2);
    app.rollback_component_with_copy<Counter>().checksum_component_with_hash<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch"); });
    app.spawn(1, {"Counter"});
    for (int i = 0; i < 20; ++i) app.update();
    auto v = app.download<Counter, uint32_t>(0);
    CHECK(v.size() == 1 && (Frame)v[0] == app.rollback_frame_count() && v[0] == 20);
    std::puts("ok component_rollback_copy");
}

// tests/component_rollback.rs:131-163: #[component(immutable)] ImmutableTick, incremented by re-inserting
// ImmutableTick(tick + 1) every frame
static void immutable_component_copy_strategy_rolls_back() {
    TestApp app(16);
    base_synctest_app(app, 2);
    app.rollback_immutable_component_with_copy<Counter>().checksum_component_with_hash<Counter>().require_rollback<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch: immutable Copy component rollback is non-deterministic"); });
    app.spawn(1, {"Counter"});
    for (int i = 0; i < 20; ++i) app.update();
    auto v = app.download<Counter, uint32_t>(0);
    CHECK(v.size() == 1 && (Frame)v[0] == app.rollback_frame_count());
    std::puts("ok immutable_component_copy_strategy_rolls_back");
}

// tests/time.rs:18-49: a session is stopped after 30 frames and a new one started at frame 0
static void ggrs_time_survives_session_restart() {
    TestApp app(16);
    base_synctest_app(app, 2);
    app.rollback_component_with_copy<Counter>().checksum_component_with_hash<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch after a session restart"); });
    app.spawn(1, {"Counter"});
    for (int i = 0; i < 30; ++i) app.update();
    CHECK(app.ggrs_time_elapsed().count() > 0 && app.rollback_frame_count() == 30);
    app.remove_session();
    app.update();                                           // no-session branch: RollbackFrameCount back to 0
    CHECK(app.rollback_frame_count() == 0);
    app.insert_resource(synctest_session(2));
    app.update();
    const Frame frame = app.rollback_frame_count();
    CHECK(frame == 1);
    CHECK((uint64_t)app.ggrs_time_elapsed().count() == (uint64_t)frame * 1000000000ULL / 60);
    app.update();
    // the no-session branch re-inserted ConfirmedFrameCount(-1) (schedule_systems.rs:75): the stale confirmed
    // frame of the first session must not prune the new session's first snapshots
    CHECK(app.backend().has_snapshot(0) && app.backend().has_snapshot(1));
    for (int i = 0; i < 9; ++i) app.update();              // the stale ring of the first session must not confuse the new one
    CHECK(app.rollback_frame_count() == 11);
    auto v = app.download<Counter, uint32_t>(0);
    CHECK(v[0] == 30 + 11);                                 // the world itself is not reset by a session restart
    std::puts("ok ggrs_time_survives_session_restart");
}

// run_ggrs_schedules accumulator (src/schedule_systems.rs:19-83) + tests/time.rs:18-49
static void fixed_timestep_accumulator() {
    TestApp app(16);
    base_synctest_app(app, 2);
    app.rollback_component_with_copy<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.spawn(1, {"Counter"});
    const uint64_t frame_ns = 1000000000ULL / 60;
    app.update(std::chrono::nanoseconds(frame_ns / 2));   CHECK(app.rollback_frame_count() == 0);   // not enough time accumulated
    app.update(std::chrono::nanoseconds(frame_ns / 2 + 1)); CHECK(app.rollback_frame_count() == 1);
    app.update(std::chrono::nanoseconds(frame_ns * 5 / 2)); CHECK(app.rollback_frame_count() == 3);  // 2 whole steps, half a frame kept
    app.update(std::chrono::nanoseconds(frame_ns / 2 + 2)); CHECK(app.rollback_frame_count() == 4);
    TestApp idle(16);                                      // no session inserted: counters reset, nothing runs
    idle.add_plugins(GgrsPlugin<Config>{});
    idle.update();
    CHECK(idle.rollback_frame_count() == 0 && idle.confirmed_frame_count() == -1 && idle.max_prediction_window() == 8);
    std::puts("ok fixed_timestep_accumulator");
}


// ---- tests/resource_lifecycle.rs:19-32
struct Wallet { uint32_t v; };
struct FrameLog { uint32_t v = 0; };
static void ggrs_hash(const FrameLog& f, SeaHasher& h) { h.write_u32(f.v); }       // #[derive(Hash)] struct FrameLog(u32)
static void track_wallet(TestApp& app, const PlayerInputs<Config>&) {              // resource_lifecycle.rs:30-32
    app.resource<FrameLog>().v += app.get_resource<Wallet>() ? 2u : 1u;
}
static void resource_app(TestApp& app) {
    base_synctest_app(app, 4);
    app.init_resource<FrameLog>();
    app.rollback_resource_with_clone<Wallet>().rollback_resource_with_clone<FrameLog>().checksum_resource_with_hash<FrameLog>();
    app.rollback_component_with_copy<Counter>();                                   // the device world needs one registered column
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.spawn(1, {"Counter"});
}
// tests/resource_lifecycle.rs:43-80
static void resource_inserted_mid_session_rolls_back() {
    TestApp app(16);
    resource_app(app);
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem([](TestApp& a, const PlayerInputs<Config>&) { if (a.rollback_frame_count() == 3) a.insert_resource(Wallet{100}); }));
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem(track_wallet));
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch: Wallet rollback (insert) is non-deterministic"); });
    for (int i = 0; i < 20; ++i) app.update();
    CHECK(app.get_resource<Wallet>() != nullptr && app.resource<Wallet>().v == 100);
    CHECK(app.rollback_frame_count() == 20);
    CHECK(app.resource<FrameLog>().v == 2 * 1 + 18 * 2);                           // frames 1, 2 without the wallet, 3..20 with it
    std::puts("ok resource_inserted_mid_session_rolls_back");
}
// tests/resource_lifecycle.rs:89-120
static void resource_removed_mid_session_rolls_back() {
    TestApp app(16);
    resource_app(app);
    app.insert_resource(Wallet{100});
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem([](TestApp& a, const PlayerInputs<Config>&) { if (a.rollback_frame_count() == 3) a.remove_resource<Wallet>(); }));
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem(track_wallet));
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch: Wallet rollback (remove) is non-deterministic"); });
    for (int i = 0; i < 20; ++i) app.update();
    CHECK(app.get_resource<Wallet>() == nullptr);
    CHECK(app.resource<FrameLog>().v == 2 * 2 + 18 * 1);
    std::puts("ok resource_removed_mid_session_rolls_back");
}
// the detection strategy of resource_lifecycle.rs:9-12, negated: a Wallet that is NOT registered for rollback
// survives LoadWorld, FrameLog accumulates differently during re-simulation, SyncTestMismatch fires
static void resource_without_rollback_fires_mismatch() {
    TestApp app(16);
    base_synctest_app(app, 4);
    app.init_resource<FrameLog>();
    app.rollback_resource_with_clone<FrameLog>().checksum_resource_with_hash<FrameLog>();
    app.rollback_component_with_copy<Counter>();
    app.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    app.spawn(1, {"Counter"});
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem([](TestApp& a, const PlayerInputs<Config>&) { if (a.rollback_frame_count() == 3) a.insert_resource(Wallet{100}); }));
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem(track_wallet));
    int fired = 0;
    app.add_observer([&](const SyncTestMismatch&) { ++fired; });
    for (int i = 0; i < 12; ++i) app.update();
    CHECK(fired > 0);
    std::puts("ok resource_without_rollback_fires_mismatch");
}
// resource checksum parts take part in the frame checksum exactly as checksum.rs:88-99 folds them, in both the
// synchronous and the pipelined mode
static void resource_checksum_part_is_folded() {
    std::vector<u128> seen[2];
    for (int pipelined = 0; pipelined < 2; ++pipelined) {
        TestApp app(16);
        resource_app(app);
        app.add_systems(GgrsSchedule{}, TestApp::HostSystem(track_wallet));
        app.set_pipelined(pipelined != 0);
        for (int i = 0; i < 8; ++i) { app.update(); app.flush(); for (auto& c : app.last_checksums()) seen[pipelined].push_back(c); }
    }
    CHECK(seen[0].size() == seen[1].size() && !seen[0].empty());
    for (size_t i = 0; i < seen[0].size(); ++i) CHECK(seen[0][i] == seen[1][i]);
    // the same world without the FrameLog part differs by exactly that part: SeaHasher(write_u32(v))
    TestApp plain(16);
    base_synctest_app(plain, 4);
    plain.rollback_component_with_copy<Counter>();
    plain.add_systems(GgrsSchedule{}, systems::add_u32<Counter>(1));
    plain.spawn(1, {"Counter"});
    plain.update();                                                                // Save(0) then Advance
    SeaHasher h; h.write_u32(0);
    CHECK(plain.last_checksums().size() == 1 && (plain.last_checksums()[0].lo ^ h.finish()) == seen[0][0].lo);
    std::puts("ok resource_checksum_part_is_folded");
}

// seahash published vector + the SURVEY 8c derived vectors, on the HOST hasher of the mirror
static void host_seahasher_known_answers() {
    SeaHasher a; a.write("to be or not to be", 18);
    CHECK(a.finish() == 1988685042348123509ULL);
    SeaHasher b; b.write("to be or ", 9); b.write("not to be", 9);                 // chunking must not matter
    CHECK(b.finish() == 1988685042348123509ULL);
    SeaHasher c; c.write_u32(42);                                                   // ChecksumPart::from_value(&42u32), checksum.rs:38-44
    CHECK(c.finish() == 0x352173bd5a4ba44bULL);
    SeaHasher d; d.write_u64(1); d.write_u64(1);                                    // entity checksum (active = 1, total = 1)
    CHECK(d.finish() == 0x7c846906b6e5a068ULL);
    std::puts("ok host_seahasher_known_answers");
}

// src/snapshot/mod.rs:349-512, the ring's own unit tests, on the host ring that holds the resources
static void host_ring_known_answers() {
    using Ring = GgrsSnapshots<int>;
    { Ring r; r.set_depth(3); for (int f = 0; f < 5; ++f) r.push(f, f * 10);              // mod.rs:369-380
      CHECK(r.size() == 3 && !r.peek(0) && !r.peek(1) && *r.peek(2) == 20 && *r.peek(4) == 40); }
    { Ring r; r.set_depth(8); for (int f = 0; f < 5; ++f) r.push(f, f * 10); r.push(2, 99); // mod.rs:384-394
      CHECK(r.size() == 3 && *r.peek(2) == 99 && !r.peek(3) && !r.peek(4) && *r.peek(1) == 10); }
    { Ring r; r.set_depth(8); r.push(0, 1); r.push(0, 2); CHECK(r.size() == 1 && *r.peek(0) == 2); }   // mod.rs:398-403
    { Ring r; r.set_depth(8); for (int f = 0; f < 5; ++f) r.push(f, f); r.confirm(3);      // mod.rs:409-422
      CHECK(!r.peek(0) && !r.peek(1) && !r.peek(2) && r.peek(3) && r.peek(4)); }
    { Ring r; r.set_depth(8); for (int f = 0; f < 3; ++f) r.push(f, f); r.confirm(100); CHECK(r.size() == 0); }   // mod.rs:426-435
    { Ring r; r.confirm(5); CHECK(r.size() == 0); }                                         // mod.rs:439-442
    { Ring r; r.set_depth(8); for (int f = 0; f < 5; ++f) r.push(f, f * 10);               // mod.rs:448-468
      CHECK(r.rollback(2).get() == 20); CHECK(!r.peek(3) && !r.peek(4) && r.peek(1)); }
    { Ring r; r.set_depth(8); r.push(0, 0); r.push(1, 1); bool threw = false;              // mod.rs:472-477
      try { r.rollback(5); } catch (const std::runtime_error&) { threw = true; } CHECK(threw); }
    { Ring r; r.set_depth(8); r.push(INT32_MAX - 1, 1); r.push(INT32_MAX, 2); r.push(INT32_MIN, 3);   // mod.rs:485-497
      CHECK(r.size() == 3 && *r.peek(INT32_MIN) == 3 && *r.peek(INT32_MAX) == 2); }
    { Ring r; r.set_depth(8); r.push(INT32_MIN, 1); r.push(INT32_MAX, 2);                  // mod.rs:502-512
      CHECK(r.size() == 1 && !r.peek(INT32_MIN) && *r.peek(INT32_MAX) == 2); }
    std::puts("ok host_ring_known_answers");
}

// examples/box_game/box_game_synctest.rs (`--num-players 2 --check-distance 7`, examples/README.md:64) with
// box_game.rs's setup_system / move_cube_system / increase_frame_system; prints every checksum and the cubes
struct FrameCount { uint32_t frame = 0; };                                           // box_game.rs:48-53
static void ggrs_hash(const FrameCount& f, SeaHasher& h) { h.write_u32(f.frame); }
static void box_game_synctest(size_t num_players, size_t check_distance, int updates) {
    TestApp app(16);
    auto sess = SessionBuilder<Config>().with_num_players(num_players).with_check_distance(check_distance).with_input_delay(2);
    for (size_t i = 0; i < num_players; ++i) sess.add_player(PlayerType::Local, i);
    app.add_plugins(GgrsPlugin<Config>{});
    app.insert_resource(RollbackFrameRate{60});
    int tick = 0;
    app.add_systems(ReadInputs{}, [&](const LocalPlayers& p, LocalInputs<Config>& in) {   // read_local_inputs with a scripted keyboard
        for (auto h : p.handles) in[h] = (uint8_t)((tick * 7 + (int)h * 3) % 16);
    });
    app.rollback_resource_with_copy<FrameCount>();
    app.rollback_component_with_copy<Velocity>().rollback_component_with_clone<Transform>().plain_component<Player>();
    app.checksum_resource_with_hash<FrameCount>();
    app.add_systems(GgrsSchedule{}, systems::move_cube_system<Transform, Velocity, Player>());
    app.add_systems(GgrsSchedule{}, TestApp::HostSystem([](TestApp& a, const PlayerInputs<Config>&) { a.resource<FrameCount>().frame += 1; }));
    app.insert_resource(sess.start_synctest_session());
    app.insert_resource(FrameCount{0});
    app.add_observer([](const SyncTestMismatch& m) { std::fprintf(stderr, "desync detected at frame %d!\n", m.current_frame); std::exit(1); });
    // setup_system (box_game.rs:89-131)
    std::vector<float> col[10]; std::vector<float> vel[3]; std::vector<uint64_t> handle;
    const float r = 5.0f / 4.0f;
    for (size_t h = 0; h < num_players; ++h) {
        const float rot = (float)h / (float)num_players * 2.0f * 3.14159265358979323846f;
        const float t[10] = {r * std::cos(rot), 0.2f / 2.0f, r * std::sin(rot), 0, 0, 0, 1, 1, 1, 1};
        for (int k = 0; k < 10; ++k) col[k].push_back(t[k]);
        for (int k = 0; k < 3; ++k) vel[k].push_back(0.0f);
        handle.push_back(h);
    }
    std::vector<const void*> cols;                                               // ascending component id: Velocity, Transform, Player
    for (int k = 0; k < 3; ++k) cols.push_back(vel[k].data());
    for (int k = 0; k < 10; ++k) cols.push_back(col[k].data());
    cols.push_back(handle.data());
    app.spawn(num_players, {"Transform", "Velocity", "Player"}, cols);
    for (tick = 0; tick < updates; ++tick) {
        app.update();
        for (auto& c : app.last_checksums()) std::printf("box checksum %d %016llx%016llx\n", tick, (unsigned long long)c.hi, (unsigned long long)c.lo);
    }
    CHECK(app.resource<FrameCount>().frame == (uint32_t)app.rollback_frame_count());   // the resource rolled back and counted up again
    { auto y = app.download<Transform, uint32_t>(1); float f; std::memcpy(&f, &y[0], 4); CHECK(f == 0.1f); }   // CUBE_SIZE / 2: nothing moves y
    for (uint32_t k = 0; k < 3; ++k) {
        auto x = app.download<Transform, uint32_t>(k); auto v = app.download<Velocity, uint32_t>(k);
        for (size_t h = 0; h < num_players; ++h) std::printf("box cube %zu word %u t %08x v %08x\n", h, k, x[h], v[h]);
    }
    std::puts("ok box_game_synctest");
}

// examples/stress_tests/particles.rs:187-240 through the plugin API; prints every checksum
static void particles(uint64_t n, int ticks, size_t cd, bool pipelined) {
    TestApp app(n + 100 * (uint64_t)ticks + 64);
    base_synctest_app(app, cd);
    app.set_pipelined(pipelined);
    app.insert_resource(RollbackFrameRate{60});
    app.rollback_component_with_clone<Transform>().rollback_component_with_copy<Velocity>().rollback_component_with_copy<Ttl>();
    app.checksum_component_with_hash<Velocity>();
    app.checksum_component<Transform>({0, 1, 2});          // translation only (particles.rs:207-222)
    const float tdef[10] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
    app.set_component_default<Transform>(tdef);
    app.add_systems(GgrsSchedule{}, systems::update_particles<Transform, Velocity>(0, -200, 0));
    app.add_systems(GgrsSchedule{}, systems::despawn_particles<Ttl>());
    app.add_systems(GgrsSchedule{}, systems::spawn_particles<Transform, Velocity, Ttl>(40, 1 << 4));
    int tick_no = 0;
    app.add_systems(ReadInputs{}, [&](const LocalPlayers& p, LocalInputs<Config>& in) { for (auto h : p.handles) in[h] = (tick_no % 3 == 1) ? (1 << 4) : 0; });
    app.set_spawn_source([](Frame f, std::vector<float>& vx, std::vector<float>& vy) {   // pure function of the frame
        vx.resize(100); vy.resize(100);
        uint32_t s = 0x9E3779B9u * (uint32_t)(f + 1);
        for (int i = 0; i < 100; ++i) { s = s * 1664525u + 1013904223u; vx[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f; s = s * 1664525u + 1013904223u; vy[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f; }
    });
    app.add_observer([](const SyncTestMismatch&) { CHECK(!"SyncTestMismatch"); });
    std::vector<float> vx(n), vy(n), vz(n, 0.0f);
    std::vector<uint64_t> ttl(n);
    uint32_t s = 123;
    for (uint64_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u; vx[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f;
        s = s * 1664525u + 1013904223u; vy[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f;
        ttl[i] = 1 + i % 300;
    }
    std::vector<const void*> cols(10, nullptr);            // Transform: defaults
    cols.push_back(vx.data()); cols.push_back(vy.data()); cols.push_back(vz.data()); cols.push_back(ttl.data());
    app.spawn(n, {"Transform", "Velocity", "Ttl"}, cols);
    for (tick_no = 0; tick_no < ticks; ++tick_no) {
        app.update();
        if (pipelined) app.flush();                        // (a real host would only flush at the next tick)
        for (auto& c : app.last_checksums()) std::printf("checksum %d %016llx%016llx\n", tick_no, (unsigned long long)c.hi, (unsigned long long)c.lo);
    }
    auto x = app.download<Transform, uint32_t>(1);
    uint64_t fold = 0;
    for (size_t i = 0; i < x.size(); ++i) fold = fold * 1099511628211ULL + x[i];
    std::printf("final frame %d len %llu active %llu ty_fold %016llx\n", app.rollback_frame_count(), (unsigned long long)app.backend().len(),
                (unsigned long long)app.active_count(), (unsigned long long)fold);
    std::puts(pipelined ? "ok particles_pipelined" : "ok particles");
}


// bevy_ggrs::SpeculativeFanout (ggrs_hip_fanout_*) from C++ at world size 1: a compact step of 3 branches x 4 frames with every frame KEPT, then the branch
// whose prediction came true is adopted -- and the world must be what a second world reaches by simulating those frames in a straight line.
// Product build only (the collectives and the branch blocks live inside libggrs_hip.so).
static void speculative_fanout_adopts_matching_branch() {
#ifndef BACKEND_ORACLE
    auto build = [](TestApp& app, uint64_t n) {
        app.insert_resource(RollbackFrameRate{60});
        app.rollback_component_with_clone<Transform>().rollback_component_with_copy<Velocity>().rollback_component_with_copy<Ttl>();
        app.checksum_component_with_hash<Velocity>();
        app.checksum_component<Transform>({0, 1, 2});
        const float tdef[10] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
        app.set_component_default<Transform>(tdef);
        app.add_systems(GgrsSchedule{}, systems::update_particles<Transform, Velocity>(0, -200, 0));
        app.add_systems(GgrsSchedule{}, systems::despawn_particles<Ttl>());
        std::vector<float> vx(n), vy(n), vz(n, 0.0f); std::vector<uint64_t> ttl(n);
        uint32_t s = 77;
        for (uint64_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; vx[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f; s = s * 1664525u + 1013904223u; vy[i] = (float)(int32_t)(s >> 8) / 41943.04f - 200.0f; ttl[i] = 2 + i % 5; }
        std::vector<const void*> cols(10, nullptr);
        cols.push_back(vx.data()); cols.push_back(vy.data()); cols.push_back(vz.data()); cols.push_back(ttl.data());
        app.spawn(n, {"Transform", "Velocity", "Ttl"}, cols);
    };
    const uint64_t n = 3000;
    TestApp a(n + 64), b(n + 64);
    build(a, n); build(b, n);
    a.backend().set_depth(6); b.backend().set_depth(6);
    SpeculativeFanout fan(a.backend(), SpeculativeFanout::unique_id(), 0, 1);
    ggrs_request save0; std::memset(&save0, 0, sizeof save0); save0.kind = GGRS_REQ_SAVE; save0.frame = 0;
    const uint32_t B = 3, T = 4;
    std::vector<uint8_t> inputs(B * T, 0);
    const uint32_t ns = fan.step_branches({save0}, B, T, 1, inputs, GGRS_BRANCH_RETAIN_ALL);
    CHECK(ns == 1 + B * (T - 1));
    uint32_t steps = 0, saves = 0;
    const std::vector<u128> table = fan.collect(&steps, &saves);
    CHECK(steps == 1 && saves == ns && table.size() == ns);
    CHECK(a.backend().frame() == 0);                                   // branches are speculation: the world is where the prefix left it
    // straight line on the second world: three frames, then the snapshot of frame 3
    ggrs_request adv; std::memset(&adv, 0, sizeof adv); adv.kind = GGRS_REQ_ADVANCE; const uint8_t zero = 0; adv.inputs = &zero; adv.n_inputs = 1;
    ggrs_request save3 = save0; save3.frame = 3;
    std::vector<ggrs_request> line = {adv, adv, adv, save3};
    uint64_t want[2] = {0, 0};
    CHECK(b.backend().handle_requests(line.data(), (uint32_t)line.size(), want) == GGRS_OK);
    for (uint32_t br = 0; br < B; ++br) CHECK(table[1 + br * (T - 1) + 2].lo == want[0] && table[1 + br * (T - 1) + 2].hi == want[1]);   // every branch predicted the same inputs here
    fan.adopt(1, 3, line);
    CHECK(a.backend().frame() == 3 && a.backend().has_snapshot(3));
    uint64_t got[2] = {0, 0};
    CHECK(a.backend().handle_requests(&save3, 1, got) == GGRS_OK);
    CHECK(got[0] == want[0] && got[1] == want[1]);
    const auto ya = a.download<Transform, uint32_t>(1), yb = b.download<Transform, uint32_t>(1);
    CHECK(ya == yb && a.active_count() == b.active_count() && a.active_count() < n);
    bool threw = false;
    try { fan.adopt(0, 2, line); } catch (const std::runtime_error&) { threw = true; }      // the speculation is history once a branch was adopted
    CHECK(threw);
#endif
    std::puts("ok speculative_fanout_adopts_matching_branch");
}

// The two restatements of ggrs's SyncTestSession::advance_frame (this header and bevy_ggrs_amd/session.py) must
// emit the same requests: printed here, compared by tests/test_cpp_host.py::test_synctest_sessions_agree
static void print_request_traces() {
    for (size_t cd : {0, 1, 3, 7}) for (size_t delay : {0, 2}) {
        auto b = SessionBuilder<Config>().with_num_players(2).with_check_distance(cd).with_input_delay(delay);
        auto s = b.start_synctest_session();
        for (int t = 0; t < 14; ++t) {
            for (PlayerHandle h = 0; h < 2; ++h) s.add_local_input(h, (uint8_t)((t * 5 + (int)h * 3) & 15));
            std::printf("trace cd=%zu delay=%zu t=%d:", cd, delay, t);
            for (auto& r : s.advance_frame()) {
                if (r.kind == GgrsRequest<Config>::SaveGameState) { std::printf(" S%d", r.frame); r.cell->save(r.frame, nullptr, u128{(uint64_t)r.frame * 7 + 1, 0}); }
                else if (r.kind == GgrsRequest<Config>::LoadGameState) std::printf(" L%d", r.frame);
                else { std::printf(" A"); for (auto& in : r.inputs) std::printf("%s%d", &in == &r.inputs[0] ? "" : ",", (int)in.first); }
            }
            std::printf("\n");
        }
    }
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 5000;
    synctest_request_shape();
    print_request_traces();
    despawn_and_rollback_does_not_panic();
    custom_system_equals_builtin();
    user_written_spawner_and_strategy();
    speculative_fanout_adopts_matching_branch();
    mismatch_fires_on_non_determinism();
    confirmed_frame_pruning();
    component_rollback_copy();
    immutable_component_copy_strategy_rolls_back();
    fixed_timestep_accumulator();
    ggrs_time_survives_session_restart();
    host_seahasher_known_answers();
    host_ring_known_answers();
    resource_inserted_mid_session_rolls_back();
    resource_removed_mid_session_rolls_back();
    resource_without_rollback_fires_mismatch();
    resource_checksum_part_is_folded();
    box_game_synctest(2, 7, 40);
    particles(n, 24, 7, false);
    particles(n, 24, 7, true);
    return 0;
}
