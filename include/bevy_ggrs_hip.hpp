// bevy_ggrs_hip.hpp -- host-side mirror of bevy_ggrs's plugin interface over the C ABI of
// libggrs_hip.so (include/ggrs_hip.h).  Header-only C++17.
//
// The reference (/root/reference, Rust) cannot be built in this image, so the host side above the
// C ABI is written in C++ with the reference's own names, argument meaning and error behaviour for
// the snapshot-and-resimulate path:
//
//   reference (file:line)                                         here
//   GgrsConfig<Input, Address, State>        src/lib.rs:45-60     bevy_ggrs::GgrsConfig<...>
//   GgrsPlugin<C>                            src/lib.rs:200-260   bevy_ggrs::GgrsPlugin<C>
//   GgrsSchedule / ReadInputs labels         src/lib.rs:76,151    bevy_ggrs::GgrsSchedule / ReadInputs
//   Session<C>, PlayerInputs, LocalInputs,   src/lib.rs:81-147    same names
//   LocalPlayers, SyncTestMismatch
//   RollbackFrameRate                        src/time.rs:20       same
//   RollbackFrameCount / ConfirmedFrameCount src/snapshot/mod.rs:70,80   App::rollback_frame_count() ...
//   RollbackApp::rollback_component_with_*   src/snapshot/rollback_app.rs:31-133   App member functions
//   run_ggrs_schedules / run_synctest /      src/schedule_systems.rs:19-118,170-289  App::update / handle_requests
//   handle_requests
//   RollbackApp::rollback_resource_with_*,   src/snapshot/rollback_app.rs:46-124     App member functions; resources are O(1)
//   checksum_resource[_with_hash]            resource_snapshot.rs, resource_checksum.rs  bytes per frame and stay on the host
//   GgrsSnapshots<For, As>                   src/snapshot/mod.rs:97-274              bevy_ggrs::GgrsSnapshots<As> (host ring for resources)
//   checksum_hasher() == SeaHasher::new()    src/snapshot/mod.rs:318-320             bevy_ggrs::SeaHasher (seahash 4.1, restated)
//   ggrs::SessionBuilder / SyncTestSession   (un-vendored `ggrs`, Cargo.toml:23; restated)  same names
//
// User systems in the reference are arbitrary Rust closures; on this path a GgrsSchedule system is a
// *kernel-backed* system descriptor (bevy_ggrs::systems::*) executed inside libggrs_hip.so.
// P2P / spectator sessions (UDP, inside ggrs) are out of scope.
//
// `Backend` abstracts the C ABI so the host logic can be exercised on CPU in tests/ (tests/cpp binds
// it to the oracle); the default and only product backend is HipBackend == libggrs_hip.so.
#pragma once

#include <chrono>
#include <deque>
#include <memory>
#include <typeindex>
#include <typeinfo>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <variant>
#include <vector>

#include "ggrs_hip.h"

namespace bevy_ggrs {

using Frame = int32_t;                 // ggrs::Frame
using PlayerHandle = size_t;           // ggrs::PlayerHandle
constexpr Frame NULL_FRAME = -1;       // ggrs::NULL_FRAME
constexpr size_t DEFAULT_FPS = 60;     // src/lib.rs:62

enum class InputStatus { Confirmed, Predicted, Disconnected };
enum class PlayerType { Local, Remote, Spectator };

// ---------------------------------------------------------------- src/lib.rs:45-60
template <class InputT, class AddressT = size_t, class StateT = uint8_t>
struct GgrsConfig {
    using Input = InputT;
    using Address = AddressT;
    using State = StateT;
};

struct GgrsSchedule {};                // src/lib.rs:76
struct ReadInputs {};                  // src/lib.rs:151
struct RollbackFrameRate { size_t fps = DEFAULT_FPS; };   // src/time.rs:20

struct SyncTestMismatch {              // src/lib.rs:133-139
    Frame current_frame;
    std::vector<Frame> mismatched_frames;
};

template <class C> using PlayerInputs = std::vector<std::pair<typename C::Input, InputStatus>>;  // src/lib.rs:98
template <class C> using LocalInputs = std::unordered_map<PlayerHandle, typename C::Input>;      // src/lib.rs:143
struct LocalPlayers { std::vector<PlayerHandle> handles; };                                       // src/lib.rs:147

// ---------------------------------------------------------------- ggrs (restated)
struct GgrsError : std::runtime_error {
    enum Kind { InvalidRequest, MismatchedChecksum, PredictionThreshold, NotSynchronized };
    Kind kind;
    Frame current_frame = NULL_FRAME;
    std::vector<Frame> mismatched_frames;
    GgrsError(Kind k, const std::string& what) : std::runtime_error(what), kind(k) {}
};

struct u128 {                          // Checksum(u128), src/snapshot/checksum.rs:49
    uint64_t lo = 0, hi = 0;
    bool operator==(const u128& o) const { return lo == o.lo && hi == o.hi; }
    bool operator!=(const u128& o) const { return !(*this == o); }
};

struct GameStateCell {                 // ggrs::GameStateCell: only frame + checksum are used (schedule_systems.rs:235-236)
    Frame frame = NULL_FRAME;
    std::optional<u128> checksum;
    void save(Frame f, std::nullptr_t, std::optional<u128> cs) { frame = f; checksum = cs; }
};

template <class C>
struct GgrsRequest {
    enum Kind { SaveGameState = GGRS_REQ_SAVE, LoadGameState = GGRS_REQ_LOAD, AdvanceFrame = GGRS_REQ_ADVANCE } kind;
    Frame frame = NULL_FRAME;          // Save / Load
    GameStateCell* cell = nullptr;     // Save
    PlayerInputs<C> inputs;            // Advance
};

// SyncTestSession::advance_frame, restated from ggrs (parity-unpinned by reference vectors; the
// reference tests that constrain it: tests/synctest.rs:84-153, tests/component_rollback.rs).
template <class C>
class SyncTestSession {
  public:
    using Input = typename C::Input;
    SyncTestSession(size_t num_players, size_t check_distance, size_t max_prediction, size_t input_delay)
        : num_players_(num_players), check_distance_(check_distance), max_prediction_(max_prediction),
          input_delay_(input_delay), cells_(std::max(max_prediction, check_distance) + 2) {}

    size_t num_players() const { return num_players_; }
    size_t max_prediction() const { return max_prediction_; }
    size_t check_distance() const { return check_distance_; }
    Frame current_frame() const { return current_frame_; }

    void add_local_input(PlayerHandle handle, Input input) {
        if (handle >= num_players_) throw GgrsError(GgrsError::InvalidRequest, "The player handle you provided is not referring to a local player.");
        local_inputs_[handle] = input;
    }

    std::vector<GgrsRequest<C>> advance_frame() {
        std::vector<GgrsRequest<C>> requests;
        const Frame cur = current_frame_;
        const Frame d = (Frame)check_distance_;
        if (d > 0 && cur > d) {
            std::vector<Frame> mismatched;
            for (Frame f = cur - d; f <= cur; ++f) if (!checksums_consistent(f)) mismatched.push_back(f);
            if (!mismatched.empty()) {
                GgrsError e(GgrsError::MismatchedChecksum, "Detected checksum mismatch during rollback on frame " + std::to_string(cur));
                e.current_frame = cur; e.mismatched_frames = mismatched;
                throw e;
            }
            // adjust_gamestate: roll back d frames and resimulate
            const Frame frame_to = cur - d;
            GgrsRequest<C> load; load.kind = GgrsRequest<C>::LoadGameState; load.frame = frame_to;
            requests.push_back(load);
            current_frame_ = frame_to;
            for (Frame i = 0; i < d; ++i) {
                if (i > 0) requests.push_back(save_request());
                requests.push_back(advance_request(inputs_for(current_frame_)));
                current_frame_ += 1;
            }
        }
        if (local_inputs_.size() != num_players_) throw GgrsError(GgrsError::InvalidRequest, "Missing local input while calling advance_frame().");
        std::vector<Input> vals(num_players_);
        for (auto& kv : local_inputs_) vals[kv.first] = kv.second;
        pending_[current_frame_ + (Frame)input_delay_] = vals;
        local_inputs_.clear();
        if (d > 0) requests.push_back(save_request());
        requests.push_back(advance_request(inputs_for(current_frame_)));
        current_frame_ += 1;
        for (auto it = history_.begin(); it != history_.end();) it = (it->first < current_frame_ - d - 2) ? history_.erase(it) : std::next(it);
        for (auto it = pending_.begin(); it != pending_.end();) it = (it->first < current_frame_ - d - 2) ? pending_.erase(it) : std::next(it);
        return requests;
    }

  private:
    const std::vector<Input>& inputs_for(Frame f) {
        auto it = history_.find(f);
        if (it == history_.end()) {
            auto p = pending_.find(f);
            it = history_.emplace(f, p != pending_.end() ? p->second : std::vector<Input>(num_players_, Input{})).first;
        }
        return it->second;
    }
    GgrsRequest<C> save_request() {
        GgrsRequest<C> r; r.kind = GgrsRequest<C>::SaveGameState; r.frame = current_frame_;
        r.cell = &cells_[(size_t)current_frame_ % cells_.size()];
        return r;
    }
    GgrsRequest<C> advance_request(const std::vector<Input>& in) {
        GgrsRequest<C> r; r.kind = GgrsRequest<C>::AdvanceFrame;
        for (auto& v : in) r.inputs.emplace_back(v, InputStatus::Confirmed);
        return r;
    }
    bool checksums_consistent(Frame frame_to_check) {
        const Frame oldest = current_frame_ - (Frame)check_distance_;
        for (auto it = checksum_history_.begin(); it != checksum_history_.end();) it = (it->first < oldest) ? checksum_history_.erase(it) : std::next(it);
        const GameStateCell& cell = cells_[(size_t)frame_to_check % cells_.size()];
        if (cell.frame != frame_to_check) return true;
        auto it = checksum_history_.find(cell.frame);
        if (it != checksum_history_.end()) return it->second == cell.checksum;
        checksum_history_[cell.frame] = cell.checksum;
        return true;
    }

    size_t num_players_, check_distance_, max_prediction_, input_delay_;
    Frame current_frame_ = 0;
    std::map<PlayerHandle, Input> local_inputs_;
    std::map<Frame, std::vector<Input>> history_, pending_;
    std::vector<GameStateCell> cells_;
    std::map<Frame, std::optional<u128>> checksum_history_;
};

template <class C>
class SessionBuilder {                 // ggrs::SessionBuilder (knobs used by the reference's examples/tests)
  public:
    SessionBuilder& with_num_players(size_t n) { num_players_ = n; return *this; }
    SessionBuilder& with_check_distance(size_t d) { check_distance_ = d; return *this; }
    SessionBuilder& with_input_delay(size_t d) { input_delay_ = d; return *this; }
    SessionBuilder& with_max_prediction_window(size_t w) { max_prediction_ = w; return *this; }
    SessionBuilder& add_player(PlayerType, PlayerHandle h) {
        if (h >= num_players_) throw GgrsError(GgrsError::InvalidRequest, "The player handle you provided is invalid.");
        return *this;
    }
    SyncTestSession<C> start_synctest_session() const {
        if (check_distance_ >= max_prediction_) throw GgrsError(GgrsError::InvalidRequest, "Check distance too big.");
        return SyncTestSession<C>(num_players_, check_distance_, max_prediction_, input_delay_);
    }
  private:
    size_t num_players_ = 2, check_distance_ = 2, max_prediction_ = 8, input_delay_ = 0;
};

// src/lib.rs:81-88.  P2P / Spectator variants live in ggrs's UDP layer: out of scope.
template <class C> using Session = std::variant<std::monostate, SyncTestSession<C>>;


// ---------------------------------------------------------------- SeaHasher (host side)
// checksum_hasher() (src/snapshot/mod.rs:318-320) = seahash::SeaHasher::new(), seahash "4.1" (Cargo.toml:24,
// un-vendored): stream hasher restated from the crate's published algorithm.  Used here for the host-resident
// ChecksumParts (resource_checksum.rs:40-44); the per-entity hashing of components runs on the device.
// Rust's `Hash` feeds integers as little-endian bytes (u32 -> 4, u64/usize -> 8), newtype structs hash their
// fields with no prefix, str/String -> bytes then 0xff.
class SeaHasher {
  public:
    void write(const void* data, size_t n) {
        const uint8_t* p = static_cast<const uint8_t*>(data);
        for (size_t i = 0; i < n; ++i) {
            tail_ |= (uint64_t)p[i] << (8 * ntail_);
            if (++ntail_ == 8) { absorb(tail_); tail_ = 0; ntail_ = 0; written_ += 8; }
        }
    }
    void write_u8(uint8_t v) { write(&v, 1); }
    void write_u32(uint32_t v) { uint8_t b[4]; for (int i = 0; i < 4; ++i) b[i] = (uint8_t)(v >> (8 * i)); write(b, 4); }
    void write_u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(v >> (8 * i)); write(b, 8); }
    void write_usize(uint64_t v) { write_u64(v); }
    void write_i32(int32_t v) { write_u32((uint32_t)v); }
    void write_str(const std::string& s) { write(s.data(), s.size()); write_u8(0xff); }
    uint64_t finish() const {
        const uint64_t a = ntail_ ? diffuse(s_[0] ^ tail_) : s_[0];
        return diffuse(a ^ s_[1] ^ s_[2] ^ s_[3] ^ (written_ + ntail_));
    }
    static uint64_t diffuse(uint64_t x) {
        const uint64_t P = 0x6eed0e9da4d94a4fULL;
        x *= P; x ^= (x >> 32) >> (x >> 60); x *= P;
        return x;
    }
  private:
    void absorb(uint64_t word) { const uint64_t a = diffuse(s_[0] ^ word); s_[0] = s_[1]; s_[1] = s_[2]; s_[2] = s_[3]; s_[3] = a; }
    uint64_t s_[4] = {0x16f11fe89b0d677cULL, 0xb480a793d8e6c86cULL, 0x6fe2e5aaf078ebc9ULL, 0x14f994a4c5259381ULL};
    uint64_t tail_ = 0, written_ = 0; unsigned ntail_ = 0;
};

// ---------------------------------------------------------------- GgrsSnapshots (host side)
// src/snapshot/mod.rs:97-274: newest snapshot at the front; `As` is the stored form.  The device ring inside
// libggrs_hip.so follows the same rules over slot indices; this one holds the host-resident resources.
template <class As>
class GgrsSnapshots {
  public:
    GgrsSnapshots& set_depth(size_t depth) { depth_ = depth; return *this; }                    // mod.rs:121-137
    size_t depth() const { return depth_; }
    size_t size() const { return frames_.size(); }
    GgrsSnapshots& push(Frame frame, As snapshot) {                                             // mod.rs:147-181
        while (!frames_.empty()) {
            const Frame cur = frames_.front();
            const uint32_t gap = cur > frame ? (uint32_t)cur - (uint32_t)frame : (uint32_t)frame - (uint32_t)cur;   // i32::abs_diff
            const bool wrapped = gap > UINT32_MAX / 2;
            if ((cur >= frame && !wrapped) || (frame >= cur && wrapped)) { frames_.pop_front(); snaps_.pop_front(); }
            else break;
        }
        frames_.push_front(frame); snaps_.push_front(std::move(snapshot));
        while (snaps_.size() > depth_) { frames_.pop_back(); snaps_.pop_back(); }
        return *this;
    }
    GgrsSnapshots& confirm(Frame confirmed_frame) {                                             // mod.rs:185-202
        while (!frames_.empty() && frames_.back() < confirmed_frame) { frames_.pop_back(); snaps_.pop_back(); }
        return *this;
    }
    GgrsSnapshots& rollback(Frame frame) {                                                      // mod.rs:210-226 (panic -> exception)
        for (;;) {
            if (frames_.empty()) throw std::runtime_error("Could not rollback to " + std::to_string(frame) + ": no snapshot at that moment could be found.");
            if (frames_.front() == frame) return *this;
            frames_.pop_front(); snaps_.pop_front();
        }
    }
    const As& get() const {                                                                     // mod.rs:229-233
        if (snaps_.empty()) throw std::runtime_error("no snapshot available - call rollback(frame) before get()");
        return snaps_.front();
    }
    const As* peek(Frame frame) const {                                                         // mod.rs:236-243
        for (size_t i = 0; i < frames_.size(); ++i) if (frames_[i] == frame) return &snaps_[i];
        return nullptr;
    }
  private:
    std::deque<As> snaps_; std::deque<Frame> frames_; size_t depth_ = DEFAULT_FPS;              // mod.rs:115
};

// ---------------------------------------------------------------- components and systems
// A rollback component on this path is plain-old-data made of words of ONE size: 1, 2, 4 or 8 bytes
// (Transform = 10 x f32, Velocity = 3 x f32, Ttl = 1 x u64, Visibility = 1 x u8).  Specialise for each type:
//   template <> struct HipComponent<Velocity> { static constexpr const char* name = "Velocity";
//       static constexpr uint32_t word_bytes = 4, n_words = 3; };
template <class T> struct HipComponent;

struct KernelSystem {                  // one system of the GgrsSchedule, executed by libggrs_hip.so
    uint32_t kind = 0;
    std::vector<std::string> comps;    // component names, resolved at build time
    uint32_t word[4] = {0, 0, 0, 0};
    int64_t iparam[2] = {0, 0};
    float fparam[4] = {0, 0, 0, 0};
};

// A user-written per-entity system: HIP C++ source defining `__device__ void ggrs_system(GgrsEntity&, const GgrsFrame&)`,
// compiled for gfx950 when it is added (ggrs_hip_add_custom_system).  bind<T>(word) appends the next e.f32(i)/e.u32(i)/e.u64(i).
struct CustomKernelSystem {
    std::string name, source;
    std::vector<std::pair<std::string, uint32_t>> bindings;      // (component name, word)
    int64_t iparam[2] = {0, 0};
    float fparam[4] = {0, 0, 0, 0};
    CustomKernelSystem(std::string n, std::string src) : name(std::move(n)), source(std::move(src)) {}
    template <class T> CustomKernelSystem& bind(uint32_t word) { bindings.emplace_back(HipComponent<T>::name, word); return *this; }
};

// add_systems(GgrsSchedule, <a system that spawns Rollback entities>): `commands.spawn((bundle.., Rollback))` (snapshot/rollback.rs:45-59) as HIP C++ source defining
//   __device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload);
// (ggrs_hip_add_spawn_system).  How many entities a frame spawns and the payload they are built from come from App::set_spawn_payload_source.
struct SpawnKernelSystem {
    std::string name, source;
    std::vector<std::string> bundle;                              // component names every spawned entity gets
    std::vector<std::pair<std::string, uint32_t>> bindings;      // (component name, word) the spawner writes
    uint32_t payload_stride = 0;                                  // bytes of payload per spawned entity; 0: one blob per AdvanceFrame
    int64_t iparam[2] = {0, 0};
    float fparam[4] = {0, 0, 0, 0};
    SpawnKernelSystem(std::string n, std::string src) : name(std::move(n)), source(std::move(src)) {}
    template <class T> SpawnKernelSystem& with() { bundle.emplace_back(HipComponent<T>::name); return *this; }
    template <class T> SpawnKernelSystem& bind(uint32_t word) { bindings.emplace_back(HipComponent<T>::name, word); return *this; }
    SpawnKernelSystem& stride(uint32_t bytes) { payload_stride = bytes; return *this; }
};

namespace systems {
// examples/stress_tests/particles.rs:272-280
template <class TransformT, class VelocityT>
KernelSystem update_particles(float gx, float gy, float gz, uint32_t translation_word = 0, uint32_t velocity_word = 0) {
    KernelSystem s; s.kind = GGRS_SYS_PARTICLES_UPDATE;
    s.comps = {HipComponent<TransformT>::name, HipComponent<VelocityT>::name};
    s.word[0] = translation_word; s.word[1] = velocity_word; s.fparam[0] = gx; s.fparam[1] = gy; s.fparam[2] = gz;
    return s;
}
// particles.rs:282-289
template <class TtlT> KernelSystem despawn_particles(uint32_t word = 0) {
    KernelSystem s; s.kind = GGRS_SYS_TTL_DESPAWN; s.comps = {HipComponent<TtlT>::name}; s.word[0] = word; return s;
}
// particles.rs:254-270
template <class TransformT, class VelocityT, class TtlT> KernelSystem spawn_particles(int64_t ttl, uint8_t input_mask) {
    KernelSystem s; s.kind = GGRS_SYS_PARTICLES_SPAWN;
    s.comps = {HipComponent<TransformT>::name, HipComponent<VelocityT>::name, HipComponent<TtlT>::name};
    s.iparam[0] = ttl; s.iparam[1] = input_mask; return s;
}
// benches/bench.rs:30-46, tests/component_rollback.rs:24-28
template <class T> KernelSystem add_u32(uint32_t delta, uint32_t word = 0) {
    KernelSystem s; s.kind = GGRS_SYS_ADD_U32; s.comps = {HipComponent<T>::name}; s.word[0] = word; s.iparam[0] = delta; return s;
}
// tests/synctest.rs:37-44
template <class T> KernelSystem saturating_sub_despawn(uint32_t amount, uint32_t word = 0) {
    KernelSystem s; s.kind = GGRS_SYS_SAT_SUB_DESPAWN; s.comps = {HipComponent<T>::name}; s.word[0] = word; s.iparam[0] = amount; return s;
}
// examples/box_game/box_game.rs:154-206 (constants :18-22): PlayerT is the plain `Player { handle: usize }`
// component (one 8-byte word), not registered for rollback in the example
template <class TransformT, class VelocityT, class PlayerT>
KernelSystem move_cube_system(float acceleration = 18.0f, float max_speed = 3.0f, float friction = 0.0018f, float plane_size = 5.0f, float cube_size = 0.2f) {
    KernelSystem s; s.kind = GGRS_SYS_BOX_MOVE;
    s.comps = {HipComponent<TransformT>::name, HipComponent<VelocityT>::name, HipComponent<PlayerT>::name};
    s.fparam[0] = acceleration; s.fparam[1] = max_speed; s.fparam[2] = friction; s.fparam[3] = (plane_size - cube_size) * 0.5f;
    return s;
}
}  // namespace systems

// ---------------------------------------------------------------- backend == the C ABI
struct HipBackend {
    ggrs_world* w = nullptr;
    HipBackend(uint64_t capacity, uint32_t max_depth, int device) {
        const int rc = ggrs_hip_world_create(device, capacity, max_depth, &w);
        if (rc != GGRS_OK) throw std::runtime_error(rc == GGRS_E_NO_DEVICE ? "no gfx950 device visible: bevy_ggrs_hip has no CPU fallback" : "ggrs_hip_world_create failed");
    }
    ~HipBackend() { if (w) ggrs_hip_world_destroy(w); }
    HipBackend(const HipBackend&) = delete;
    const char* last_error() { return ggrs_hip_last_error(w); }
    int register_component(const char* n, uint32_t wb, uint32_t nw, uint32_t* id) { return ggrs_hip_register_component(w, n, wb, nw, id); }
    int register_component_ex(const char* n, uint32_t wb, uint32_t nw, uint32_t flags, uint32_t* id) { return ggrs_hip_register_component_ex(w, n, wb, nw, flags, id); }
    int set_component_default(uint32_t c, const void* p) { return ggrs_hip_set_component_default(w, c, p); }
    int checksum_component(uint32_t c, const uint32_t* idx, uint32_t n) { return ggrs_hip_checksum_component(w, c, idx, n); }
    int checksum_component_custom(uint32_t c, const char* source) { return ggrs_hip_checksum_component_custom(w, c, source); }
    int add_system(const ggrs_system_desc* d) { return ggrs_hip_add_system(w, d); }
    int add_custom_system(const ggrs_custom_system_desc* d) { return ggrs_hip_add_custom_system(w, d); }
    int add_spawn_system(const ggrs_spawn_system_desc* d) { return ggrs_hip_add_spawn_system(w, d); }
    int register_component_strategy(uint32_t c, uint32_t stored_word_bytes, uint32_t stored_n_words, const char* source) { return ggrs_hip_register_component_strategy(w, c, stored_word_bytes, stored_n_words, source); }
    int set_frame_rate(uint64_t fps) { return ggrs_hip_set_frame_rate(w, fps); }
    int spawn(uint64_t count, uint64_t mask, const void* const* cols, uint64_t* first) { return ggrs_hip_spawn(w, count, mask, cols, first); }
    int set_depth(uint32_t d) { return ggrs_hip_set_depth(w, d); }
    int set_confirmed(int has, int32_t f) { return ggrs_hip_set_confirmed(w, has, f); }
    int set_synctest_check_distance(int32_t cd) { return ggrs_hip_set_synctest_check_distance(w, cd); }
    int set_input_layout(uint32_t input_bytes, uint32_t max_players) { return ggrs_hip_set_input_layout(w, input_bytes, max_players); }
    int handle_requests(const ggrs_request* r, uint32_t n, uint64_t* out) { return ggrs_hip_handle_requests(w, r, n, out); }
    int enqueue_requests(const ggrs_request* r, uint32_t n) { return ggrs_hip_enqueue_requests(w, r, n, nullptr); }
    int collect_checksums(uint64_t* out, uint32_t max_saves) { return ggrs_hip_collect_checksums(w, out, max_saves, nullptr); }
    bool specialise_wait() { return ggrs_hip_specialise_wait(w) == 1; }     // a loading screen: block until the kernel built for the session's steady tick is in (include/ggrs_hip.h)
    int32_t frame() { return ggrs_hip_frame(w); }
    int set_frame(int32_t f) { return ggrs_hip_set_frame(w, f); }
    uint64_t len() { return ggrs_hip_len(w); }
    int active_count(uint64_t* out) { return ggrs_hip_active_count(w, out); }
    int download_word(uint32_t c, uint32_t word, uint64_t first, uint64_t count, void* dst) { return ggrs_hip_download_word(w, c, word, first, count, dst); }
    int download_alive(uint64_t* dst, uint64_t n) { return ggrs_hip_download_alive(w, dst, n); }
    int has_snapshot(int32_t f) { return ggrs_hip_has_snapshot(w, f); }
    uint64_t snapshot_count() { return ggrs_hip_snapshot_count(w); }
};

// ---------------------------------------------------------------- speculative fan-out (no reference analogue: SURVEY.md 8e)
// RAII over ggrs_fanout: predicted-input branches off the confirmed snapshot, their Checksum(u128)s gathered over RCCL inside the library, branch states kept and
// the matching one ADOPTED when the true inputs arrive.  One object per rank; the caller carries the 128-byte id from rank 0 to the others.
class SpeculativeFanout {
public:
    static std::vector<uint8_t> unique_id() {
        std::vector<uint8_t> id(GGRS_FANOUT_ID_BYTES);
        if (ggrs_hip_fanout_unique_id(id.data()) != GGRS_OK) throw std::runtime_error("ncclGetUniqueId failed (is librccl.so loadable?)");
        return id;
    }
    SpeculativeFanout(HipBackend& be, const std::vector<uint8_t>& id, int rank, int world_size) : w_(be.w) {
        if (id.size() != GGRS_FANOUT_ID_BYTES) throw std::invalid_argument("the fan-out id is GGRS_FANOUT_ID_BYTES bytes");
        if (ggrs_hip_fanout_init(be.w, id.data(), rank, world_size, &f_) != GGRS_OK) throw std::runtime_error(std::string("ggrs_hip_fanout_init: ") + ggrs_hip_last_error(be.w));
    }
    ~SpeculativeFanout() { if (f_) ggrs_hip_fanout_destroy(f_); }
    SpeculativeFanout(const SpeculativeFanout&) = delete;
    void sync_confirmed(int root = 0) { check(ggrs_hip_fanout_sync_confirmed(f_, root)); }
    void set_interval(uint32_t steps_per_all_gather) { check(ggrs_hip_fanout_set_interval(f_, steps_per_all_gather)); }
    // prefix requests, then n_branches x n_frames predicted inputs ([branch][frame][player x input bytes]); flags: GGRS_BRANCH_*; returns the step's SaveGameState count
    uint32_t step_branches(const std::vector<ggrs_request>& prefix, uint32_t n_branches, uint32_t n_frames, uint32_t n_inputs, const std::vector<uint8_t>& inputs,
                           uint32_t flags = 0, const std::vector<ggrs_branch_spawn>& spawn_table = {}, const std::vector<uint16_t>& spawn_sel = {}) {
        ggrs_branch_step st; std::memset(&st, 0, sizeof st);
        st.prefix = prefix.data(); st.n_prefix = (uint32_t)prefix.size(); st.n_branches = n_branches; st.n_frames = n_frames; st.n_inputs = n_inputs; st.flags = flags;
        st.inputs = inputs.data(); st.spawn_table = spawn_table.empty() ? nullptr : spawn_table.data(); st.n_spawn_table = (uint32_t)spawn_table.size();
        st.spawn_sel = spawn_sel.empty() ? nullptr : spawn_sel.data();
        uint32_t ns = 0;
        check(ggrs_hip_fanout_step_branches(f_, &st, &ns));
        return ns;
    }
    // oldest all-gather group: [rank][step][save] Checksum(u128)s
    std::vector<u128> collect(uint32_t* n_steps = nullptr, uint32_t* n_saves = nullptr) {
        int size = 1; check(ggrs_hip_fanout_comm_info(f_, nullptr, &size, nullptr));
        std::vector<uint64_t> raw((size_t)size * 4096 * 2);
        uint32_t steps = 0, saves = 0;
        check(ggrs_hip_fanout_collect(f_, raw.data(), 4096, &steps, &saves));
        std::vector<u128> out((size_t)size * steps * saves);
        for (size_t i = 0; i < out.size(); ++i) { out[i].lo = raw[2 * i]; out[i].hi = raw[2 * i + 1]; }
        if (n_steps) *n_steps = steps;
        if (n_saves) *n_saves = saves;
        return out;
    }
    // the true inputs matched GLOBAL branch `branch` up to `frame`: its retained state becomes the world on every rank (collective).  replay: what a rank that
    // does not own the branch runs instead; returns the Checksum(u128)s of its SaveGameStates (empty on the owner)
    std::vector<u128> adopt(uint32_t branch, Frame frame, const std::vector<ggrs_request>& replay = {}, uint32_t mode = GGRS_ADOPT_RECOMPUTE) {
        uint32_t ns = 0; for (auto& r : replay) ns += r.kind == GGRS_REQ_SAVE;
        std::vector<uint64_t> raw(2 * (size_t)ns + 2); uint32_t got = 0;
        check(ggrs_hip_fanout_adopt(f_, branch, frame, mode, replay.empty() ? nullptr : replay.data(), (uint32_t)replay.size(), raw.data(), &got));
        std::vector<u128> out(got);
        for (uint32_t i = 0; i < got; ++i) { out[i].lo = raw[2 * i]; out[i].hi = raw[2 * i + 1]; }
        return out;
    }
private:
    void check(int rc) { if (rc != GGRS_OK) throw std::runtime_error(std::string("ggrs_hip_fanout: ") + ggrs_hip_fanout_last_error(f_)); }
    ggrs_world* w_ = nullptr; ggrs_fanout* f_ = nullptr;
};

// ---------------------------------------------------------------- GgrsPlugin + App
template <class C> struct GgrsPlugin {};            // src/lib.rs:200-260 (the schedule label argument has no meaning without Bevy)

template <class C, class Backend = HipBackend>
class App {
  public:
    using Input = typename C::Input;
    static_assert(sizeof(Input) == 1, "this path carries one input byte per player (GGRS_MAX_PLAYERS bytes per AdvanceFrame)");
    using ReadInputsSystem = std::function<void(const LocalPlayers&, LocalInputs<C>&)>;
    using SpawnSource = std::function<void(Frame, std::vector<float>& vx, std::vector<float>& vy)>;
    // A GgrsSchedule system that runs on the HOST: it may read RollbackFrameCount / PlayerInputs and insert,
    // mutate or remove *resources* (which live on the host); entity components are device-resident and
    // only kernel-backed systems touch them.  Host systems run in registration order on every AdvanceFrame,
    // after `RollbackFrameCount += 1` (schedule_systems.rs:254-268).
    using HostSystem = std::function<void(App&, const PlayerInputs<C>&)>;

    explicit App(uint64_t capacity, uint32_t max_depth = 16, int device = 0) : be_(capacity, max_depth, device), max_depth_(max_depth) {
        // PlayerInputs<C>: size_of::<C::Input>() plain bytes + one InputStatus byte per player reach the device systems (src/lib.rs:98)
        using Input = typename C::Input;
        static_assert(std::is_trivially_copyable<Input>::value && sizeof(Input) <= GGRS_MAX_INPUT_BYTES, "T::Input reaches the device as its plain bytes (ggrs_hip_set_input_layout)");
        if (sizeof(Input) != 1 && be_.set_input_layout((uint32_t)sizeof(Input), GGRS_MAX_PLAYERS) != GGRS_OK) throw std::runtime_error(be_.last_error());
    }

    // ---- App::add_plugins(GgrsPlugin::<C>::default())
    App& add_plugins(GgrsPlugin<C>) { plugin_ = true; return *this; }
    // ---- app.insert_resource(RollbackFrameRate(FPS)) / insert_resource(Session)
    App& insert_resource(RollbackFrameRate r) { fps_ = r.fps; check(be_.set_frame_rate(r.fps)); return *this; }
    App& insert_resource(SyncTestSession<C> s) {
        session_ = std::move(s);
        auto& ss = std::get<SyncTestSession<C>>(session_);
        // handle_requests, schedule_systems.rs:197-220: MaxPredictionWindow = max_prediction (sync_depth,
        // mod.rs:263-273); SyncTest: ConfirmedFrameCount = frame - check_distance
        max_prediction_window_ = ss.max_prediction();
        if (max_prediction_window_ > max_depth_) throw std::invalid_argument("max_prediction exceeds the ring depth provisioned for this App");
        check(be_.set_depth((uint32_t)max_prediction_window_));
        check(be_.set_synctest_check_distance((int32_t)ss.check_distance()));
        return *this;
    }
    // ---- app.add_systems(ReadInputs, read_local_inputs) / add_systems(GgrsSchedule, system)
    App& add_systems(ReadInputs, ReadInputsSystem f) { read_inputs_ = std::move(f); return *this; }
    App& add_systems(GgrsSchedule, const KernelSystem& s) {
        ggrs_system_desc d; std::memset(&d, 0, sizeof d);
        d.kind = s.kind;
        for (size_t k = 0; k < s.comps.size(); ++k) d.comp[k] = comp_id(s.comps[k]);
        for (int k = 0; k < 4; ++k) { d.word[k] = s.word[k]; d.fparam[k] = s.fparam[k]; }
        d.iparam[0] = s.iparam[0]; d.iparam[1] = s.iparam[1];
        check(be_.add_system(&d));
        has_spawn_system_ |= s.kind == GGRS_SYS_PARTICLES_SPAWN;
        if (s.kind == GGRS_SYS_PARTICLES_SPAWN) spawn_mask_ = (uint8_t)s.iparam[1];
        return *this;
    }
    // add_systems(GgrsSchedule, <user system>): a compile error throws with the hiprtc log
    App& add_systems(GgrsSchedule, const CustomKernelSystem& s) {
        if (s.bindings.size() > GGRS_CUSTOM_MAX_BINDINGS) throw std::invalid_argument("a custom kernel system binds at most 8 words");
        ggrs_custom_system_desc d; std::memset(&d, 0, sizeof d);
        d.name = s.name.c_str(); d.source = s.source.c_str(); d.n_bindings = (uint32_t)s.bindings.size();
        for (size_t k = 0; k < s.bindings.size(); ++k) { d.comp[k] = comp_id(s.bindings[k].first); d.word[k] = s.bindings[k].second; }
        for (int k = 0; k < 4; ++k) d.fparam[k] = s.fparam[k];
        d.iparam[0] = s.iparam[0]; d.iparam[1] = s.iparam[1];
        check(be_.add_custom_system(&d));
        return *this;
    }
    // host-side stand-in for the rolled-back ParticleRng resource (particles.rs:125,201): must be a
    // pure function of the frame, or SyncTest reports a mismatch -- exactly like a non-deterministic system
    App& set_spawn_source(SpawnSource f) { spawn_source_ = std::move(f); return *this; }
    // add_systems(GgrsSchedule, <user-written spawner>): a compile error throws with the hiprtc log
    App& add_systems(GgrsSchedule, const SpawnKernelSystem& s) {
        if (s.bindings.size() > GGRS_CUSTOM_MAX_BINDINGS) throw std::invalid_argument("a spawn system binds at most 8 words");
        ggrs_spawn_system_desc d; std::memset(&d, 0, sizeof d);
        d.name = s.name.c_str(); d.source = s.source.c_str(); d.payload_stride = s.payload_stride; d.n_bindings = (uint32_t)s.bindings.size();
        for (auto& c : s.bundle) d.bundle_mask |= 1ull << comp_id(c);
        for (size_t k = 0; k < s.bindings.size(); ++k) { d.comp[k] = comp_id(s.bindings[k].first); d.word[k] = s.bindings[k].second; }
        for (int k = 0; k < 4; ++k) d.fparam[k] = s.fparam[k];
        d.iparam[0] = s.iparam[0]; d.iparam[1] = s.iparam[1];
        check(be_.add_spawn_system(&d));
        has_custom_spawn_ = true;
        return *this;
    }
    // what the spawner of frame `f` is handed: the number of entities it spawns (0: none this frame) and the payload blob they are built from.  Like the
    // particles' spawn source it must be a pure function of (frame, inputs): a resimulated frame asks again (ParticleRng is a rollback resource, particles.rs:201)
    using SpawnPayloadSource = std::function<uint64_t(Frame, const std::vector<std::pair<Input, InputStatus>>&, std::vector<uint8_t>& payload)>;
    App& set_spawn_payload_source(SpawnPayloadSource f) { spawn_payload_source_ = std::move(f); return *this; }

    App& add_systems(GgrsSchedule, HostSystem f) { host_systems_.push_back(std::move(f)); return *this; }

    // ---- resources (host-resident).  app.insert_resource / init_resource / commands.remove_resource
    template <class R> App& insert_resource(R value) { res_entry<R>().value = std::make_shared<R>(std::move(value)); return *this; }
    template <class R> App& init_resource() { auto& e = res_entry<R>(); if (!e.value) e.value = std::make_shared<R>(); return *this; }
    template <class R> App& remove_resource() { res_entry<R>().value.reset(); return *this; }
    template <class R> R* get_resource() {                                  // Option<Res<R>> / world.get_resource::<R>()
        auto it = resources_.find(std::type_index(typeid(R)));
        return it == resources_.end() ? nullptr : static_cast<R*>(it->second.value.get());
    }
    template <class R> R& resource() {                                      // Res<R>: the reference panics when it is missing
        R* r = get_resource<R>();
        if (!r) throw std::runtime_error(std::string("Requested resource ") + typeid(R).name() + " does not exist");
        return *r;
    }
    // rollback_resource_with_copy / _clone (rollback_app.rs:46-50,64-68 -> ResourceSnapshotPlugin,
    // resource_snapshot.rs:70-98): SaveWorld stores Some(clone) or None, LoadWorld updates / inserts / removes.
    template <class R> App& rollback_resource_with_copy() { return rollback_resource_with_clone<R>(); }
    template <class R> App& rollback_resource_with_clone() {
        auto& e = res_entry<R>();
        e.rollback = true;
        e.clone = [](const void* p) -> std::shared_ptr<void> { return std::make_shared<R>(*static_cast<const R*>(p)); };
        return *this;
    }
    // ReflectStrategy (strategy.rs:86-110) stores reflect_clone() and applies it back: for a host value type that
    // is the clone strategy
    template <class R> App& rollback_resource_with_reflect() { return rollback_resource_with_clone<R>(); }
    // checksum_resource(fn(&R) -> u64) / checksum_resource_with_hash (rollback_app.rs:109-112,124-127 ->
    // ResourceChecksumPlugin, resource_checksum.rs:40-83).  `_with_hash` needs `void ggrs_hash(const R&, SeaHasher&)`
    // (the `#[derive(Hash)]` of the reference) findable by ADL.
    template <class R> App& checksum_resource(uint64_t (*hasher)(const R&)) {
        auto& e = res_entry<R>();
        e.hasher = [hasher](const void* p) { return hasher(*static_cast<const R*>(p)); };
        return *this;
    }
    template <class R> App& checksum_resource_with_hash() {
        auto& e = res_entry<R>();
        e.hasher = [](const void* p) { SeaHasher h; ggrs_hash(*static_cast<const R*>(p), h); return h.finish(); };
        return *this;
    }
    template <class R> App& update_resource_with_map_entities() {
        throw std::logic_error("update_resource_with_map_entities: entity remapping (resource_map.rs) is out of scope - slots are stable, the entity map is the identity");
    }

    // ---- RollbackApp (src/snapshot/rollback_app.rs:31-133)
    // a component that kernel systems read but that is NOT registered for rollback (box_game's `Player`,
    // a mesh handle): device-resident, outside every snapshot, kept across LoadWorld like any non-rollback
    // component of a surviving entity
    template <class T> App& plain_component() {
        uint32_t id = 0;
        check(be_.register_component_ex(HipComponent<T>::name, HipComponent<T>::word_bytes, HipComponent<T>::n_words, GGRS_COMP_NO_ROLLBACK, &id));
        comp_ids_[HipComponent<T>::name] = id;
        return *this;
    }
    template <class T> App& rollback_component_with_copy() { return register_component<T>(); }
    template <class T> App& rollback_component_with_clone() { return register_component<T>(); }   // bitwise for POD (strategy.rs:62-83)
    // rollback_component_with::<S>() for a Strategy whose Stored differs from the component (strategy.rs:22-40): `source` defines ggrs_store / ggrs_load
    // (ggrs_hip_register_component_strategy); the ring then holds stored_n_words words of stored_word_bytes per entity for T
    template <class T> App& rollback_component_with_strategy(uint32_t stored_word_bytes, uint32_t stored_n_words, const std::string& source) {
        register_component<T>();
        check(be_.register_component_strategy(comp_id(HipComponent<T>::name), stored_word_bytes, stored_n_words, source.c_str()));
        return *this;
    }
    // rollback_immutable_component_with_* (rollback_app.rs:40-44,58-62; ImmutableComponentSnapshotPlugin::load,
    // component_snapshot.rs:218-245): LoadWorld re-INSERTS the stored value instead of updating in place so that
    // component hooks fire.  In a SoA column "insert" is "store the words + set the presence bit", which is what
    // the device's LoadWorld does for every component; hooks are host callbacks and do not exist on this path.
    template <class T> App& rollback_immutable_component_with_copy() { return register_component<T>(); }
    template <class T> App& rollback_immutable_component_with_clone() { return register_component<T>(); }
    template <class T> App& rollback_immutable_component_with_reflect() { return rollback_component_with_reflect<T>(); }
    // require_rollback (rollback_app.rs:130-133,241-247): every entity spawned through App::spawn IS a Rollback
    // entity on this path (slot == RollbackOrdered index), so the requirement always holds
    template <class T> App& require_rollback() { comp_id(HipComponent<T>::name); return *this; }
    template <class T> App& update_component_with_map_entities() {
        throw std::logic_error("update_component_with_map_entities: entity remapping (component_map.rs) is out of scope - slots are stable, the entity map is the identity");
    }
    template <class T> App& rollback_component_with_reflect() {
        throw std::logic_error("rollback_component_with_reflect: ReflectStrategy (strategy.rs:86-110) is dynamic reflection, out of scope for the device path");
    }
    template <class T> App& checksum_component_with_hash() {                                    // derive(Hash) over every field
        std::vector<uint32_t> idx(HipComponent<T>::n_words);
        for (uint32_t k = 0; k < idx.size(); ++k) idx[k] = k;
        return checksum_component<T>(idx);
    }
    template <class T> App& checksum_component(const std::vector<uint32_t>& hashed_words) {     // the fn(&T)->u64 of the reference becomes the list of hashed words
        check(be_.checksum_component(comp_id(HipComponent<T>::name), hashed_words.data(), (uint32_t)hashed_words.size()));
        return *this;
    }
    // checksum_component::<T>(fn(&T) -> u64) with an ARBITRARY hasher (rollback_app.rs:119-121): HIP C++ source defining
    // `__device__ ggrs_u64 ggrs_hash(const GgrsComponent& c)` (include/ggrs_hip.h, ggrs_hip_checksum_component_custom)
    template <class T> App& checksum_component_with_source(const std::string& hasher_source) {
        check(be_.checksum_component_custom(comp_id(HipComponent<T>::name), hasher_source.c_str()));
        return *this;
    }
    template <class T> App& set_component_default(const void* words) { check(be_.set_component_default(comp_id(HipComponent<T>::name), words)); return *this; }

    // ---- commands.spawn((bundle.., Rollback)) x count; columns in registration order of `names`, nullptr = default
    uint64_t spawn(uint64_t count, const std::vector<std::string>& names, const std::vector<const void*>& columns = {}) {
        uint64_t mask = 0;
        for (auto& n : names) mask |= 1ULL << comp_id(n);
        uint64_t first = 0;
        check(be_.spawn(count, mask, columns.empty() ? nullptr : columns.data(), &first));
        return first;
    }

    // ---- pipelined mode: handle_requests only ENQUEUES the tick on the device; the checksums are handed
    // to their GameStateCells right before the next advance_frame() (the first moment ggrs looks at
    // them), so the GPU tick overlaps the rest of the host's frame.  Results are identical.
    App& set_pipelined(bool on) { flush(); pipelined_ = on; return *this; }
    void flush() {                         // collect every outstanding batch (oldest first)
        while (!in_flight_.empty()) {
            auto saves = std::move(in_flight_.front()); in_flight_.erase(in_flight_.begin());
            auto parts = std::move(in_flight_parts_.front()); in_flight_parts_.erase(in_flight_parts_.begin());
            std::vector<uint64_t> sums(2 * saves.size() + 2);
            check(be_.collect_checksums(sums.data(), (uint32_t)saves.size()));
            last_checksums_.clear();
            for (size_t k = 0; k < saves.size(); ++k) {
                const u128 cs{sums[2 * k] ^ parts[k], sums[2 * k + 1]};
                saves[k].first->save(saves[k].second, nullptr, cs);
                last_checksums_.push_back(cs);
            }
        }
    }

    // ---- world.remove_resource::<Session<C>>(): the next update() takes the no-session branch
    // (schedule_systems.rs:70-78)
    App& remove_session() { flush(); session_ = std::monostate{}; return *this; }
    // ---- Time<GgrsTime> (src/time.rs:63-87): fully derived from RollbackFrameCount and RollbackFrameRate,
    // elapsed = Duration::from_nanos(frame * 1e9 / fps); a kernel system's dt for frame f is
    // as_secs_f32(elapsed(f) - elapsed(f-1)) and is computed inside libggrs_hip.so (ggrs_request.dt_bits == 0)
    std::chrono::nanoseconds ggrs_time_elapsed() { return std::chrono::nanoseconds((uint64_t)rollback_frame_count() * 1000000000ULL / fps_); }

    // ---- observers / resources
    App& add_observer(std::function<void(const SyncTestMismatch&)> f) { on_mismatch_ = std::move(f); return *this; }
    Frame rollback_frame_count() { return in_requests_ ? req_frame_ : be_.frame(); }                                        // RollbackFrameCount, mod.rs:70
    Frame confirmed_frame_count() const { return confirmed_; }                                  // ConfirmedFrameCount, mod.rs:80
    size_t max_prediction_window() const { return max_prediction_window_; }                     // MaxPredictionWindow, lib.rs:119
    const std::vector<u128>& last_checksums() const { return last_checksums_; }
    Backend& backend() { return be_; }
    uint64_t active_count() { uint64_t n = 0; check(be_.active_count(&n)); return n; }
    uint64_t len() { return be_.len(); }                             // RollbackOrdered::len(): every Rollback entity ever spawned (snapshot/rollback.rs:69-88)
    template <class T, class W> std::vector<W> download(uint32_t word) {
        static_assert(sizeof(W) == HipComponent<T>::word_bytes, "word type must match the component's word size");
        std::vector<W> out(be_.len());
        if (!out.empty()) check(be_.download_word(comp_id(HipComponent<T>::name), word, 0, out.size(), out.data()));
        return out;
    }

    // ---- App::update(): run_ggrs_schedules (src/schedule_systems.rs:19-83) with
    // TimeUpdateStrategy::ManualDuration(delta)
    void update(std::chrono::nanoseconds delta = std::chrono::nanoseconds(1000000000ULL / 60 + 1)) {
        if (!plugin_) throw std::logic_error("GgrsPlugin was not added");
        const uint64_t fps_delta = run_slow_ ? 1000000000ULL * 11 / (fps_ * 10) : 1000000000ULL / fps_;
        accumulator_ += (uint64_t)delta.count();
        while (accumulator_ >= fps_delta) {
            accumulator_ -= fps_delta;
            if (auto* s = std::get_if<SyncTestSession<C>>(&session_)) run_synctest(*s);
            else {   // no session yet: reset time data and counters (schedule_systems.rs:70-78)
                accumulator_ = 0; run_slow_ = false; confirmed_ = -1; max_prediction_window_ = 8;
                check(be_.set_frame(0));
                check(be_.set_confirmed(1, -1));           // world.insert_resource(ConfirmedFrameCount(-1))
                check(be_.set_depth(8));                   // world.insert_resource(MaxPredictionWindow(8))
            }
        }
    }

    // ---- handle_requests (src/schedule_systems.rs:170-289): the whole list is ONE device submission
    void handle_requests(std::vector<GgrsRequest<C>>& requests) {
        std::vector<ggrs_request> reqs(requests.size());
        std::vector<std::vector<uint8_t>> input_bytes; input_bytes.reserve(2 * requests.size());     // per AdvanceFrame: the inputs' bytes, then the status bytes
        using Input = typename C::Input;
        std::vector<std::vector<float>> payload; payload.reserve(2 * requests.size());
        std::vector<std::vector<uint8_t>> blobs; blobs.reserve(requests.size());             // (reserved: the requests keep pointers into the elements)
        std::vector<GgrsRequest<C>*> saves;
        Frame cur = be_.frame();
        const SyncTestSession<C>* st = std::get_if<SyncTestSession<C>>(&session_);
        std::vector<uint64_t> res_parts;           // XOR of the host-resident ChecksumParts, one per SaveGameState
        in_requests_ = true;
        struct Leave { bool& f; ~Leave() { f = false; } } leave{in_requests_};
        for (size_t i = 0; i < requests.size(); ++i) {
            auto& r = requests[i];
            ggrs_request& q = reqs[i]; std::memset(&q, 0, sizeof q);
            q.kind = (uint32_t)r.kind;
            // schedule_systems.rs:197-220: ConfirmedFrameCount is refreshed before EVERY request (SyncTest:
            // current_frame - check_distance when >= 0)
            Frame confirmed_now = confirmed_;
            if (st) { const Frame c = cur - (Frame)st->check_distance(); if (c >= 0) confirmed_now = c; }
            req_frame_ = cur;
            switch (r.kind) {
            case GgrsRequest<C>::SaveGameState:
                q.frame = r.frame; saves.push_back(&r);
                res_parts.push_back(save_resources(cur, confirmed_now));
                break;
            case GgrsRequest<C>::LoadGameState:
                q.frame = r.frame; cur = r.frame; req_frame_ = cur;
                load_resources(cur);
                break;
            case GgrsRequest<C>::AdvanceFrame: {
                // PlayerInputs<T>(Vec<(T::Input, InputStatus)>) (src/lib.rs:98): every player's input bytes, then every player's status
                input_bytes.emplace_back(); input_bytes.emplace_back();
                auto& ibytes = input_bytes[input_bytes.size() - 2]; auto& sbytes = input_bytes[input_bytes.size() - 1];
                bool pressed = false;
                for (auto& in : r.inputs) {
                    uint8_t b[sizeof(Input)]; std::memcpy(b, &in.first, sizeof(Input));
                    ibytes.insert(ibytes.end(), b, b + sizeof(Input)); sbytes.push_back((uint8_t)in.second);
                    pressed |= (b[0] & spawn_mask_) != 0;
                }
                q.inputs = ibytes.data(); q.status = sbytes.data(); q.n_inputs = (uint32_t)r.inputs.size();
                if (has_custom_spawn_ && spawn_payload_source_) {
                    blobs.emplace_back();
                    auto& blob = blobs.back();
                    q.spawn_count = spawn_payload_source_(cur, r.inputs, blob);
                    if (q.spawn_count) { q.spawn_payload = blob.data(); q.spawn_payload_bytes = blob.size(); }
                }
                if (has_spawn_system_ && pressed && spawn_source_) {
                    payload.emplace_back(); payload.emplace_back();
                    auto& vx = payload[payload.size() - 2]; auto& vy = payload[payload.size() - 1];
                    spawn_source_(cur, vx, vy);
                    q.spawn_count = vx.size(); q.spawn_vx = vx.data(); q.spawn_vy = vy.data();
                }
                cur += 1; req_frame_ = cur;
                for (auto& sys : host_systems_) sys(*this, r.inputs);
            } break;
            }
        }
        in_requests_ = false;
        if (pipelined_) {
            if (be_.enqueue_requests(reqs.data(), (uint32_t)reqs.size()) != GGRS_OK) throw std::runtime_error(be_.last_error());
            std::vector<std::pair<GameStateCell*, Frame>> cells;
            for (auto* sv : saves) cells.emplace_back(sv->cell, sv->frame);
            in_flight_.push_back(std::move(cells));
            in_flight_parts_.push_back(std::move(res_parts));
            if (auto* s = std::get_if<SyncTestSession<C>>(&session_)) {
                const Frame c = be_.frame() - (Frame)s->check_distance();
                if (c >= 0) confirmed_ = c;
            }
            return;
        }
        std::vector<uint64_t> sums(2 * saves.size() + 2);
        const int rc = be_.handle_requests(reqs.data(), (uint32_t)reqs.size(), sums.data());
        if (rc != GGRS_OK) throw std::runtime_error(be_.last_error());      // the reference panics here (mod.rs:213-215)
        last_checksums_.clear();
        for (size_t k = 0; k < saves.size(); ++k) {
            const u128 cs{sums[2 * k] ^ res_parts[k], sums[2 * k + 1]};       // checksum.rs:88-99: XOR of ALL parts
            saves[k]->cell->save(saves[k]->frame, nullptr, cs);             // schedule_systems.rs:231-236
            last_checksums_.push_back(cs);
        }
        if (auto* s = std::get_if<SyncTestSession<C>>(&session_)) {         // schedule_systems.rs:204-220
            const Frame c = be_.frame() - (Frame)s->check_distance();
            if (c >= 0) confirmed_ = c;
        }
    }

  private:
    void check(int rc) { if (rc != GGRS_OK) throw std::runtime_error(be_.last_error()); }
    struct ResourceEntry {
        std::shared_ptr<void> value;                                         // null: the resource is absent
        bool rollback = false;
        std::function<std::shared_ptr<void>(const void*)> clone;             // Strategy::store / load (bijection, strategy.rs:21)
        std::function<uint64_t(const void*)> hasher;                         // ResourceChecksumPlugin(fn)
        GgrsSnapshots<std::shared_ptr<void>> snapshots;                      // GgrsResourceSnapshots<R, Option<Stored>>
        bool part_spawned = false; uint64_t part = 0;                        // its ChecksumPart entity (resource_checksum.rs:70-80)
    };
    template <class R> ResourceEntry& res_entry() { return resources_[std::type_index(typeid(R))]; }
    // SaveWorld for the host-resident state: ResourceChecksumPlugin::update (Checksum set), then
    // sync_depth -> discard_old_snapshots -> ResourceSnapshotPlugin::save (Snapshot set).  Returns the XOR of the
    // resource ChecksumParts.  A part, once spawned, stays in the fold even while its resource is absent (the
    // update system needs Res<R> and does not run then; its ChecksumPart entity is never despawned).
    uint64_t save_resources(Frame frame, Frame confirmed) {
        uint64_t fold = 0;
        for (auto& kv : resources_) {
            ResourceEntry& e = kv.second;
            if (e.hasher) {
                if (e.value) { e.part = e.hasher(e.value.get()); e.part_spawned = true; }
                if (e.part_spawned) fold ^= e.part;
            }
            if (!e.rollback) continue;
            e.snapshots.set_depth(max_prediction_window_);
            if (confirmed >= 0) e.snapshots.confirm(confirmed);
            e.snapshots.push(frame, e.value ? e.clone(e.value.get()) : nullptr);
        }
        return fold;
    }
    // LoadWorld (resource_snapshot.rs:81-97): (Some, Some) update, (Some, None) remove, (None, Some) insert
    void load_resources(Frame frame) {
        for (auto& kv : resources_) {
            ResourceEntry& e = kv.second;
            if (!e.rollback) continue;
            const std::shared_ptr<void>& snap = e.snapshots.rollback(frame).get();
            e.value = snap ? e.clone(snap.get()) : nullptr;
        }
    }
    uint32_t comp_id(const std::string& name) const {
        auto it = comp_ids_.find(name);
        if (it == comp_ids_.end()) throw std::invalid_argument("component " + name + " is not registered for rollback");
        return it->second;
    }
    template <class T> App& register_component() {
        uint32_t id = 0;
        check(be_.register_component(HipComponent<T>::name, HipComponent<T>::word_bytes, HipComponent<T>::n_words, &id));
        comp_ids_[HipComponent<T>::name] = id;
        return *this;
    }
    // run_synctest, src/schedule_systems.rs:85-118
    void run_synctest(SyncTestSession<C>& sess) {
        LocalPlayers players;
        for (PlayerHandle h = 0; h < sess.num_players(); ++h) players.handles.push_back(h);
        if (!read_inputs_) throw std::logic_error("No local player inputs found. Did you insert systems into the ReadInputs schedule?");
        LocalInputs<C> local;
        read_inputs_(players, local);
        for (auto& kv : local) sess.add_local_input(kv.first, kv.second);
        flush();                               // cell.save() of the previous tick, before ggrs compares checksums
        try {
            auto requests = sess.advance_frame();
            handle_requests(requests);
        } catch (const GgrsError& e) {
            if (e.kind != GgrsError::MismatchedChecksum) throw;
            if (on_mismatch_) on_mismatch_(SyncTestMismatch{e.current_frame, e.mismatched_frames});   // world.trigger(SyncTestMismatch{..})
        }
    }

    Backend be_;
    uint32_t max_depth_;
    bool plugin_ = false;
    size_t fps_ = DEFAULT_FPS;
    Session<C> session_;
    ReadInputsSystem read_inputs_;
    SpawnSource spawn_source_;
    SpawnPayloadSource spawn_payload_source_; bool has_custom_spawn_ = false;
    bool has_spawn_system_ = false; uint8_t spawn_mask_ = 0;
    std::function<void(const SyncTestMismatch&)> on_mismatch_;
    std::unordered_map<std::string, uint32_t> comp_ids_;
    uint64_t accumulator_ = 0; bool run_slow_ = false;
    Frame confirmed_ = -1;
    size_t max_prediction_window_ = 8;
    std::vector<u128> last_checksums_;
    bool pipelined_ = false;
    std::vector<std::vector<std::pair<GameStateCell*, Frame>>> in_flight_;
    std::vector<std::vector<uint64_t>> in_flight_parts_;
    std::map<std::type_index, ResourceEntry> resources_;
    std::vector<HostSystem> host_systems_;
    bool in_requests_ = false; Frame req_frame_ = 0;
};

}  // namespace bevy_ggrs
