/* ggrs_hip.h -- C ABI of libggrs_hip.so: the MI355X (gfx950) rollback re-simulation engine.
 *
 * This is the drop-in boundary for bevy_ggrs's snapshot-and-resimulate hot path.  The
 * reference (pure Rust, /root/reference) has no FFI of its own; each entry point below
 * names the Rust-level seam it replaces (file:line relative to /root/reference).  A thin
 * Rust shim binds these with `extern "C"` and keeps the reference's
 * names: GgrsPlugin, RollbackApp::rollback_component_with_*, GgrsSchedule, ReadInputs.
 *
 * Conventions
 *   - every function returns int: 0 = GGRS_OK, negative = error; no exceptions, no panics
 *     (where the reference panics -- e.g. rollback to a missing frame, snapshot/mod.rs:213 --
 *     an error code is returned and ggrs_hip_last_error() holds the text);
 *   - plain pointers and sizes only; the library owns all device memory, the caller owns
 *     every host buffer it passes in (it may be freed as soon as the call returns);
 *   - one ggrs_world = one HIP stream; NOT thread-safe (the reference's caller is an
 *     exclusive system holding &mut World, src/lib.rs:252-257);
 *   - no callbacks into the host;
 *   - registered component data lives as SoA *word columns* in HBM: a component is
 *     n_words words of word_bytes (1, 2, 4 or 8) bytes -- a bool / u8 enum is a 1-byte word, an f32 a
 *     4-byte word, a usize an 8-byte word; column w of component c is a dense array indexed by slot.
 *     slot == RollbackOrdered insertion index (snapshot/rollback.rs:69-88): stable, never reused.
 *   - snapshots are complete at every observable point, but a SaveWorld only MOVES the columns whose bytes
 *     in the ring slot differ from the live ones (row versions: every system's write set, every spawn /
 *     upload / insert gives the columns it writes a fresh version; GGRS_ROW_VERSIONS=0 disables it).
 */
#ifndef GGRS_HIP_H
#define GGRS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGRS_HIP_ABI_VERSION 9

/* limits */
#define GGRS_MAX_COMPONENTS 32
#define GGRS_MAX_WORDS      16
#define GGRS_MAX_CKS_UNITS  32
#define GGRS_MAX_SYSTEMS    16
#define GGRS_MAX_PLAYERS    16
#define GGRS_MAX_INPUT_BYTES 16   /* bytes of one player's T::Input (POD) */

/* error codes */
#define GGRS_OK             0
#define GGRS_E_INVALID     -1   /* bad argument / bad call order                                  */
#define GGRS_E_NO_SNAPSHOT -2   /* LoadGameState for a frame not in the ring (mod.rs:213 panic)   */
#define GGRS_E_CAPACITY    -3   /* spawn beyond the world's slot capacity                          */
#define GGRS_E_HIP         -4   /* a HIP runtime call failed; see ggrs_hip_last_error             */
#define GGRS_E_NO_DEVICE   -5   /* no gfx950 device visible: the product path has no CPU fallback */

typedef struct ggrs_world ggrs_world;

/* -------------------------------------------------------------------------------------------
 * World lifetime.  Replaces the Bevy World's archetype storage for registered components plus
 * every GgrsSnapshots<_, _> resource (snapshot/mod.rs:97-119).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t  device;        /* HIP device ordinal                                             */
    uint32_t max_depth;     /* ring slots to provision (>= any depth later set)                */
    uint64_t capacity;      /* max slots (Rollback entities ever spawned)                      */
    void*    stream;        /* hipStream_t to run on, or NULL: the library creates its own     */
    void*    arena;         /* optional caller-provided device memory (e.g. a torch tensor)    */
    uint64_t arena_bytes;   /*   size of that arena; 0 = library calls hipMalloc               */
    uint32_t flags;         /* GGRS_WORLD_* bits                                               */
    uint32_t reserved;
} ggrs_world_desc;

#define GGRS_WORLD_DEFAULT      0u
#define GGRS_WORLD_UNFUSED      2u   /* one kernel per reference system (save/checksum split)  */
#define GGRS_WORLD_NT_COPY      4u   /* snapshot copies use non-temporal loads/stores           */
#define GGRS_WORLD_NO_GROUPS    8u   /* one launch per request: no [Load?](Save|Advance)* fusion  */
#define GGRS_WORLD_LAYOUT_ONLY 16u   /* no device: registration, layout and ggrs_hip_generated_kernel_source only (every
                                        call that would touch the GPU returns GGRS_E_NO_DEVICE) -- a build machine can check
                                        that a schema and its custom systems compile for gfx950 before they are deployed   */

int  ggrs_hip_world_create(int device, uint64_t capacity, uint32_t max_depth, ggrs_world** out);
int  ggrs_hip_world_create_ex(const ggrs_world_desc* desc, ggrs_world** out);
/* bytes of device memory a world of this shape needs (for sizing a caller-provided arena);
 * call after registration with out-of-band numbers: total bytes/slot of all registered words. */
uint64_t ggrs_hip_arena_bytes(uint64_t capacity, uint32_t max_depth, uint32_t n_components,
                              uint32_t bytes_per_slot);
void ggrs_hip_world_destroy(ggrs_world* w);
const char* ggrs_hip_last_error(ggrs_world* w);
int  ggrs_hip_abi_version(void);
int  ggrs_hip_device_count(void);     /* HIP devices this process sees (0: none -- world creation would fail with GGRS_E_NO_DEVICE) */

/* -------------------------------------------------------------------------------------------
 * Registration (build time).  All registration must precede the first spawn/save.
 * ------------------------------------------------------------------------------------------- */

/* RollbackApp::rollback_component_with_copy / _with_clone (snapshot/rollback_app.rs:34-36,52-54;
 * CopyStrategy/CloneStrategy, snapshot/strategy.rs:43-83 -- bitwise for POD). */
int ggrs_hip_register_component(ggrs_world* w, const char* name, uint32_t word_bytes,
                                uint32_t n_words, uint32_t* comp_id);

/* A component that lives on the device next to the rollback components but is NOT registered for
 * rollback ("static collision properties or mesh handles", snapshot/despawn.rs:3-6): it is never
 * snapshotted or restored, it dies with its entity, and an entity that LoadWorld has to re-create
 * (entity.rs:80-90 spawns a fresh one with the old RollbackId) comes back without it.  Such
 * components are the reason RollbackDespawned exists; see ggrs_hip_despawn_rollback below. */
#define GGRS_COMP_ROLLBACK     0u   /* rollback_component_with_copy / _clone                      */
#define GGRS_COMP_NO_ROLLBACK  1u   /* plain device-resident component, outside every snapshot   */
int ggrs_hip_register_component_ex(ggrs_world* w, const char* name, uint32_t word_bytes,
                                   uint32_t n_words, uint32_t flags, uint32_t* comp_id);

/* value given to a freshly spawned entity's component when the spawner passes no data
 * (e.g. Transform::default for `Sprite`-required Transform, particles.rs:262). */
int ggrs_hip_set_component_default(ggrs_world* w, uint32_t comp_id, const void* words);

/* RollbackApp::checksum_component / checksum_component_with_hash (rollback_app.rs:99-101,
 * 119-121; ComponentChecksumPlugin, snapshot/component_checksum.rs:67-108).  The per-entity
 * custom hasher is SeaHash over the listed words, in order, each written as word_bytes
 * little-endian bytes (== derive(Hash) over those fields, or particles.rs:207-222). */
int ggrs_hip_checksum_component(ggrs_world* w, uint32_t comp_id, const uint32_t* word_idx,
                                uint32_t n_idx);
/* RollbackApp::checksum_component::<T>(fn(&T) -> u64) with an ARBITRARY hasher (rollback_app.rs:119-121; the default one is
 * component_checksum.rs:44-48).  `source` is HIP C++ defining
 *
 *     __device__ ggrs_u64 ggrs_hash(const GgrsComponent& c);
 *
 *   c.f32(i) / c.u32(i) / c.i32(i) / c.u64(i) / c.u16(i) / c.u8(i)   word i of the component, c.slot its RollbackOrdered index
 *   GgrsHasher h; h.write_u8/_u16/_u32/_i32/_u64/_usize/_f32_bits(v); h.finish()   == checksum_hasher() (SeaHasher, mod.rs:318-320)
 *
 * e.g. the stress_test's Transform hasher (examples/stress_tests/particles.rs:207-222):
 *     GgrsHasher h; h.write_u32(c.u32(0)); h.write_u32(c.u32(1)); h.write_u32(c.u32(2)); return h.finish();
 * The function is inlined into the request-group kernel the library generates for the world (hiprtc), under the library's
 * floating-point contract; a world with such a hasher needs that kernel (GGRS_E_INVALID at seal without the run-time compiler,
 * or with GGRS_WORLD_NO_GROUPS / GGRS_WORLD_UNFUSED).  A compile error surfaces at seal with the compiler log in
 * ggrs_hip_last_error.  Replaces any word-list spec of the component. */
int ggrs_hip_checksum_component_custom(ggrs_world* w, uint32_t comp_id, const char* source);

/* Kernel-backed systems of the GgrsSchedule (lib.rs:76, 247-251): add_systems(GgrsSchedule, ..).
 * Systems run in registration order; despawns/spawns are deferred to the end of the frame like
 * Bevy Commands (snapshot/set.rs:118-134). */
#define GGRS_SYS_PARTICLES_UPDATE 1u /* particles.rs:272-280  comp[0]=Transform comp[1]=Velocity
                                        word[0]=translation.x word[1]=velocity.x fparam=gravity  */
#define GGRS_SYS_TTL_DESPAWN      2u /* particles.rs:282-289  comp[0]=Ttl(u64) word[0]             */
#define GGRS_SYS_PARTICLES_SPAWN  3u /* particles.rs:254-270  comp[0..2]=Transform,Velocity,Ttl
                                        iparam[0]=ttl iparam[1]=input mask (INPUT_SPAWN)           */
#define GGRS_SYS_ADD_U32          4u /* benches/bench.rs:30-46 comp[0],word[0] += iparam[0]        */
#define GGRS_SYS_SAT_SUB_DESPAWN  5u /* tests/synctest.rs:37-44 saturating_sub(iparam[0]), ==0 despawn;
                                        iparam[1] = GGRS_DESPAWN_*: how the entity is despawned      */
#define GGRS_SYS_BOX_MOVE          6u /* examples/box_game/box_game.rs:154-206 move_cube_system
                                        comp[0]=Transform word[0]=translation.x  comp[1]=Velocity word[1]=velocity.x
                                        comp[2]=Player(handle: usize = one 8-byte word) word[2]
                                        fparam = {ACCELERATION, MAX_SPEED, FRICTION, half_width}.
                                        FRICTION.powf(dt) is evaluated once per frame on the HOST with the
                                        platform libm (what a Linux build of the reference calls) and handed to
                                        the kernel as bits; everything else is IEEE single ops, unfused         */
#define GGRS_DESPAWN_IMMEDIATE 0   /* commands.entity(e).despawn()                                    */
#define GGRS_DESPAWN_ROLLBACK  1   /* commands.entity(e).despawn_rollback()  (snapshot/despawn.rs:114-143) */

typedef struct {
    uint32_t kind;
    uint32_t comp[4];
    uint32_t word[4];
    int64_t  iparam[2];
    float    fparam[4];
} ggrs_system_desc;

int ggrs_hip_add_system(ggrs_world* w, const ggrs_system_desc* desc);

/* A user-written GgrsSchedule system, compiled for gfx950 at run time (hiprtc; libhiprtc is dlopen'ed on first use).
 * The reference lets an app add ANY Bevy system to GgrsSchedule (lib.rs:76, 247-251; e.g. examples/particles/
 * particles.rs:152-160); the kinds above are that set restated as kernels, and this is the open door next to them for a
 * system of the common shape  Query<(&mut A, &mut B, ..), With<Rollback>>  + Commands: it sees ONE entity at a time.
 * `source` is HIP C++ that defines
 *
 *     __device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f);
 *
 *   e.f32(i) / e.u32(i) / e.i32(i) / e.u64(i)   reference to bound word i  (binding i = word `word[i]` of component `comp[i]`;
 *                                               4-byte words as f32/u32/i32, 8-byte words as u64), written back afterwards
 *   e.slot                                      the entity's RollbackOrdered index (snapshot/rollback.rs:69-74)
 *   e.despawn() / e.despawn_rollback()          commands.entity(e).despawn() / .despawn_rollback() (snapshot/despawn.rs:114-143).  A world in which some system's
 *                                               source names despawn_rollback (or the `kill` field) can hold RollbackDespawned markers -- live-only state --: it
 *                                               keeps its live block written every tick and is closed to ggrs_hip_fanout_step_branches; other worlds are not
 *   e.spawn(n)                                  commands.spawn((.., Rollback)) x n, decided HERE, on the device: see GGRS_SPAWN_PAYLOAD_PARENT below
 *   f.dt  f.frame  f.n_inputs                   Time<GgrsTime>::delta_secs (time.rs), the frame being simulated, PlayerInputs::len()
 *   f.input[h]                                  first byte of player h's input (the whole input of a Config<Input = u8> session)
 *   f.input_u8(h) / _u16(h) / _u32(h) / _u64(h) player h's T::Input, little-endian (ggrs_hip_set_input_layout: 1..16 bytes); f.input_ptr(h): its bytes
 *   f.input_status(h)                           GGRS_INPUT_CONFIRMED / _PREDICTED / _DISCONNECTED (PlayerInputs<T>: (T::Input, InputStatus), src/lib.rs:98)
 *   f.fparam[4]  f.iparam[2]                    the desc's constants
 *
 * The system runs for every live entity that has all bound components, in registration order with the other systems;
 * despawns take effect before the next system, as with the built-in kinds.  The code is compiled with -ffp-contract=off
 * and correctly rounded fp32 divide/sqrt: what the source says is what runs, bit for bit, on every rank and every replay
 * -- determinism is the author's contract exactly as it is for a Bevy system (no atomics, no cross-entity reads).
 * The system is inlined into the request-group kernel the library generates for the world (and compiled as a kernel of its own for the
 * one-launch-per-request path).  A compile error returns GGRS_E_INVALID with the compiler log in ggrs_hip_last_error. */
#define GGRS_SYS_CUSTOM 7u
#define GGRS_SYS_SPAWN_CUSTOM 8u   /* a user-written spawn system: ggrs_hip_add_spawn_system below (not a kind for ggrs_hip_add_system) */
#define GGRS_CUSTOM_MAX_BINDINGS 8
typedef struct {
    const char* name;                               /* for error messages and traces; may be NULL               */
    const char* source;                             /* HIP C++ defining ggrs_system (NUL-terminated)            */
    uint32_t n_bindings;
    uint32_t comp[GGRS_CUSTOM_MAX_BINDINGS];
    uint32_t word[GGRS_CUSTOM_MAX_BINDINGS];
    int64_t  iparam[2];
    float    fparam[4];
} ggrs_custom_system_desc;
int ggrs_hip_add_custom_system(ggrs_world* w, const ggrs_custom_system_desc* desc);

/* ComponentSnapshotPlugin<S: Strategy> (snapshot/strategy.rs:22-40, component_snapshot.rs:42-63): what a snapshot HOLDS of a component is
 * S::Stored, produced by S::store and turned back by S::load / S::update -- CopyStrategy / CloneStrategy (Stored == the component, bitwise for
 * POD) are what ggrs_hip_register_component gives; this is the open door next to them: quantised, packed or partial snapshots.
 * `source` is HIP C++ defining BOTH
 *
 *     __device__ void ggrs_store(const GgrsWords& target, GgrsWords& stored);    // Strategy::store(&Target) -> Stored
 *     __device__ void ggrs_load(const GgrsWords& stored, GgrsWords& target);     // Strategy::load (::update defaults to `*target = load(stored)`, strategy.rs:37-39): `target` arrives zeroed
 *
 *   GgrsWords: w.f32(i) / .u32(i) / .i32(i) / .u64(i) / .u16(i) / .u8(i) -- word i of the value (references on the non-const side)
 *
 * The ring slots then hold stored_n_words words of stored_word_bytes per entity for this component instead of its own words (bytes per SaveWorld
 * drop accordingly: ggrs_hip_profile_read_bytes counts the Stored form), the live block holds the component; checksums hash the live component,
 * as the reference's do.  store / load run inside the generated request-group kernel under the library's floating-point contract; a world with
 * such a component needs that kernel (GGRS_E_INVALID at seal without it).  Row versions treat the component as a whole: it is stored
 * when any of its words may have changed. */
int ggrs_hip_register_component_strategy(ggrs_world* w, uint32_t comp_id, uint32_t stored_word_bytes, uint32_t stored_n_words, const char* source);

/* PlayerInputs<T>(Vec<(T::Input, InputStatus)>) (src/lib.rs:98, inserted before every AdvanceWorld: schedule_systems.rs:262-265).
 * input_bytes = size_of::<T::Input>() (POD, 1..16; default 1: Config<Input = u8>), max_players = players of the session (1..16; default 16).
 * Must precede the first ggrs_hip_add_system / _add_custom_system / _add_spawn_system call (GGRS_E_INVALID afterwards).  From then on EVERY AdvanceFrame's
 * `inputs` buffer is read as n_inputs x input_bytes bytes: the library cannot see a buffer's length, so a caller that widens the layout must widen its buffers. */
#define GGRS_INPUT_CONFIRMED    0   /* ggrs::InputStatus::Confirmed    */
#define GGRS_INPUT_PREDICTED    1   /* ggrs::InputStatus::Predicted    */
#define GGRS_INPUT_DISCONNECTED 2   /* ggrs::InputStatus::Disconnected */
int ggrs_hip_set_input_layout(ggrs_world* w, uint32_t input_bytes, uint32_t max_players);

/* A user-written GgrsSchedule system that SPAWNS Rollback entities -- `commands.spawn((.., Rollback))` from any system of the schedule
 * (snapshot/rollback.rs:45-59; examples/stress_tests/particles.rs:254-270 is the built-in GGRS_SYS_PARTICLES_SPAWN).  How many entities a frame
 * spawns is decided on the host, per AdvanceFrame (ggrs_request::spawn_count: the host sees the inputs the system would look at); what they are
 * is `source`, HIP C++ defining
 *
 *     __device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload);
 *
 *   e            the k-th entity of this frame's spawn (0 <= k < spawn_count), every component of the bundle at its registered default
 *                (ggrs_hip_set_component_default); e.f32(i) / e.u32(i) / .. = bound word i, written back afterwards; e.slot = its RollbackOrdered index
 *   payload      the request's spawn_payload blob as the device sees it: + k * payload_stride when payload_stride != 0 (one record per entity),
 *                the whole blob (spawn_payload_bytes) otherwise
 *   f            as for ggrs_system: dt, frame, PlayerInputs, fparam / iparam
 *
 * The new entities are RollbackOrdered's next indices (== the next slots), appended after the frame's other systems ran (Bevy applies Commands
 * at the end of the schedule), inside the request group's launch.  One spawn system per world.  Needs the generated kernel (GGRS_E_INVALID at
 * seal without it). */
/* SPAWNS DECIDED ON THE DEVICE (payload_stride = GGRS_SPAWN_PAYLOAD_PARENT).  In the reference ANY GgrsSchedule system may `commands.spawn((.., Rollback))`, as many as
 * its data says -- a particle that splits, a bullet fired when an entity's own cooldown runs out (rollback.rs:45-59).  A user-written system (ggrs_hip_add_custom_system)
 * asks for it with   e.spawn(n)   (0 <= n <= 255 children of THIS entity in this frame; a later call for the same entity in the same frame replaces it).  After the
 * frame's systems ran -- where Bevy applies Commands -- the children take RollbackOrdered's next indices in the slot order of their parents (an exclusive scan over
 * wave, workgroup and grid inside the group's launch), so every rank and every replay numbers them alike; child k of a parent is built by the world's spawn system:
 *     ggrs_spawn(e, k, f, payload)   with payload = the parent's record: the 8 x ggrs_u64 bound words of the system that called e.spawn(n), as that call left them.
 * ggrs_request::spawn_count and its payload fields are ignored for such a world.  RollbackOrdered::len then lives on the device: ggrs_hip_len and every entry point that
 * needs it waits for the world's stream first; children beyond the world's capacity are dropped and reported (GGRS_E_CAPACITY at the next collect / blocking call).
 * Every launch of such a world covers its whole capacity and is a COOPERATIVE launch (all workgroups resident: grid barriers inside), which bounds the capacity by what
 * the device holds of the world's kernel (GGRS_E_CAPACITY at seal beyond: 256 slots per workgroup x the workgroups resident per CU -- bounded by the kernel's VGPRs and SGPRs, 6 for the
 * splitting-cells world of the tests = 393 216 slots -- x 256 CUs); no depth-parallel roles, no branch steps. */
#define GGRS_SPAWN_PAYLOAD_PARENT 0xFFFFFFFFu
typedef struct {
    const char* name;                               /* for error messages and traces; may be NULL                                  */
    const char* source;                             /* HIP C++ defining ggrs_spawn (NUL-terminated)                                */
    uint64_t bundle_mask;                           /* bit c: the spawned entity has component c                                   */
    uint32_t payload_stride;                        /* bytes of payload per spawned entity; 0: one blob per AdvanceFrame; GGRS_SPAWN_PAYLOAD_PARENT: see above */
    uint32_t n_bindings;                            /* words the spawner writes (all of components in bundle_mask)                 */
    uint32_t comp[GGRS_CUSTOM_MAX_BINDINGS];
    uint32_t word[GGRS_CUSTOM_MAX_BINDINGS];
    int64_t  iparam[2];
    float    fparam[4];
} ggrs_spawn_system_desc;
int ggrs_hip_add_spawn_system(ggrs_world* w, const ggrs_spawn_system_desc* desc);

/* The request-group kernel the library WRITES for a world when it is sealed: one slot per lane, every
 * registered word of the slot in a register, the GgrsSchedule systems -- built-in kinds and custom sources alike -- inlined
 * in registration order, every checksum spec (word lists and custom hashers) unrolled, one 256-slot workgroup per tile (roles, batches,
 * fold-forward); compiled with hiprtc -- or loaded from a shipped code object (`make -C bevy_ggrs_amd/csrc aot`; GGRS_AOT_DIR) when its text
 * hashes to one.  This returns that HIP C++ source (NUL-terminated): *needed = bytes incl. the NUL, min(cap, *needed)
 * bytes are copied.  compile != 0 also builds it for gfx950 (no device needed) and fails with the compiler log in
 * ggrs_hip_last_error if it does not build.  GGRS_E_INVALID: the world is outside what the generator covers (a system that
 * writes a live-only component, more than 64 words per entity).  Registration must be
 * complete; on a GGRS_WORLD_LAYOUT_ONLY world this works without a GPU.  docs/generated/ holds the text of the headline world in both forms. */
#define GGRS_KERNEL_FORM_TILES      1u
#define GGRS_KERNEL_FORM_STEADY     3u   /* the same kernel specialised for the steady SyncTest tick of this world at full length ([Load, Advance, (Save,
                                            Advance) x (max_depth - 1)], the rows its systems write, the store / load / role policies of that size): what
                                            ggrs_hip_specialise_wait waits for and what `make aot` ships, here for inspection / a build-machine compile check */
int ggrs_hip_generated_kernel_source(ggrs_world* w, uint32_t form, char* buf, uint64_t cap, uint64_t* needed, int compile);
/* the file name a shipped code object of `source` carries in the aot directory (scripts/aot_build.py): at least 40 bytes of buf */
int ggrs_hip_aot_object_name(const char* source, char* buf, uint64_t cap);

/* RollbackFrameRate (time.rs:20); default 60 (lib.rs:62). */
int ggrs_hip_set_frame_rate(ggrs_world* w, uint64_t fps);

/* -------------------------------------------------------------------------------------------
 * Entities and host<->device column traffic.
 * ------------------------------------------------------------------------------------------- */

/* commands.spawn((.., Rollback)) x count: Rollback on_add hook -> RollbackOrdered::push
 * (snapshot/rollback.rs:45-59,69-74).  comp_mask bit c = the bundle has component c.
 * cols: for each component in comp_mask (ascending id), n_words host pointers to `count`
 * words each (NULL pointer or NULL cols = component default). */
int ggrs_hip_spawn(ggrs_world* w, uint64_t count, uint64_t comp_mask, const void* const* cols,
                   uint64_t* first_slot);
int ggrs_hip_despawn(ggrs_world* w, uint64_t slot);                       /* commands.entity(e).despawn() */
/* commands.entity(e).despawn_rollback() (snapshot/despawn.rs:114-143): while the current frame is
 * unconfirmed (ConfirmedFrameCount < RollbackFrameCount) the entity is only DISABLED -- marked
 * RollbackDespawned(frame), invisible to every query, snapshot and checksum -- so that LoadWorld can
 * resurrect it with its non-rollback components intact (resurrect_entities, despawn.rs:69-87: marks
 * > the loaded frame are removed) and AdvanceWorld frees it once its frame is confirmed
 * (despawn_confirmed_entities, despawn.rs:89-112: marks <= ConfirmedFrameCount, checked whenever that
 * counter changed).  With the frame already confirmed this is a plain despawn.  The marks are
 * peer-local state: they are not part of any snapshot, checksum or exported state block. */
int ggrs_hip_despawn_rollback(ggrs_world* w, uint64_t slot);
/* RollbackDespawned markers of the live world: bit i of host_dst = slot i is disabled; frames (may be
 * NULL) receives the marked frame of slots [first, first+count) (undefined where not disabled). */
int ggrs_hip_download_disabled(ggrs_world* w, uint64_t* host_dst, uint64_t n_words64);
int ggrs_hip_download_despawned_frames(ggrs_world* w, uint64_t first, uint64_t count, int32_t* frames);
int ggrs_hip_insert_component(ggrs_world* w, uint32_t comp_id, uint64_t slot, const void* words);
int ggrs_hip_remove_component(ggrs_world* w, uint32_t comp_id, uint64_t slot);

int ggrs_hip_upload_word(ggrs_world* w, uint32_t comp_id, uint32_t word, uint64_t first,
                         uint64_t count, const void* host_src);
int ggrs_hip_download_word(ggrs_world* w, uint32_t comp_id, uint32_t word, uint64_t first,
                           uint64_t count, void* host_dst);
int ggrs_hip_download_alive(ggrs_world* w, uint64_t* host_dst, uint64_t n_words64);
int ggrs_hip_download_present(ggrs_world* w, uint32_t comp_id, uint64_t* host_dst, uint64_t n_words64);
/* device address of a live column (for zero-copy interop).  Word columns are stored tile-major: element e
 * of the column lives at dev_ptr + (e / 8192) * tile_stride + (e % 8192) * word_bytes; *tile_stride (may be
 * NULL) is 8192 * word_bytes for a plain array (non-rollback components) and the bytes of all rollback words
 * of 8192 slots otherwise (8192-slot layout tiles).  Whoever holds such a pointer may write the column
 * behind the library's back, so from this call on the column takes no row-version shortcut: every SaveWorld / LoadWorld moves it,
 * and every tick writes the live block (no lazy live block). */
int ggrs_hip_column_device_ptr(ggrs_world* w, uint32_t comp_id, uint32_t word, void** dev_ptr,
                               uint64_t* tile_stride);

uint64_t ggrs_hip_len(ggrs_world* w);            /* RollbackOrdered::len (rollback.rs:91-93)   */
int      ggrs_hip_active_count(ggrs_world* w, uint64_t* out);   /* live Rollback entities    */

/* -------------------------------------------------------------------------------------------
 * Frame counters and the snapshot ring (snapshot/mod.rs:68-86, 121-274).
 * ------------------------------------------------------------------------------------------- */
int32_t ggrs_hip_frame(ggrs_world* w);                           /* RollbackFrameCount       */
int  ggrs_hip_set_frame(ggrs_world* w, int32_t frame);
int  ggrs_hip_set_depth(ggrs_world* w, uint32_t depth);          /* sync_depth, mod.rs:263-273 */
int  ggrs_hip_set_confirmed(ggrs_world* w, int has, int32_t confirmed_frame); /* ConfirmedFrameCount;
                                              applied by discard_old_snapshots before each save */
int  ggrs_hip_has_snapshot(ggrs_world* w, int32_t frame);        /* peek(frame).is_some(): 1/0 */
uint64_t ggrs_hip_snapshot_count(ggrs_world* w);

/* -------------------------------------------------------------------------------------------
 * Request execution (handle_requests, src/schedule_systems.rs:170-289).
 * ------------------------------------------------------------------------------------------- */

/* SaveWorld (snapshot/set.rs:104-107): Checksum systems -> ChecksumPlugin::update fold
 * (snapshot/checksum.rs:88-99) -> Snapshot systems push at RollbackFrameCount
 * (component_snapshot.rs:66-84, entity.rs:39-51).  checksum_out = Checksum(u128) as {lo, hi}. */
int ggrs_hip_save(ggrs_world* w, uint64_t checksum_out[2]);

/* LoadWorld (set.rs:92-103): RollbackFrameCount = frame (schedule_systems.rs:244-247), ring
 * rollback (mod.rs:210-226), entity reconcile (entity.rs:55-99) and component restore
 * (component_snapshot.rs:95-123). */
int ggrs_hip_load(ggrs_world* w, int32_t frame);

/* AdvanceWorld (set.rs:108-134): RollbackFrameCount += 1 (schedule_systems.rs:254-259),
 * GgrsTimePlugin::update (time.rs:63-87; dt_bits==0 -> derived from the frame number and
 * RollbackFrameRate), then the registered GgrsSchedule systems. */
int ggrs_hip_advance(ggrs_world* w, uint32_t dt_bits, const uint8_t* inputs, uint32_t n_inputs,
                     uint64_t spawn_count, const float* spawn_vx, const float* spawn_vy);

#define GGRS_REQ_SAVE    1u   /* GgrsRequest::SaveGameState  (schedule_systems.rs:223-237) */
#define GGRS_REQ_LOAD    2u   /* GgrsRequest::LoadGameState  (schedule_systems.rs:238-250) */
#define GGRS_REQ_ADVANCE 3u   /* GgrsRequest::AdvanceFrame   (schedule_systems.rs:251-268) */

typedef struct {
    uint32_t kind;            /* GGRS_REQ_*                                                    */
    int32_t  frame;           /* SAVE: frame handed to cell.save; LOAD: frame to restore       */
    uint32_t dt_bits;         /* ADVANCE: f32 bits of Time::delta_secs, 0 = derive (time.rs)   */
    uint32_t n_inputs;        /* ADVANCE: PlayerInputs length (players)                        */
    const uint8_t* inputs;    /* ADVANCE: EXACTLY n_inputs x input_bytes bytes are read: T::Input of every player (ggrs_hip_set_input_layout; default 1 byte each) */
    const uint8_t* status;    /* ADVANCE: n_inputs InputStatus bytes (GGRS_INPUT_*), NULL = every input Confirmed                                  */
    uint64_t spawn_count;     /* ADVANCE: entities the world's spawn system appends in this frame (PARTICLES_SPAWN: if its input bit is held)     */
    const float* spawn_vx;    /*   GGRS_SYS_PARTICLES_SPAWN: host arrays of spawn_count f32 (host-side ParticleRng draw)                          */
    const float* spawn_vy;
    const void* spawn_payload;     /* a user-written spawn system's payload (ggrs_hip_add_spawn_system): spawn_count x payload_stride bytes, or  */
    uint64_t spawn_payload_bytes;  /*   one blob of this many bytes when payload_stride == 0                                                       */
} ggrs_request;

/* Executes a whole request list as one device submission (one stream sync at the end).
 * Before each request the session-derived ConfirmedFrameCount rule for SyncTest sessions can be
 * applied by the library (schedule_systems.rs:204-220) when check_distance >= 0 is set through
 * ggrs_hip_set_synctest_check_distance; otherwise the caller sets it via ggrs_hip_set_confirmed.
 * checksums_out receives {lo,hi} per SAVE request, in request order. */
int ggrs_hip_handle_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n,
                             uint64_t* checksums_out);
int ggrs_hip_set_synctest_check_distance(ggrs_world* w, int32_t check_distance /* <0: off */);

/* Asynchronous form of the same call.  enqueue: all host-side bookkeeping (RollbackFrameCount, ring
 * push/confirm/rollback) happens immediately and in request order, the device work is only queued on the
 * world's stream; *n_saves_out = SAVE requests in the list.  collect: blocks until the OLDEST uncollected
 * batch has finished and copies its checksums ({lo,hi} per SAVE, request order).  ggrs reads a
 * SaveGameState cell (schedule_systems.rs:231-236) no earlier than the next advance_frame(), so a shim
 * collects right before that call and the tick overlaps the rest of the host's frame.  At most 16 batches /
 * 8192 checksums (4096 per list) may be outstanding; the synchronous calls above refuse to run while any are. */
int ggrs_hip_enqueue_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out);
int ggrs_hip_collect_checksums(ggrs_world* w, uint64_t* checksums_out, uint32_t max_saves, uint32_t* n_saves_out);
uint32_t ggrs_hip_pending_batches(ggrs_world* w);

/* Blocks until all work submitted on the world's stream has finished. */
int ggrs_hip_synchronize(ggrs_world* w);

/* -------------------------------------------------------------------------------------------
 * Speculative fan-out support (BASELINE config 5): export / import one snapshot's bytes so a
 * confirmed frame can be broadcast rank->rank (RCCL) and adopted without touching the host.
 * ------------------------------------------------------------------------------------------- */
/* bytes of one packed world state (all columns [0,capacity), masks, header) */
uint64_t ggrs_hip_state_bytes(ggrs_world* w);
/* device pointer of the LIVE packed state block (contiguous; valid until destroy).  The block is brought up to date first (a session in
 * steady rollback leaves it unwritten between ticks: DESIGN.md section 3 "lazy live block") and is written by every tick from this call on. */
int ggrs_hip_live_state_ptr(ggrs_world* w, void** dev_ptr);
/* after bytes were written into the live state block by an external producer (collective),
 * re-read its header so host-side bookkeeping (len, frame) matches */
int ggrs_hip_adopt_live_state(ggrs_world* w);

/* Speculative fan-out ACROSS GPUs, one process per GPU (north_star: "RCCL broadcast of the confirmed-frame snapshot
 * and all-gather of per-branch checksums over xGMI").  librccl is dlopen'ed by the library (no link-time
 * dependency); the host's only job is to carry the 128-byte ncclUniqueId from rank 0 to the other ranks (any side
 * channel: the reference has none -- ggrs's UDP sockets, a file, MPI ...).
 *   unique_id        rank 0: ncclGetUniqueId.
 *   init             every rank: ncclCommInitRank on the world's device; the communicator lives until fanout_destroy.
 *   sync_confirmed   ONE ncclBroadcast of `root`'s packed live block (ggrs_hip_state_bytes bytes, in place in HBM) on
 *                    the world's stream; receivers adopt it (len, frame).  Start-up / desync recovery only.
 *   set_interval     steps whose checksums travel in ONE all-gather (default 1, at most 16: a group's steps stay outstanding
 *                    batches of the world until the group is collected).  The reference's stress_test exchanges
 *                    checksums every --desync-detection-interval frames, default 10 (examples/stress_tests/particles.rs:49).
 *   step             ggrs_hip_enqueue_requests of this rank's branch list; behind an event, on a side stream -- the next
 *                    step's kernels are not held up -- its Checksum(u128)s join the current group, and every `interval`-th
 *                    step issues ONE ncclAllGather for the group.  Every rank must pass lists with the same number of
 *                    SaveGameState requests.
 *   collect          oldest uncollected group (a partly filled one is closed first): blocks until its all-gather has
 *                    landed; checksums_out is [world_size][n_steps][n_saves][2] u64 ({lo, hi} per Save, rank-major).
 *                    Collectives pair up by ORDER: every rank must call step the same number of times, with the same group
 *                    boundaries.  Each step therefore carries a tag {frame of its first request, number of saves} behind its
 *                    checksums through the all-gather, and collect fails with GGRS_E_INVALID ("ranks are out of step", naming
 *                    both frames) instead of handing out a table whose rows belong to different frames.  A step that ONE rank
 *                    refuses after the first (a list of another shape, a spawn beyond its capacity ..) still takes that rank through
 *                    the group's all-gather, with a tag that says so and the group closed at once: the call returns its error there,
 *                    and the other ranks' collect fails with "rank r refused step k" instead of waiting for a rank that stopped calling.
 * At most 8 groups may be in flight. */
#define GGRS_FANOUT_ID_BYTES 128
typedef struct ggrs_fanout ggrs_fanout;
int  ggrs_hip_fanout_unique_id(uint8_t id_out[GGRS_FANOUT_ID_BYTES]);
int  ggrs_hip_fanout_init(ggrs_world* w, const uint8_t id[GGRS_FANOUT_ID_BYTES], int rank, int world_size, ggrs_fanout** out);
int  ggrs_hip_fanout_sync_confirmed(ggrs_fanout* f, int root);
int  ggrs_hip_fanout_step(ggrs_fanout* f, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out);
int  ggrs_hip_fanout_set_interval(ggrs_fanout* f, uint32_t steps_per_all_gather);
int  ggrs_hip_fanout_collect(ggrs_fanout* f, uint64_t* checksums_out, uint32_t max_u128_per_rank, uint32_t* n_steps_out, uint32_t* n_saves_out);
void ggrs_hip_fanout_destroy(ggrs_fanout* f);
const char* ggrs_hip_fanout_last_error(ggrs_fanout* f);
/* what the communicator itself says (ncclCommUserRank / ncclCommCount) and the HIP device the world runs on: bench.py prints
 * n_gpus from here, not from the environment. */
int  ggrs_hip_fanout_comm_info(ggrs_fanout* f, int* rank_out, int* size_out, int* device_out);

/* ---- A step in compact form, branch states that are KEPT, and adoption ("each branch ... on a private scratch copy", "when the true input
 * arrives, the matching branch's state is adopted (device-local pointer swap on the owning rank + broadcast if ranks must stay replicated)").
 *
 * ggrs_hip_fanout_step_branches is ggrs_hip_fanout_step for the list
 *     prefix[0 .. n_prefix)                                                      any requests: they run first, through the ordinary path (ring, counters), and
 *                                                                                must leave a snapshot of the frame F the world is then at (end them with SaveGameState)
 *     per branch b:  LoadGameState(F), (AdvanceFrame(inputs[b][i]), SaveGameState(F+1+i)) x n_frames      -- the last SaveGameState only with GGRS_BRANCH_SAVE_LAST
 * without ~ 4 x n_branches x n_frames requests crossing the ABI: the library expands it, and ALL branches ride in ONE launch of the world's generated kernel whatever
 * their inputs and spawns are (one record per branch in device memory).  The branches are speculation, not history: they do not touch the world's ring or its
 * frame counters -- after the call the world is where the prefix left it (frame F, live world and newest snapshot = F), so no "settle" load is needed.
 * Checksum order of the step: the prefix's SaveGameStates, then branch 0's, branch 1's, ...  (what the request list above would have produced).
 *     inputs       [n_branches][n_frames][n_inputs x input_bytes]   predicted PlayerInputs of frame F+i of branch b
 *     status       [n_branches][n_frames][n_inputs] InputStatus bytes, or NULL (every input Confirmed, as for ggrs_request::status)
 *     spawn_sel    [n_branches][n_frames] or NULL: 0 = the world's spawn system does not fire in that frame of that branch, j = it appends spawn_table[j - 1]
 *                  (the host decides, as with ggrs_request::spawn_count; a rolled-back RNG makes the payload a function of the frame, so branches share entries)
 * GGRS_BRANCH_RETAIN_ALL keeps every frame a branch produces -- each SaveGameState's snapshot and, without GGRS_BRANCH_SAVE_LAST, the state after the last AdvanceFrame --
 * in private packed state blocks outside the ring (state_bytes each, allocated on first use, owned by the world); GGRS_BRANCH_RETAIN_NEWEST only the last frame (F + n_frames).
 * Row versions apply: a column no system writes reaches a branch block once.  They stay valid until the next step of this fan-out.
 * Needs the generated kernel and a world without live-only state (RollbackDespawned markers: a system that can call despawn_rollback(), or a host-issued
 * ggrs_hip_despawn_rollback on an unconfirmed frame; GGRS_COMP_NO_ROLLBACK components): GGRS_E_INVALID otherwise.
 *
 * ggrs_hip_fanout_adopt: the true inputs of frames F .. frame-1 have arrived and equal what `branch` (GLOBAL index: rank x n_branches + local index of the LAST step)
 * predicted: that branch's retained state of `frame` becomes the world -- RollbackFrameCount = ConfirmedFrameCount = frame, a snapshot of `frame` in the ring, the live
 * world loaded from it (what LoadGameState leaves behind, schedule_systems.rs:238-250).  Collective: every rank calls it with the same arguments, no step in flight.
 *   on the OWNING rank      the retained block trades places with a ring slot (no bytes move) + one LoadWorld launch
 *   on the other ranks      GGRS_ADOPT_RECOMPUTE (default): `replay` -- the caller's request list that re-simulates F -> frame with the confirmed inputs and ends with
 *                           SaveGameState(frame), typically [AdvanceFrame x (frame - F), SaveGameState(frame)] -- runs through the ordinary path: no bytes cross xGMI, and
 *                           its Checksum(u128)s (checksums_out, {lo, hi} per SaveGameState of replay; *n_checksums_out = 0 on the owner) can be compared with the
 *                           branch's gathered ones: a free desync check.  GGRS_ADOPT_BROADCAST: ONE ncclBroadcast of the owner's packed block (state_bytes) into a ring
 *                           slot of every other rank, for worlds whose re-simulation costs more than the block's trip over one xGMI link (replay is ignored).
 * Errors: GGRS_E_NO_SNAPSHOT when the owner did not keep that frame (GGRS_BRANCH_RETAIN_*).  Only the owner can know: with GGRS_ADOPT_BROADCAST the ranks exchange one
 * status word first, so EVERY rank returns the error and none waits in the broadcast; with GGRS_ADOPT_RECOMPUTE there is no collective inside the call -- the owner
 * returns the error, the other ranks have re-simulated (they hold the right state; the owner's caller re-simulates too). */
typedef struct {
    uint64_t count;                  /* entities the spawn system appends                                                  */
    const float* vx;                 /* GGRS_SYS_PARTICLES_SPAWN: count f32 each (as ggrs_request::spawn_vx / _vy)         */
    const float* vy;
    const void* payload;             /* a user-written spawn system's payload (as ggrs_request::spawn_payload)             */
    uint64_t payload_bytes;
} ggrs_branch_spawn;
#define GGRS_BRANCH_SAVE_LAST      1u
#define GGRS_BRANCH_RETAIN_NEWEST  2u
#define GGRS_BRANCH_RETAIN_ALL     4u
typedef struct {
    const ggrs_request* prefix;
    uint32_t n_prefix;
    uint32_t n_branches;
    uint32_t n_frames;
    uint32_t n_inputs;
    uint32_t flags;                  /* GGRS_BRANCH_* */
    uint32_t n_spawn_table;
    const uint8_t* inputs;
    const uint8_t* status;
    const ggrs_branch_spawn* spawn_table;
    const uint16_t* spawn_sel;
} ggrs_branch_step;
int  ggrs_hip_fanout_step_branches(ggrs_fanout* f, const ggrs_branch_step* step, uint32_t* n_saves_out);
#define GGRS_ADOPT_RECOMPUTE 0u
#define GGRS_ADOPT_BROADCAST 1u
int  ggrs_hip_fanout_adopt(ggrs_fanout* f, uint32_t branch, int32_t frame, uint32_t mode, const ggrs_request* replay, uint32_t n_replay,
                           uint64_t* checksums_out, uint32_t* n_checksums_out);

/* -------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): per-kernel-class HIP-event timing on the world's stream.
 * ------------------------------------------------------------------------------------------- */
#define GGRS_KERNEL_SAVE     0u
#define GGRS_KERNEL_LOAD     1u
#define GGRS_KERNEL_ADVANCE  2u
#define GGRS_KERNEL_CHECKSUM 3u   /* standalone checksum passes and the per-group finalize   */
#define GGRS_KERNEL_TICK     4u   /* fused request group: [Load?] (Save | Advance)* in one pass */
#define GGRS_KERNEL_CLASSES  5u
int ggrs_hip_profile_enable(ggrs_world* w, int on);
/* total milliseconds and launch count per class since enable; sizes GGRS_KERNEL_CLASSES */
int ggrs_hip_profile_read(ggrs_world* w, double* ms_out, uint64_t* launches_out);
/* duration (microseconds) of every launch of one class since enable, in submission order: min(cap, *n_out) values are
 * copied, *n_out = launches recorded (bench.py: first vs last timed launch, clock ramp diagnosis). */
int ggrs_hip_profile_read_launches(ggrs_world* w, uint32_t kernel_class, float* us_out, uint32_t cap, uint32_t* n_out);
/* algorithmic bytes the launches of each class were asked to move since enable: the rows a launch loads and stores (after row
 * versions) x the slots it covers -- the numerator of bench.py's roofline; size GGRS_KERNEL_CLASSES */
int ggrs_hip_profile_read_bytes(ggrs_world* w, uint64_t* bytes_out);

/* Where the HOST spends a tick: microseconds summed over the calls since the timeline was enabled.  enable: 1 = reset and start, 0 = stop, -1 = leave.
 * us_out[GGRS_TIMELINE_FIELDS] = {enqueue calls, of them: request validation, of them: the launch calls into the HIP runtime, collect calls, of them: waiting
 * for the batch's event, of them: waiting for fold-forward tags, of them: hashing / folding on the host}; counts_out[3] = {enqueue calls, collect calls, launches}. */
#define GGRS_TIMELINE_FIELDS 7
int ggrs_hip_host_timeline(ggrs_world* w, int enable, double* us_out, uint64_t* counts_out);

/* -------------------------------------------------------------------------------------------
 * Introspection: which kernel serves this world's request lists right now and why, what kind of arena
 * it lives on, whether the run-time compiler (libhiprtc.so, dlopen'ed) is available.  `key=value` lines,
 * NUL-terminated; *needed = bytes incl. the NUL, min(cap, *needed) are copied.  Keys: sealed, arena,
 * arena_bytes, hiprtc, generated_kernel, generated_kernel_origin, request_group_kernel, checksum_fold, kernarg_bytes, group_caps, specialised_kernel, slots_covered, row_versions.
 * ------------------------------------------------------------------------------------------- */
int ggrs_hip_world_kernel_info(ggrs_world* w, char* buf, uint64_t cap, uint64_t* needed);

/* Once a session has sent a request-group shape GGRS_JIT_SPECIALISE_AFTER (16) times, the library builds -- on a worker thread, never
 * on the caller's, one build at a time -- a copy of the world's generated kernel with that shape's op sequence and row masks as literals
 * (12 % faster at 1 M entities, 15-20 % at 10 k - 300 k) and switches to it when it is ready; a shape without a kernel runs on the
 * general one.  Shapes are counted one by one, 16 per world (least recently used out): a SyncTest session has one steady shape, a P2P
 * session one per rollback length (BASELINE config 4 at 100 k: 8 kernels, the tick's kernel time 8.9 -> 6.9 us, profiles/r03zi).
 * ggrs_hip_specialise_wait blocks until the build in flight has finished (a loading screen, a benchmark; shapes still waiting for their
 * turn start theirs with their next group): 1 = at least one specialised kernel is ready, 0 = none (no shape has come up often enough,
 * the build failed -- ggrs_hip_world_kernel_info "specialised_kernel" says which and how many). */
int ggrs_hip_specialise_wait(ggrs_world* w);

#ifdef __cplusplus
}
#endif
#endif /* GGRS_HIP_H */
