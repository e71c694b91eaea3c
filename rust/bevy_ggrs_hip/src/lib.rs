//! `bevy_ggrs_hip` -- drives `libggrs_hip.so` (MI355X / gfx950) from a Bevy app, keeping bevy_ggrs's names.
//!
//! UN-BUILT SOURCE.  The build image has neither a Rust toolchain nor vendored `bevy` / `ggrs`; this crate is the
//! shim a bevy_ggrs user links instead of `bevy_ggrs`.  What is built and tested in this
//! repository is the C ABI it binds (`include/ggrs_hip.h`, mirrored 1:1 in `ffi.rs`; `tests/test_abi.py` checks that
//! every `ffi::` symbol used below is declared there and in the header) and a C++ twin of this file
//! (`include/bevy_ggrs_hip.hpp`) that runs the reference's integration tests on the GPU.
//!
//! A user switches by replacing `use bevy_ggrs::prelude::*` with `use bevy_ggrs_hip::prelude::*`:
//!
//! ```ignore
//! App::new()
//!     .insert_resource(HipWorldConfig { capacity: 1 << 20, max_depth: 9, device: 0 })
//!     .add_plugins(GgrsPlugin::<GgrsConfig<u8>>::default())            // same name; OWNS the session-driving system
//!     .insert_resource(RollbackFrameRate(60))
//!     .add_systems(ReadInputs, read_local_inputs)
//!     .rollback_component_with_copy::<Velocity>()                     // RollbackApp, same method names
//!     .rollback_component_with_clone::<Transform>()
//!     .checksum_component_with_hash::<Velocity>()
//!     .mirror_component::<Transform>()                                // device -> Bevy copy for rendering
//!     .add_kernel_system(GgrsSchedule, systems::update_particles::<Transform, Velocity>(Vec3::NEG_Y * 200.0))
//!     .insert_resource(Session::SyncTest(session));
//! ```
//!
//! What this crate owns (paths relative to the bevy_ggrs repository):
//! * the session-driving system: the accumulator, the per-session-type step and the no-session reset of
//!   `run_ggrs_schedules` / `run_synctest` / `run_p2p` / `run_spectator` (src/schedule_systems.rs:19-168) -- [`drive_session`];
//! * `handle_requests` (src/schedule_systems.rs:170-289) -- [`handle_requests`]: ONE device submission per request list;
//! * `RollbackApp::{rollback,checksum}_component_*` (src/snapshot/rollback_app.rs:31-133) -- [`RollbackApp`];
//! * the `Rollback` marker and its `on_add` hook (src/snapshot/rollback.rs:45-59): the hook assigns the device slot,
//!   which IS the `RollbackOrdered` index -- [`Rollback`], [`HipSlot`], [`PendingSpawns`];
//! * `GgrsSnapshots<_, _>`, `ComponentSnapshotPlugin::{save,load}`, the checksum plugins: inside the library.
//! Unchanged bevy_ggrs items are re-exported (`ReadInputs`, `LocalInputs`, `PlayerInputs`, `Session`, `SyncTestMismatch`,
//! `RollbackFrameRate`, the frame-count resources, `GgrsTime`); resource snapshots, reflect and hierarchy strategies
//! are host-side features of bevy_ggrs this shim does not touch.

pub mod ffi;

use bevy::ecs::component::HookContext;
use bevy::ecs::world::DeferredWorld;
use bevy::prelude::*;
use bevy_ggrs::{
    ConfirmedFrameCount, GgrsSchedule, LocalInputs, LocalPlayers, MaxPredictionWindow, PlayerInputs, ReadInputs, RollbackFrameCount,
    RollbackFrameRate, Session, SyncTestMismatch,
};
use core::time::Duration;
use ggrs::{Config, GgrsError, GgrsRequest, InputStatus, SessionState};
use std::ffi::{CStr, CString};
use std::marker::PhantomData;

pub mod prelude {
    pub use crate::{
        hip_component, systems, DeviceInput, GgrsPlugin, HipComponent, HipSlot, HipWorld, HipWorldConfig, KernelSystem, Rollback, RollbackApp, SpawnBlob, SpawnKernelSystem, SpawnPayload,
    };
    pub use bevy_ggrs::prelude::{
        GgrsConfig, GgrsSchedule, GgrsTime, LocalInputs, LocalPlayers, PlayerInputs, ReadInputs, RollbackFrameRate, Session, SyncTestMismatch,
    };
    pub use ggrs::{GgrsEvent, PlayerType, SessionBuilder};
}

// ------------------------------------------------------------------------------------------------ device world

/// Shape of the device world; insert before `GgrsPlugin`.
#[derive(Resource, Clone, Copy)]
pub struct HipWorldConfig {
    pub capacity: u64,
    pub max_depth: u32,
    pub device: i32,
}

/// The device world: replaces the archetype columns of every registered component and every `GgrsSnapshots<_, _>`
/// resource (src/snapshot/mod.rs:97-119).  One world == one HIP stream; same exclusive-system contract as the
/// reference (src/lib.rs:252-257), hence the manual `Send + Sync`.
#[derive(Resource)]
pub struct HipWorld {
    raw: *mut ffi::ggrs_world,
    comp_ids: bevy::platform::collections::HashMap<core::any::TypeId, u32>,
    uploaders: Vec<fn(&mut World)>,    // one per registered component: stage its columns for pending spawns
    in_flight: Option<Vec<(ggrs::GameStateCell<u128>, i32)>>,   // cells of the batch enqueued last tick (async mode)
}
unsafe impl Send for HipWorld {}
unsafe impl Sync for HipWorld {}

impl HipWorld {
    pub fn new(cfg: HipWorldConfig) -> Self {
        let mut raw = core::ptr::null_mut();
        let rc = unsafe { ffi::ggrs_hip_world_create(cfg.device, cfg.capacity, cfg.max_depth, &mut raw) };
        assert!(rc != ffi::GGRS_E_NO_DEVICE, "no gfx950 device visible: bevy_ggrs_hip has no CPU fallback");
        assert_eq!(rc, ffi::GGRS_OK, "ggrs_hip_world_create failed");
        assert_eq!(unsafe { ffi::ggrs_hip_abi_version() }, ffi::GGRS_HIP_ABI_VERSION, "libggrs_hip.so ABI mismatch");
        Self { raw, comp_ids: default(), uploaders: Vec::new(), in_flight: None }
    }
    /// Error codes become the panics the reference raises on the same paths (e.g. src/snapshot/mod.rs:213-215).
    fn check(&self, rc: i32) {
        if rc != ffi::GGRS_OK {
            panic!("{}", unsafe { CStr::from_ptr(ffi::ggrs_hip_last_error(self.raw)) }.to_string_lossy());
        }
    }
    pub fn comp_id<T: HipComponent>(&self) -> u32 {
        *self.comp_ids.get(&core::any::TypeId::of::<T>()).expect("component is not registered for rollback")
    }
    pub fn len(&self) -> u64 {
        unsafe { ffi::ggrs_hip_len(self.raw) }
    }
    /// After 16 identical request groups the library builds, on a worker thread, a copy of the world's kernel specialised for that
    /// shape (include/ggrs_hip.h `ggrs_hip_specialise_wait`).  A loading screen that has driven a few warm-up ticks can block here
    /// until it is in; true = the steady tick now runs on its own kernel.
    pub fn specialise_wait(&self) -> bool {
        unsafe { ffi::ggrs_hip_specialise_wait(self.raw) == 1 }
    }
    pub fn frame(&self) -> i32 {
        unsafe { ffi::ggrs_hip_frame(self.raw) }
    }
    /// Word `word` of every slot `[0, len)` of component `T` (device -> host).
    pub fn download_word<T: HipComponent, W: Copy + Default>(&self, word: u32) -> Vec<W> {
        assert_eq!(core::mem::size_of::<W>() as u32, T::WORD_BYTES);
        let n = self.len();
        let mut out = vec![W::default(); n as usize];
        if n > 0 {
            self.check(unsafe { ffi::ggrs_hip_download_word(self.raw, self.comp_id::<T>(), word, 0, n, out.as_mut_ptr().cast()) });
        }
        out
    }
    pub fn alive_mask(&self) -> Vec<u64> {
        let mut m = vec![0u64; (self.len() as usize + 63) / 64];
        self.check(unsafe { ffi::ggrs_hip_download_alive(self.raw, m.as_mut_ptr(), m.len() as u64) });
        m
    }
}
impl Drop for HipWorld {
    fn drop(&mut self) {
        unsafe { ffi::ggrs_hip_world_destroy(self.raw) }
    }
}

/// Speculative fan-out across the GPUs of a node (no bevy_ggrs analogue): predicted-input branches off the confirmed snapshot, one
/// process per GPU, RCCL inside `libggrs_hip.so`.  `step_branches` hands the library ONE compact description of a step -- a prefix request list and
/// `n_branches x n_frames` predicted inputs -- which it runs as one launch; with `GGRS_BRANCH_RETAIN_*` every branch's frames are kept in private
/// state blocks, and when the true inputs arrive `adopt` makes the matching branch's state the world: a ring-slot swap on the rank that ran the
/// branch, a re-simulation with the confirmed inputs (`replay`) or one broadcast on the others.  The C++ twin (`bevy_ggrs::SpeculativeFanout`) is what
/// tests/cpp/host_test.cpp runs.
pub struct SpeculativeFanout {
    raw: *mut ffi::ggrs_fanout,
    size: i32,
}
unsafe impl Send for SpeculativeFanout {}

impl SpeculativeFanout {
    /// Rank 0 creates the 128-byte id and carries it to the other ranks (any side channel).
    pub fn unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        assert_eq!(unsafe { ffi::ggrs_hip_fanout_unique_id(id.as_mut_ptr()) }, ffi::GGRS_OK, "ncclGetUniqueId failed (is librccl.so loadable?)");
        id
    }
    pub fn new(world: &HipWorld, id: &[u8; 128], rank: i32, world_size: i32) -> Self {
        let mut raw = core::ptr::null_mut();
        world.check(unsafe { ffi::ggrs_hip_fanout_init(world.raw, id.as_ptr(), rank, world_size, &mut raw) });
        Self { raw, size: world_size }
    }
    fn check(&self, rc: i32) {
        if rc != ffi::GGRS_OK {
            panic!("{}", unsafe { CStr::from_ptr(ffi::ggrs_hip_fanout_last_error(self.raw)) }.to_string_lossy());
        }
    }
    pub fn sync_confirmed(&self, root: i32) {
        self.check(unsafe { ffi::ggrs_hip_fanout_sync_confirmed(self.raw, root) });
    }
    pub fn set_interval(&self, steps_per_all_gather: u32) {
        self.check(unsafe { ffi::ggrs_hip_fanout_set_interval(self.raw, steps_per_all_gather) });
    }
    /// `inputs`: `[branch][frame][player x input bytes]`; returns the step's SaveGameState count.
    pub fn step_branches(&self, prefix: &[ffi::ggrs_request], n_branches: u32, n_frames: u32, n_inputs: u32, inputs: &[u8], flags: u32,
                         spawn_table: &[ffi::ggrs_branch_spawn], spawn_sel: &[u16]) -> u32 {
        let st = ffi::ggrs_branch_step {
            prefix: prefix.as_ptr(), n_prefix: prefix.len() as u32, n_branches, n_frames, n_inputs, flags, n_spawn_table: spawn_table.len() as u32,
            inputs: inputs.as_ptr(), status: core::ptr::null(),
            spawn_table: if spawn_table.is_empty() { core::ptr::null() } else { spawn_table.as_ptr() },
            spawn_sel: if spawn_sel.is_empty() { core::ptr::null() } else { spawn_sel.as_ptr() },
        };
        let mut ns = 0u32;
        self.check(unsafe { ffi::ggrs_hip_fanout_step_branches(self.raw, &st, &mut ns) });
        ns
    }
    /// Oldest all-gather group: `(steps, saves, [rank][step][save] Checksum(u128))`.
    pub fn collect(&self) -> (u32, u32, Vec<u128>) {
        let mut raw = vec![0u64; self.size as usize * 4096 * 2];
        let (mut steps, mut saves) = (0u32, 0u32);
        self.check(unsafe { ffi::ggrs_hip_fanout_collect(self.raw, raw.as_mut_ptr(), 4096, &mut steps, &mut saves) });
        let n = self.size as usize * steps as usize * saves as usize;
        (steps, saves, (0..n).map(|i| (raw[2 * i] as u128) | ((raw[2 * i + 1] as u128) << 64)).collect())
    }
    /// Collective: GLOBAL branch `branch`'s retained state of `frame` becomes the world.  Returns the Checksum(u128)s of `replay`'s SaveGameStates
    /// on the ranks that re-simulated (empty on the owner): compare them with the branch's gathered ones.
    pub fn adopt(&self, branch: u32, frame: i32, replay: &[ffi::ggrs_request], broadcast: bool) -> Vec<u128> {
        let ns = replay.iter().filter(|r| r.kind == ffi::GGRS_REQ_SAVE).count();
        let mut raw = vec![0u64; 2 * ns + 2];
        let mut got = 0u32;
        let mode = if broadcast { ffi::GGRS_ADOPT_BROADCAST } else { ffi::GGRS_ADOPT_RECOMPUTE };
        self.check(unsafe { ffi::ggrs_hip_fanout_adopt(self.raw, branch, frame, mode, if replay.is_empty() { core::ptr::null() } else { replay.as_ptr() }, replay.len() as u32, raw.as_mut_ptr(), &mut got) });
        (0..got as usize).map(|i| (raw[2 * i] as u128) | ((raw[2 * i + 1] as u128) << 64)).collect()
    }
}
impl Drop for SpeculativeFanout {
    fn drop(&mut self) {
        unsafe { ffi::ggrs_hip_fanout_destroy(self.raw) }
    }
}

/// A plain-old-data component whose fields are 4- or 8-byte words, stored as one SoA column per word
/// (Transform = 10 x f32, Velocity = 3 x f32, Ttl = 1 x u64).  `unsafe`: the layout claim must hold.
/// Implement it with [`hip_component!`].
pub unsafe trait HipComponent: Component + Copy {
    const NAME: &'static str;
    const WORD_BYTES: u32;
    const N_WORDS: u32;
    /// The component as `N_WORDS` little-endian words (what the column upload sends).
    fn to_words(&self, out: &mut [u64]);
    /// The inverse (what the mirror system writes back into the Bevy component).
    fn from_words(words: &[u64]) -> Self;
}

/// Derive-style helper: `hip_component!(Velocity, f32, [x, y, z]);` / `hip_component!(Ttl, u64, [0]);` /
/// `hip_component!(InheritedVisibility, u8, [0]);` -- the word type may be 1, 2, 4 or 8 bytes wide (the library stores a `bool` /
/// u8 enum as a 1-byte word and hashes it as `derive(Hash)` does: one byte).  Expands to the `unsafe impl HipComponent` with
/// `to_words` / `from_words` over the listed fields, in order.
#[macro_export]
macro_rules! hip_component {
    ($ty:ty, $word:ty, [$($field:tt),+ $(,)?]) => {
        unsafe impl $crate::HipComponent for $ty {
            const NAME: &'static str = stringify!($ty);
            const WORD_BYTES: u32 = core::mem::size_of::<$word>() as u32;
            const N_WORDS: u32 = [$(stringify!($field)),+].len() as u32;
            fn to_words(&self, out: &mut [u64]) {
                let mut k = 0;
                $( out[k] = $crate::word_bits::<$word>(self.$field); k += 1; )+
                let _ = k;
            }
            fn from_words(words: &[u64]) -> Self {
                let mut v: Self = unsafe { core::mem::zeroed() };
                let mut k = 0;
                $( v.$field = $crate::word_from_bits::<$word>(words[k]); k += 1; )+
                let _ = k;
                v
            }
        }
    };
}
#[doc(hidden)]
pub fn word_bits<W: Copy>(w: W) -> u64 {
    let mut b = 0u64;
    unsafe { core::ptr::copy_nonoverlapping(&w as *const W as *const u8, &mut b as *mut u64 as *mut u8, core::mem::size_of::<W>()) };
    b
}
#[doc(hidden)]
pub fn word_from_bits<W: Copy>(b: u64) -> W {
    unsafe { core::ptr::read(&b as *const u64 as *const W) }
}

// ------------------------------------------------------------------------------------------------ Rollback marker

/// Same name and role as `bevy_ggrs::Rollback` (src/snapshot/rollback.rs:22-28).  Its `on_add` hook does what
/// `on_rollback_added` does (rollback.rs:45-59): the entity gets its stable identity -- here the DEVICE SLOT, which is
/// the `RollbackOrdered` insertion index (rollback.rs:69-88) -- and is queued for upload.
#[derive(Component, Default, Clone, Copy)]
#[component(on_add = rollback_added)]
pub struct Rollback;

/// Device slot of a rollback entity: `RollbackId` and `RollbackOrdered::order` in one number (stable, never reused).
#[derive(Component, Clone, Copy, PartialEq, Eq, Hash, Debug)]
#[component(immutable)]
pub struct HipSlot(pub u64);

/// Entities whose `Rollback` was added since the last upload, in hook order (== slot order).
#[derive(Resource, Default)]
pub struct PendingSpawns {
    entities: Vec<Entity>,
}

fn rollback_added(mut world: DeferredWorld, ctx: HookContext) {
    if world.get::<HipSlot>(ctx.entity).is_some() {
        return; // re-inserted marker on an entity that already has its identity
    }
    let staged = world.resource::<PendingSpawns>().entities.len() as u64;
    let slot = world.resource::<HipWorld>().len() + staged;
    world.commands().entity(ctx.entity).insert(HipSlot(slot));
    world.resource_mut::<PendingSpawns>().entities.push(ctx.entity);
}

/// Uploads every pending spawn as ONE `ggrs_hip_spawn` call: SoA columns gathered from the Bevy components of the
/// queued entities (a component missing on an entity clears its presence bit -- the bundle did not have it).
/// Runs at the top of [`drive_session`], i.e. before the tick's requests; entities spawned by host code between two
/// ticks therefore exist in the device world from the next `SaveGameState` on, exactly as `commands.spawn` does in
/// the reference.  (Spawns INSIDE the GgrsSchedule are kernel systems: [`systems::spawn_particles`].)
fn upload_pending_spawns(world: &mut World) {
    let n = world.resource::<PendingSpawns>().entities.len();
    if n == 0 {
        return;
    }
    let uploaders = world.resource::<HipWorld>().uploaders.clone();
    world.insert_resource(SpawnStaging::default());
    for up in uploaders {
        up(world); // fills SpawnStaging for its component
    }
    let staging = world.remove_resource::<SpawnStaging>().unwrap();
    // Entities of one batch may carry different bundles.  ggrs_hip_spawn takes one component mask per call, so the
    // queue is split into runs of equal masks (slot order is preserved: runs are uploaded front to back).
    let hip = world.resource::<HipWorld>();
    let mut start = 0usize;
    while start < n {
        let mask = staging.mask[start];
        let mut end = start + 1;
        while end < n && staging.mask[end] == mask {
            end += 1;
        }
        let mut cols: Vec<*const core::ffi::c_void> = Vec::new();
        let mut keep: Vec<Vec<u8>> = Vec::new();
        for col in staging.columns.iter().filter(|c| (mask >> c.comp) & 1 == 1) {
            let wb = col.word_bytes as usize;
            keep.push(col.bytes[start * wb..end * wb].to_vec());
            cols.push(keep.last().unwrap().as_ptr().cast());
        }
        let mut first = 0u64;
        hip.check(unsafe { ffi::ggrs_hip_spawn(hip.raw, (end - start) as u64, mask, if cols.is_empty() { core::ptr::null() } else { cols.as_ptr() }, &mut first) });
        debug_assert_eq!(first, staging.first_slot + start as u64, "device slots and HipSlot assignments diverged");
        start = end;
    }
    world.resource_mut::<PendingSpawns>().entities.clear();
}

#[derive(Resource, Default)]
struct SpawnStaging {
    first_slot: u64,
    mask: Vec<u64>,              // per queued entity: bit c = it has registered component c
    columns: Vec<StagedColumn>,  // ascending (component id, word)
}
struct StagedColumn {
    comp: u32,
    word_bytes: u32,
    bytes: Vec<u8>,
}

fn stage_component<T: HipComponent>(world: &mut World) {
    let entities = world.resource::<PendingSpawns>().entities.clone();
    let comp = world.resource::<HipWorld>().comp_id::<T>();
    let first_slot = entities.first().and_then(|e| world.get::<HipSlot>(*e)).map(|s| s.0).unwrap_or(0);
    let wb = T::WORD_BYTES as usize;
    let mut cols: Vec<Vec<u8>> = (0..T::N_WORDS).map(|_| vec![0u8; entities.len() * wb]).collect();
    let mut has = vec![false; entities.len()];
    let mut words = vec![0u64; T::N_WORDS as usize];
    for (i, e) in entities.iter().enumerate() {
        if let Some(c) = world.get::<T>(*e) {
            has[i] = true;
            c.to_words(&mut words);
            for (k, w) in words.iter().enumerate() {
                cols[k][i * wb..(i + 1) * wb].copy_from_slice(&w.to_le_bytes()[..wb]);
            }
        }
    }
    let mut st = world.resource_mut::<SpawnStaging>();
    st.first_slot = first_slot;
    if st.mask.is_empty() {
        st.mask = vec![0; entities.len()];
    }
    for (i, h) in has.iter().enumerate() {
        if *h {
            st.mask[i] |= 1 << comp;
        }
    }
    for (k, bytes) in cols.into_iter().enumerate() {
        let _ = k;
        st.columns.push(StagedColumn { comp, word_bytes: T::WORD_BYTES, bytes });
    }
    st.columns.sort_by_key(|c| c.comp); // stable: words of a component stay in order
}

/// Device -> Bevy copy of one component for host-side readers (rendering): every live slot's words are written back
/// into the entity that owns the slot.  Registered with `.mirror_component::<T>()`, runs after [`drive_session`].
fn mirror_component_system<T: HipComponent>(hip: Res<HipWorld>, mut q: Query<(&HipSlot, &mut T)>) {
    let wb = T::WORD_BYTES as usize;
    let cols: Vec<Vec<u8>> = (0..T::N_WORDS)
        .map(|k| {
            if wb == 4 {
                hip.download_word::<T, u32>(k).into_iter().flat_map(|w| w.to_le_bytes()).collect()
            } else {
                hip.download_word::<T, u64>(k).into_iter().flat_map(|w| w.to_le_bytes()).collect()
            }
        })
        .collect();
    let alive = hip.alive_mask();
    let mut words = vec![0u64; T::N_WORDS as usize];
    for (slot, mut c) in &mut q {
        let s = slot.0 as usize;
        if s >= hip.len() as usize || (alive[s / 64] >> (s % 64)) & 1 == 0 {
            continue; // despawned on the device: the despawn mirror removes the entity
        }
        for k in 0..T::N_WORDS as usize {
            let mut b = [0u8; 8];
            b[..wb].copy_from_slice(&cols[k][s * wb..(s + 1) * wb]);
            words[k] = u64::from_le_bytes(b);
        }
        *c = T::from_words(&words);
    }
}

/// Marks a Bevy entity whose device slot is currently NOT alive, with the frame at which that was first seen.  A predicted
/// despawn can still be rolled back (LoadWorld resurrects the slot, src/snapshot/entity.rs:62-98), so the Bevy entity -- and every
/// non-mirrored component it carries (meshes, handles) -- must survive until the despawn is CONFIRMED.  Game and render queries
/// that must not see such entities filter `Without<HipDeadSince>`.
#[derive(Component, Clone, Copy)]
pub struct HipDeadSince(pub i32);

/// Mirrors device-side liveness onto the Bevy entities.  Slots are never reused, so slot <-> entity is 1:1 for the whole session:
///   * slot dead, no marker        -> mark `HipDeadSince(frame)` (the entity is hidden from gameplay, not despawned);
///   * slot alive again, marker    -> a rollback resurrected it: unmark, the entity and its non-mirrored components are intact;
///   * slot dead, marker confirmed -> no rollback can reach a confirmed frame (ConfirmedFrameCount): despawn the Bevy entity now.
fn mirror_despawns(
    mut commands: Commands,
    hip: Res<HipWorld>,
    frame: Res<RollbackFrameCount>,
    confirmed: Res<ConfirmedFrameCount>,
    q: Query<(Entity, &HipSlot, Option<&HipDeadSince>)>,
) {
    let alive = hip.alive_mask();
    for (e, slot, dead) in &q {
        let s = slot.0 as usize;
        if s >= hip.len() as usize {
            continue;
        }
        let is_alive = (alive[s / 64] >> (s % 64)) & 1 == 1;
        match (is_alive, dead) {
            (true, Some(_)) => {
                commands.entity(e).remove::<HipDeadSince>();
            }
            (false, None) => {
                commands.entity(e).insert(HipDeadSince(frame.0));
            }
            (false, Some(d)) if d.0 <= confirmed.0 => {
                commands.entity(e).despawn();
            }
            _ => {}
        }
    }
}

// ------------------------------------------------------------------------------------------------ kernel systems

/// One kernel-backed system of the `GgrsSchedule` (include/ggrs_hip.h `GGRS_SYS_*`).
#[derive(Clone, Copy)]
pub struct KernelSystem {
    desc: ffi::ggrs_system_desc,
    comps: [Option<fn(&HipWorld) -> u32>; 4],
}

/// A per-entity `GgrsSchedule` system written as HIP C++ source and compiled for gfx950 when it is added
/// (include/ggrs_hip.h `ggrs_hip_add_custom_system`): the open counterpart of [`KernelSystem`] for systems of the shape
/// `Query<(&mut A, &mut B, ..), With<Rollback>>` + `Commands::despawn`.  `source` defines
/// `__device__ void ggrs_system(GgrsEntity& e, const GgrsFrame& f)`; binding `i` is seen as `e.f32(i)` / `e.u32(i)` / `e.u64(i)`.
pub struct CustomKernelSystem {
    pub name: &'static str,
    pub source: String,
    pub bindings: Vec<(fn(&HipWorld) -> u32, u32)>,
    pub iparam: [i64; 2],
    pub fparam: [f32; 4],
}
impl CustomKernelSystem {
    pub fn new(name: &'static str, source: impl Into<String>) -> Self {
        CustomKernelSystem { name, source: source.into(), bindings: Vec::new(), iparam: [0; 2], fparam: [0.0; 4] }
    }
    /// Bind word `word` of component `T` as the next `e.*(i)`.
    pub fn bind<T: HipComponent>(mut self, word: u32) -> Self {
        self.bindings.push((HipWorld::comp_id::<T>, word));
        self
    }
    pub fn with_params(mut self, iparam: [i64; 2], fparam: [f32; 4]) -> Self {
        self.iparam = iparam;
        self.fparam = fparam;
        self
    }
}

pub mod systems {
    use super::*;
    /// examples/stress_tests/particles.rs:272-280
    pub fn update_particles<T: HipComponent, V: HipComponent>(gravity: Vec3) -> KernelSystem {
        let mut desc = ffi::ggrs_system_desc { kind: ffi::GGRS_SYS_PARTICLES_UPDATE, ..default() };
        desc.fparam = [gravity.x, gravity.y, gravity.z, 0.0];
        KernelSystem { desc, comps: [Some(HipWorld::comp_id::<T>), Some(HipWorld::comp_id::<V>), None, None] }
    }
    /// examples/stress_tests/particles.rs:282-289
    pub fn despawn_particles<L: HipComponent>() -> KernelSystem {
        KernelSystem { desc: ffi::ggrs_system_desc { kind: ffi::GGRS_SYS_TTL_DESPAWN, ..default() }, comps: [Some(HipWorld::comp_id::<L>), None, None, None] }
    }
    /// examples/stress_tests/particles.rs:254-270.  The per-frame random velocities come from [`SpawnPayload`].
    pub fn spawn_particles<T: HipComponent, V: HipComponent, L: HipComponent>(ttl: usize, input_mask: u8) -> KernelSystem {
        let mut desc = ffi::ggrs_system_desc { kind: ffi::GGRS_SYS_PARTICLES_SPAWN, ..default() };
        desc.iparam = [ttl as i64, input_mask as i64];
        KernelSystem { desc, comps: [Some(HipWorld::comp_id::<T>), Some(HipWorld::comp_id::<V>), Some(HipWorld::comp_id::<L>), None] }
    }
    /// examples/box_game/box_game.rs:154-206
    pub fn move_cube_system<T: HipComponent, V: HipComponent, P: HipComponent>(acceleration: f32, max_speed: f32, friction: f32, half_width: f32) -> KernelSystem {
        let mut desc = ffi::ggrs_system_desc { kind: ffi::GGRS_SYS_BOX_MOVE, ..default() };
        desc.fparam = [acceleration, max_speed, friction, half_width];
        KernelSystem { desc, comps: [Some(HipWorld::comp_id::<T>), Some(HipWorld::comp_id::<V>), Some(HipWorld::comp_id::<P>), None] }
    }
}

/// Host-side source of what `spawn_particles` draws from its rolled-back `ParticleRng` (particles.rs:125,243,265):
/// a PURE function of the frame being advanced -> (vx, vy) arrays, so that a resimulated frame redraws the same values
/// (the reference gets that from snapshotting the RNG resource).  `rate` = particles per spawning frame.
#[derive(Resource)]
pub struct SpawnPayload {
    pub rate: usize,
    pub input_mask: u8,
    pub draw: Box<dyn Fn(i32) -> (Vec<f32>, Vec<f32>) + Send + Sync>,
}

/// Host side of a user-written spawn system (`ggrs_hip_add_spawn_system`; `commands.spawn((.., Rollback))` from a GgrsSchedule system,
/// src/snapshot/rollback.rs:45-59): a PURE function of (frame being advanced, that frame's PlayerInputs bytes, their InputStatus bytes) ->
/// how many entities the frame spawns and the payload blob the device-side `ggrs_spawn` reads.  Pure, so that a resimulated frame
/// spawns the same entities.
#[derive(Resource)]
pub struct SpawnBlob {
    pub draw: Box<dyn Fn(i32, &[u8], &[u8]) -> (u64, Vec<u8>) + Send + Sync>,
}

/// `T::Input` as the device sees it: `BYTES` little-endian bytes per player (`ggrs_hip_set_input_layout`).  ggrs only asks inputs to be
/// `Copy + Serialize`; the device path needs their plain bytes, so a game's input type says how (one line for a newtype over an integer).
pub trait DeviceInput: Copy {
    const BYTES: usize;
    fn write_device_bytes(&self, out: &mut [u8]);
}
macro_rules! device_input_int { ($($t:ty),*) => { $(impl DeviceInput for $t {
    const BYTES: usize = core::mem::size_of::<$t>();
    fn write_device_bytes(&self, out: &mut [u8]) { out.copy_from_slice(&self.to_le_bytes()); }
})* } }
device_input_int!(u8, u16, u32, u64, i8, i16, i32, i64);
impl<const N: usize> DeviceInput for [u8; N] {
    const BYTES: usize = N;
    fn write_device_bytes(&self, out: &mut [u8]) { out.copy_from_slice(self); }
}
fn status_byte(s: InputStatus) -> u8 {
    match s { InputStatus::Confirmed => ffi::GGRS_INPUT_CONFIRMED, InputStatus::Predicted => ffi::GGRS_INPUT_PREDICTED, InputStatus::Disconnected => ffi::GGRS_INPUT_DISCONNECTED }
}

// ------------------------------------------------------------------------------------------------ registration

/// Same method names as `bevy_ggrs::RollbackApp` (src/snapshot/rollback_app.rs:31-133) for the component kinds the
/// device path owns; resources, reflect and hierarchy strategies keep going through bevy_ggrs's own trait.
/// `add_systems(GgrsSchedule, ..)` for a system that SPAWNS Rollback entities -- `commands.spawn((bundle.., Rollback))`
/// (src/snapshot/rollback.rs:45-59; examples/stress_tests/particles.rs:254-270) -- as HIP C++ source defining
/// `__device__ void ggrs_spawn(GgrsEntity& e, ggrs_u64 k, const GgrsFrame& f, const unsigned char* payload)`
/// (include/ggrs_hip.h `ggrs_hip_add_spawn_system`).  How many entities a frame spawns and the payload they are built from
/// come from the [`SpawnBlob`] resource.
pub struct SpawnKernelSystem {
    pub name: &'static str,
    pub source: String,
    pub bundle: Vec<fn(&HipWorld) -> u32>,
    pub bindings: Vec<(fn(&HipWorld) -> u32, u32)>,
    pub payload_stride: u32,
    pub iparam: [i64; 2],
    pub fparam: [f32; 4],
}
impl SpawnKernelSystem {
    pub fn new(name: &'static str, source: impl Into<String>) -> Self {
        SpawnKernelSystem { name, source: source.into(), bundle: Vec::new(), bindings: Vec::new(), payload_stride: 0, iparam: [0; 2], fparam: [0.0; 4] }
    }
    /// Every spawned entity gets component `T` (at its registered default unless the spawner writes it).
    pub fn with<T: HipComponent>(mut self) -> Self {
        self.bundle.push(HipWorld::comp_id::<T>);
        self
    }
    /// Bind word `word` of component `T` as the next `e.*(i)` the spawner writes.
    pub fn bind<T: HipComponent>(mut self, word: u32) -> Self {
        self.bindings.push((HipWorld::comp_id::<T>, word));
        self
    }
    /// Bytes of payload per spawned entity (0: one blob per AdvanceFrame).
    pub fn stride(mut self, bytes: u32) -> Self {
        self.payload_stride = bytes;
        self
    }
}

pub trait RollbackApp {
    fn rollback_component_with_copy<T: HipComponent>(&mut self) -> &mut Self;
    fn rollback_component_with_clone<T: HipComponent>(&mut self) -> &mut Self;
    fn rollback_immutable_component_with_copy<T: HipComponent>(&mut self) -> &mut Self;
    fn rollback_immutable_component_with_clone<T: HipComponent>(&mut self) -> &mut Self;
    fn checksum_component_with_hash<T: HipComponent>(&mut self) -> &mut Self;
    /// `checksum_component::<T>(fn(&T) -> u64)` with the closure replaced by the list of hashed words
    /// (each word is fed to SeaHash as its little-endian bytes, i.e. `write_u32` / `write_u64`).
    fn checksum_component<T: HipComponent>(&mut self, hashed_words: &[u32]) -> &mut Self;
    /// `checksum_component::<T>(fn(&T) -> u64)` with an ARBITRARY hasher (rollback_app.rs:119-121): HIP C++ source defining
    /// `__device__ ggrs_u64 ggrs_hash(const GgrsComponent& c)` (include/ggrs_hip.h `ggrs_hip_checksum_component_custom`); the
    /// closure body is written once more in device code -- `GgrsHasher h; h.write_u32(c.u32(0)); ..; return h.finish();` is
    /// `checksum_hasher()` + `Hash::hash` calls.  Compiled into the world's generated kernel when the world is sealed.
    fn checksum_component_with_source<T: HipComponent>(&mut self, hasher_source: &str) -> &mut Self;
    fn add_kernel_system(&mut self, schedule: GgrsSchedule, system: KernelSystem) -> &mut Self;
    /// `add_systems(GgrsSchedule, ..)` for a user-written per-entity system (HIP C++ source, compiled at registration).
    fn add_custom_kernel_system(&mut self, schedule: GgrsSchedule, system: CustomKernelSystem) -> &mut Self;
    /// `add_systems(GgrsSchedule, ..)` for a user-written system that spawns Rollback entities (one per world).
    fn add_spawn_kernel_system(&mut self, schedule: GgrsSchedule, system: SpawnKernelSystem) -> &mut Self;
    /// `rollback_component_with::<S>()` for a `Strategy` whose `Stored` differs from the component (src/snapshot/strategy.rs:22-40):
    /// `source` defines `ggrs_store(const GgrsWords& target, GgrsWords& stored)` and `ggrs_load(const GgrsWords& stored, GgrsWords& target)`
    /// (include/ggrs_hip.h `ggrs_hip_register_component_strategy`); ring slots then hold `stored_n_words` words of `stored_word_bytes`.
    fn rollback_component_with_strategy<T: HipComponent>(&mut self, stored_word_bytes: u32, stored_n_words: u32, source: &str) -> &mut Self;
    /// Keep the Bevy copy of `T` current for host-side readers (rendering); off by default: it is a device -> host
    /// copy of the whole column every rendered frame.
    fn mirror_component<T: HipComponent>(&mut self) -> &mut Self;
}

fn hip_world(app: &mut App) -> Mut<'_, HipWorld> {
    if !app.world().contains_resource::<HipWorld>() {
        let cfg = *app.world().get_resource::<HipWorldConfig>().expect("insert HipWorldConfig before registering components");
        app.insert_resource(HipWorld::new(cfg));
    }
    app.world_mut().resource_mut::<HipWorld>()
}

fn register<T: HipComponent>(app: &mut App, flags: u32) {
    let mut w = hip_world(app);
    let name = CString::new(T::NAME).unwrap();
    let mut id = 0u32;
    let rc = unsafe { ffi::ggrs_hip_register_component_ex(w.raw, name.as_ptr(), T::WORD_BYTES, T::N_WORDS, flags, &mut id) };
    w.check(rc);
    w.comp_ids.insert(core::any::TypeId::of::<T>(), id);
    w.uploaders.push(stage_component::<T>);
}

impl RollbackApp for App {
    fn rollback_component_with_copy<T: HipComponent>(&mut self) -> &mut Self {
        register::<T>(self, ffi::GGRS_COMP_ROLLBACK);
        self
    }
    fn rollback_component_with_clone<T: HipComponent>(&mut self) -> &mut Self {
        register::<T>(self, ffi::GGRS_COMP_ROLLBACK); // `Clone` of a POD is the same bits (src/snapshot/strategy.rs:62-83)
        self
    }
    fn rollback_immutable_component_with_copy<T: HipComponent>(&mut self) -> &mut Self {
        // ImmutableComponentSnapshotPlugin always re-inserts (component_snapshot.rs:218-245); on the device a load
        // rewrites words + presence bit of every slot, which is that re-insertion
        register::<T>(self, ffi::GGRS_COMP_ROLLBACK);
        self
    }
    fn rollback_immutable_component_with_clone<T: HipComponent>(&mut self) -> &mut Self {
        self.rollback_immutable_component_with_copy::<T>()
    }
    fn checksum_component_with_hash<T: HipComponent>(&mut self) -> &mut Self {
        let all: Vec<u32> = (0..T::N_WORDS).collect();
        self.checksum_component::<T>(&all)
    }
    fn checksum_component<T: HipComponent>(&mut self, hashed_words: &[u32]) -> &mut Self {
        let w = hip_world(self);
        let rc = unsafe { ffi::ggrs_hip_checksum_component(w.raw, w.comp_id::<T>(), hashed_words.as_ptr(), hashed_words.len() as u32) };
        w.check(rc);
        self
    }
    fn checksum_component_with_source<T: HipComponent>(&mut self, hasher_source: &str) -> &mut Self {
        let w = hip_world(self);
        let src = std::ffi::CString::new(hasher_source).expect("hasher source contains a NUL byte");
        let rc = unsafe { ffi::ggrs_hip_checksum_component_custom(w.raw, w.comp_id::<T>(), src.as_ptr()) };
        w.check(rc);
        self
    }
    fn add_kernel_system(&mut self, _schedule: GgrsSchedule, system: KernelSystem) -> &mut Self {
        let w = hip_world(self);
        let mut desc = system.desc;
        for (k, f) in system.comps.iter().enumerate() {
            if let Some(f) = f {
                desc.comp[k] = f(&w);
            }
        }
        let rc = unsafe { ffi::ggrs_hip_add_system(w.raw, &desc) };
        w.check(rc);
        self
    }
    fn add_custom_kernel_system(&mut self, _schedule: GgrsSchedule, system: CustomKernelSystem) -> &mut Self {
        let w = hip_world(self);
        let name = std::ffi::CString::new(system.name).expect("system name");
        let source = std::ffi::CString::new(system.source).expect("system source");
        let mut desc = ffi::ggrs_custom_system_desc {
            name: name.as_ptr(),
            source: source.as_ptr(),
            n_bindings: system.bindings.len() as u32,
            comp: [0; ffi::GGRS_CUSTOM_MAX_BINDINGS],
            word: [0; ffi::GGRS_CUSTOM_MAX_BINDINGS],
            iparam: system.iparam,
            fparam: system.fparam,
        };
        assert!(system.bindings.len() <= ffi::GGRS_CUSTOM_MAX_BINDINGS, "a custom kernel system binds at most 8 words");
        for (k, (comp, word)) in system.bindings.iter().enumerate() {
            desc.comp[k] = comp(&w);
            desc.word[k] = *word;
        }
        let rc = unsafe { ffi::ggrs_hip_add_custom_system(w.raw, &desc) };
        w.check(rc); // a compile error panics with the hiprtc log (ggrs_hip_last_error), like a system that fails to build
        self
    }
    fn add_spawn_kernel_system(&mut self, _schedule: GgrsSchedule, system: SpawnKernelSystem) -> &mut Self {
        let w = hip_world(self);
        let name = std::ffi::CString::new(system.name).expect("system name");
        let source = std::ffi::CString::new(system.source).expect("system source");
        assert!(system.bindings.len() <= ffi::GGRS_CUSTOM_MAX_BINDINGS, "a spawn system binds at most 8 words");
        let mut desc = ffi::ggrs_spawn_system_desc {
            name: name.as_ptr(),
            source: source.as_ptr(),
            bundle_mask: 0,
            payload_stride: system.payload_stride,
            n_bindings: system.bindings.len() as u32,
            comp: [0; ffi::GGRS_CUSTOM_MAX_BINDINGS],
            word: [0; ffi::GGRS_CUSTOM_MAX_BINDINGS],
            iparam: system.iparam,
            fparam: system.fparam,
        };
        for comp in system.bundle.iter() {
            desc.bundle_mask |= 1u64 << comp(&w);
        }
        for (k, (comp, word)) in system.bindings.iter().enumerate() {
            desc.comp[k] = comp(&w);
            desc.word[k] = *word;
        }
        let rc = unsafe { ffi::ggrs_hip_add_spawn_system(w.raw, &desc) };
        w.check(rc);
        self
    }
    fn rollback_component_with_strategy<T: HipComponent>(&mut self, stored_word_bytes: u32, stored_n_words: u32, source: &str) -> &mut Self {
        register::<T>(self, ffi::GGRS_COMP_ROLLBACK);
        let w = hip_world(self);
        let source = std::ffi::CString::new(source).expect("strategy source");
        let rc = unsafe { ffi::ggrs_hip_register_component_strategy(w.raw, HipWorld::comp_id::<T>(&w), stored_word_bytes, stored_n_words, source.as_ptr()) };
        w.check(rc);
        self
    }
    fn mirror_component<T: HipComponent>(&mut self) -> &mut Self {
        self.add_systems(PreUpdate, mirror_component_system::<T>.after(drive_session_marker));
        self
    }
}

// ------------------------------------------------------------------------------------------------ plugin + driver

/// `GgrsPlugin`: same name, same generic parameter, same default schedule (`PreUpdate`, src/lib.rs:214-224).  Unlike
/// a wrapper around `bevy_ggrs::GgrsPlugin` it installs ITS OWN session-driving system, because the stock one calls
/// the stock `handle_requests` (src/schedule_systems.rs:98,123,156) and the device world would never be driven.
pub struct GgrsPlugin<C: Config> {
    _marker: PhantomData<C>,
}
impl<C: Config> Default for GgrsPlugin<C> {
    fn default() -> Self {
        Self { _marker: PhantomData }
    }
}
/// Ordering anchor for systems that must run after the tick (`mirror_component`).
fn drive_session_marker() {}

impl<C: Config> Plugin for GgrsPlugin<C> where C::Input: DeviceInput {
    fn build(&self, app: &mut App) {
        // What src/lib.rs:227-259 registers, minus everything that lives on the device now (SnapshotPlugin's
        // component / entity snapshots, ChecksumPlugin, EntityChecksumPlugin) and with GgrsTimePlugin's dt rule
        // evaluated inside the library (time.rs:63-87 == ggrs_hip.hip dt_bits_for_frame).
        app.init_resource::<MaxPredictionWindow>()
            .init_resource::<LocalPlayers>()
            .init_resource::<Pacer>()
            .init_resource::<PendingSpawns>()
            .init_resource::<RollbackFrameCount>()
            .init_resource::<ConfirmedFrameCount>()
            .init_schedule(ReadInputs)
            .init_schedule(GgrsSchedule) // host systems may still be added; they run after the device tick (see handle_requests)
            .add_systems(PreUpdate, (drive_session::<C>, drive_session_marker, mirror_despawns).chain().after(bevy::input::InputSystems));
        let hip = hip_world(app); // creates the device world now so that component registration can follow in any order
        // PlayerInputs<C>: size_of::<C::Input>() bytes + one InputStatus byte per player reach every user-written device system (src/lib.rs:98)
        const { assert!(<C::Input as DeviceInput>::BYTES >= 1 && <C::Input as DeviceInput>::BYTES <= ffi::GGRS_MAX_INPUT_BYTES) };
        hip.check(unsafe { ffi::ggrs_hip_set_input_layout(hip.raw, <C::Input as DeviceInput>::BYTES as u32, ffi::GGRS_MAX_PLAYERS as u32) });
    }
}

/// The fixed-timestep accumulator of `run_ggrs_schedules` (src/schedule_systems.rs:19-83), integer nanoseconds:
/// period = 1e9 / fps, stretched by 11/10 while a P2P session reports `frames_ahead() > 0`.
#[derive(Resource, Default)]
struct Pacer {
    banked: Duration,
    slow: bool,
}
impl Pacer {
    fn period(&self, fps: usize) -> Duration {
        let ns = if self.slow { 1_000_000_000u64 * 11 / (fps as u64 * 10) } else { 1_000_000_000u64 / fps as u64 };
        Duration::from_nanos(ns)
    }
    fn take_step(&mut self, fps: usize) -> bool {
        let p = self.period(fps);
        if self.banked < p {
            return false;
        }
        self.banked = self.banked.saturating_sub(p);
        true
    }
}

/// What one session type contributes to a tick; the three implementations are the bodies of `run_synctest`,
/// `run_p2p` and `run_spectator` (src/schedule_systems.rs:85-168).
enum Step<C: Config> {
    Requests(Vec<GgrsRequest<C>>),
    Mismatch { current_frame: i32, mismatched_frames: Vec<i32> },
    Skipped(&'static str),
    Failed(String),
    Idle,
}

fn feed_local_inputs<C: Config>(world: &mut World, mut add: impl FnMut(usize, C::Input)) {
    world.run_schedule(ReadInputs);
    let inputs = world
        .remove_resource::<LocalInputs<C>>()
        .expect("No local player inputs found. Did you insert systems into the ReadInputs schedule?");
    for (handle, input) in inputs.0 {
        add(handle, input);
    }
}

fn step_session<C: Config>(world: &mut World, session: &mut Session<C>, pacer: &mut Pacer) -> Step<C> {
    let classify = |r: Result<Vec<GgrsRequest<C>>, GgrsError>, skip_note: &'static str| match r {
        Ok(reqs) => Step::Requests(reqs),
        Err(GgrsError::MismatchedChecksum { current_frame, mismatched_frames }) => Step::Mismatch { current_frame, mismatched_frames },
        Err(GgrsError::PredictionThreshold) => Step::Skipped(skip_note),
        Err(e) => Step::Failed(e.to_string()),
    };
    match session {
        Session::SyncTest(s) => {
            world.insert_resource(LocalPlayers((0..s.num_players()).collect()));
            feed_local_inputs::<C>(world, |h, i| s.add_local_input(h, i).expect("All handles in local_handles should be valid"));
            classify(s.advance_frame(), "")
        }
        Session::P2P(s) => {
            pacer.slow = s.frames_ahead() > 0; // "if we are ahead, run slow" (:66-67)
            world.insert_resource(LocalPlayers(s.local_player_handles()));
            if s.current_state() != SessionState::Running {
                return Step::Idle;
            }
            feed_local_inputs::<C>(world, |h, i| s.add_local_input(h, i).expect("All handles in local_inputs should be valid"));
            classify(s.advance_frame(), "Skipping a frame: PredictionThreshold.")
        }
        Session::Spectator(s) => {
            if s.current_state() != SessionState::Running {
                return Step::Idle;
            }
            classify(s.advance_frame(), "P2PSpectatorSession: Waiting for input from host.")
        }
    }
}

/// The system `GgrsPlugin` installs: pending spawns -> device, poll the session, then one [`handle_requests`] per
/// elapsed simulation period.
pub fn drive_session<C: Config>(world: &mut World) where C::Input: DeviceInput {
    upload_pending_spawns(world);
    let fps: usize = **world.get_resource_or_insert_with::<RollbackFrameRate>(default);
    let delta = world.get_resource::<Time>().expect("Time resource not found, did you remove it?").delta();
    let mut pacer = world.remove_resource::<Pacer>().expect("GgrsPlugin was not added");
    pacer.banked = pacer.banked.saturating_add(delta);

    // "no matter what, poll remotes and send responses" (:44-55)
    if let Some(mut s) = world.get_resource_mut::<Session<C>>() {
        match &mut *s {
            Session::P2P(p) => p.poll_remote_clients(),
            Session::Spectator(p) => p.poll_remote_clients(),
            Session::SyncTest(_) => {}
        }
    }

    while pacer.take_step(fps) {
        // last step's Checksum(u128)s -> their cells, BEFORE advance_frame() reads them (SyncTest compares there)
        collect_in_flight(world);
        let Some(mut session) = world.remove_resource::<Session<C>>() else {
            // No session yet: the reference resets its time data and frame counters (:70-78).  The DEVICE world is
            // reset the same way, or a session restart would meet a stale frame counter and ring and its first
            // LoadGameState would fail with GGRS_E_NO_SNAPSHOT.
            pacer.banked = Duration::ZERO;
            pacer.slow = false;
            world.insert_resource(LocalPlayers::default());
            world.insert_resource(RollbackFrameCount(0));
            world.insert_resource(ConfirmedFrameCount(-1));
            world.insert_resource(MaxPredictionWindow(8));
            let hip = world.resource::<HipWorld>();
            unsafe {
                hip.check(ffi::ggrs_hip_set_frame(hip.raw, 0));
                hip.check(ffi::ggrs_hip_set_confirmed(hip.raw, 1, -1));
                hip.check(ffi::ggrs_hip_set_depth(hip.raw, 8));
            }
            continue;
        };
        let step = step_session::<C>(world, &mut session, &mut pacer);
        world.insert_resource(session); // handle_requests reads the session (prediction window, confirmed frame)
        match step {
            Step::Requests(reqs) => handle_requests::<C>(reqs, world),
            Step::Mismatch { current_frame, mismatched_frames } => {
                warn!("Detected checksum mismatch during rollback on frame {current_frame}, mismatched frames: {mismatched_frames:?}");
                world.trigger(SyncTestMismatch { current_frame, mismatched_frames });
            }
            Step::Skipped(note) => info!("{note}"),
            Step::Failed(e) => warn!("{e}"),
            Step::Idle => {}
        }
    }
    world.insert_resource(pacer);
}

// ------------------------------------------------------------------------------------------------ handle_requests

/// Replaces `schedule_systems::handle_requests` (src/schedule_systems.rs:170-289): the WHOLE request list of a tick is
/// one device submission (request-group fusion inside the library).
///
/// Checksum hand-back.  ggrs reads a `SaveGameState` cell no earlier than the next `advance_frame()` (SyncTest compares
/// at the top of it, P2P sends checksums from `poll_remote_clients`), so the list is ENQUEUED
/// (`ggrs_hip_enqueue_requests`) and its `Checksum(u128)`s are collected (`ggrs_hip_collect_checksums` -> `cell.save`)
/// by [`drive_session`] at the top of the NEXT simulation step, right before that step's `advance_frame()`: the GPU
/// tick overlaps the rest of the host's frame (rendering, networking).  Build with `--features sync-checksums` to
/// block inside this call instead.
pub fn handle_requests<T: Config>(requests: Vec<GgrsRequest<T>>, world: &mut World) where T::Input: DeviceInput {
    // 1. nothing may still be in flight here (drive_session collected it before advance_frame)
    debug_assert!(world.resource::<HipWorld>().in_flight.is_none());

    // 2. session-derived resources, refreshed as the reference does before the requests run (:197-220)
    enum Kind { SyncTest(i32), P2P(i32), Spectator, None }
    let (max_prediction, kind) = match world.get_resource::<Session<T>>() {
        Some(Session::SyncTest(s)) => (Some(s.max_prediction()), Kind::SyncTest(s.check_distance() as i32)),
        Some(Session::P2P(s)) => (Some(s.max_prediction()), Kind::P2P(s.confirmed_frame())),
        Some(Session::Spectator(_)) => (Some(0), Kind::Spectator),
        None => (None, Kind::None),
    };
    let fps = world.get_resource::<RollbackFrameRate>().map(|r| r.0).unwrap_or(60);
    let payload_rate = world.get_resource::<SpawnPayload>().map(|p| (p.rate, p.input_mask));
    let hip = world.resource::<HipWorld>();
    let raw = hip.raw;
    unsafe {
        hip.check(ffi::ggrs_hip_set_frame_rate(raw, fps as u64));
        if let Some(m) = max_prediction {
            hip.check(ffi::ggrs_hip_set_depth(raw, m as u32)); // sync_depth, src/snapshot/mod.rs:263-273
        }
        match kind {
            // `current_frame - check_distance`, re-evaluated before every request inside the library (:204-208)
            Kind::SyncTest(cd) => { hip.check(ffi::ggrs_hip_set_synctest_check_distance(raw, cd)); }
            Kind::P2P(confirmed) => {
                hip.check(ffi::ggrs_hip_set_synctest_check_distance(raw, -1));
                hip.check(ffi::ggrs_hip_set_confirmed(raw, 1, confirmed)); // `s.confirmed_frame()` (:202)
            }
            // spectators never roll back: confirmed == current frame, refreshed per request (:209-212).  Check
            // distance 0 is exactly that rule (frame - 0) in the library's per-request evaluation.
            Kind::Spectator => { hip.check(ffi::ggrs_hip_set_synctest_check_distance(raw, 0)); }
            Kind::None => {}
        }
    }

    // 3. marshal.  `frame` tracks RollbackFrameCount through the list so that spawn payloads are drawn for the frame
    //    each AdvanceFrame simulates (a pure function of the frame: resimulation redraws the same values).
    let mut frame = hip.frame();
    let mut input_bytes: Vec<Vec<u8>> = Vec::new();       // keep-alive for the pointers below
    let mut status_bytes: Vec<Vec<u8>> = Vec::new();
    let mut payloads: Vec<(Vec<f32>, Vec<f32>)> = Vec::new();
    let mut blobs: Vec<Vec<u8>> = Vec::new();
    let ib = <T::Input as DeviceInput>::BYTES;
    let mut reqs: Vec<ffi::ggrs_request> = Vec::with_capacity(requests.len());
    let mut cells = Vec::new();
    for r in &requests {
        match r {
            GgrsRequest::SaveGameState { cell, frame: f } => {
                cells.push((cell.clone(), *f));
                reqs.push(ffi::ggrs_request { kind: ffi::GGRS_REQ_SAVE, frame: *f, ..ffi::ggrs_request::zeroed() });
            }
            GgrsRequest::LoadGameState { frame: f, .. } => {
                frame = *f;
                reqs.push(ffi::ggrs_request { kind: ffi::GGRS_REQ_LOAD, frame: *f, ..ffi::ggrs_request::zeroed() });
            }
            GgrsRequest::AdvanceFrame { inputs } => {
                // PlayerInputs<T>(Vec<(T::Input, InputStatus)>) (src/lib.rs:98): the input's bytes and its status, per player
                let mut bytes = vec![0u8; ib * inputs.len()];
                for (k, (inp, _)) in inputs.iter().enumerate() { inp.write_device_bytes(&mut bytes[k * ib..(k + 1) * ib]); }
                input_bytes.push(bytes);
                status_bytes.push(inputs.iter().map(|(_, st)| status_byte(*st)).collect());
                let bytes = input_bytes.last().unwrap();
                let status = status_bytes.last().unwrap();
                let mut q = ffi::ggrs_request { kind: ffi::GGRS_REQ_ADVANCE, inputs: bytes.as_ptr(), status: status.as_ptr(), n_inputs: inputs.len() as u32, ..ffi::ggrs_request::zeroed() };
                if let Some(blob) = world.get_resource::<SpawnBlob>() {
                    let (count, payload) = (blob.draw)(frame, bytes, status);
                    if count > 0 {
                        blobs.push(payload);
                        let p = blobs.last().unwrap();
                        q.spawn_count = count;
                        q.spawn_payload = p.as_ptr() as *const core::ffi::c_void;
                        q.spawn_payload_bytes = p.len() as u64;
                    }
                }
                if let Some((rate, mask)) = payload_rate {
                    if bytes.iter().step_by(ib).any(|b| b & mask != 0) {
                        let draw = &world.resource::<SpawnPayload>().draw;
                        payloads.push(draw(frame));
                        let (vx, vy) = payloads.last().unwrap();
                        assert!(vx.len() == rate && vy.len() == rate);
                        q.spawn_count = rate as u64;
                        q.spawn_vx = vx.as_ptr();
                        q.spawn_vy = vy.as_ptr();
                    }
                }
                reqs.push(q);
                frame += 1;
            }
        }
    }

    // 4. submit
    #[cfg(not(feature = "sync-checksums"))]
    {
        let mut n_saves = 0u32;
        hip.check(unsafe { ffi::ggrs_hip_enqueue_requests(raw, reqs.as_ptr(), reqs.len() as u32, &mut n_saves) });
        debug_assert_eq!(n_saves as usize, cells.len());
        world.resource_mut::<HipWorld>().in_flight = Some(cells);
    }
    #[cfg(feature = "sync-checksums")]
    {
        let mut sums = vec![0u64; 2 * cells.len() + 2];
        hip.check(unsafe { ffi::ggrs_hip_handle_requests(raw, reqs.as_ptr(), reqs.len() as u32, sums.as_mut_ptr()) });
        for (k, (cell, f)) in cells.into_iter().enumerate() {
            cell.save(f, None, Some((sums[2 * k] as u128) | ((sums[2 * k + 1] as u128) << 64)));
        }
    }

    // 5. the counters user systems read; host systems of the GgrsSchedule (those that do not touch device columns)
    //    run once per AdvanceFrame of the list, with PlayerInputs set, after the device tick has been queued
    let hip = world.resource::<HipWorld>();
    let now = hip.frame();
    world.insert_resource(RollbackFrameCount(now));
    if let Some(m) = max_prediction {
        world.insert_resource(MaxPredictionWindow(m));
    }
    match kind {
        Kind::SyncTest(cd) if now - cd >= 0 => world.insert_resource(ConfirmedFrameCount(now - cd)),
        Kind::P2P(c) => world.insert_resource(ConfirmedFrameCount(c)),
        Kind::Spectator => world.insert_resource(ConfirmedFrameCount(now)),
        _ => {}
    }
    for r in requests {
        if let GgrsRequest::AdvanceFrame { inputs } = r {
            world.insert_resource(PlayerInputs::<T>(inputs));
            world.run_schedule(GgrsSchedule);
            world.remove_resource::<PlayerInputs<T>>();
        }
    }
}

/// `cell.save(frame, None, Some(checksum))` (src/schedule_systems.rs:231-236) for the batch enqueued by the previous
/// call; `as u128` of a u64 hash -- the upper half is always 0.
fn collect_in_flight(world: &mut World) {
    let Some(cells) = world.resource_mut::<HipWorld>().in_flight.take() else { return };
    let hip = world.resource::<HipWorld>();
    if unsafe { ffi::ggrs_hip_pending_batches(hip.raw) } == 0 {
        return;
    }
    let mut sums = vec![0u64; 2 * cells.len() + 2];
    let mut got = 0u32;
    hip.check(unsafe { ffi::ggrs_hip_collect_checksums(hip.raw, sums.as_mut_ptr(), cells.len() as u32 + 1, &mut got) });
    assert_eq!(got as usize, cells.len());
    for (k, (cell, f)) in cells.into_iter().enumerate() {
        cell.save(f, None, Some((sums[2 * k] as u128) | ((sums[2 * k + 1] as u128) << 64)));
    }
}
