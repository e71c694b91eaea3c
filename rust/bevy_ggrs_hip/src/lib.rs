//! `bevy_ggrs_hip` -- drives `libggrs_hip.so` (MI355X / gfx950) from a Bevy app, keeping bevy_ggrs's names.
//!
//! UN-BUILT SOURCE.  The build image has neither a Rust toolchain nor vendored `bevy` / `ggrs`; this crate
//! is the shim a bevy_ggrs maintainer would add.  What is built and tested in this
//! repository is the C ABI it binds (`include/ggrs_hip.h`, mirrored 1:1 in `ffi.rs`) and a C++ twin of this
//! file (`include/bevy_ggrs_hip.hpp`) that runs the reference's integration tests.
//!
//! A user switches by replacing `use bevy_ggrs::prelude::*` with `use bevy_ggrs_hip::prelude::*`:
//!
//! ```ignore
//! App::new()
//!     .add_plugins(GgrsPlugin::<GgrsConfig<u8>>::default())            // same name, device-backed handle_requests
//!     .insert_resource(HipWorldConfig { capacity: 1 << 20, max_depth: 9, device: 0 })
//!     .insert_resource(RollbackFrameRate(60))
//!     .add_systems(ReadInputs, read_local_inputs)
//!     .rollback_component_with_copy::<Velocity>()                     // RollbackApp, same method names
//!     .rollback_component_with_clone::<Transform>()
//!     .checksum_component_with_hash::<Velocity>()
//!     .add_kernel_system(GgrsSchedule, systems::update_particles::<Transform, Velocity>(Vec3::NEG_Y * 200.0))
//!     .insert_resource(Session::SyncTest(session));
//! ```
//!
//! Reference seams replaced (paths relative to the bevy_ggrs repository):
//! * `schedule_systems::handle_requests` (src/schedule_systems.rs:170-289)      -> [`handle_requests`]
//! * `RollbackApp::{rollback,checksum}_component_*` (src/snapshot/rollback_app.rs:31-133) -> [`RollbackApp`]
//! * `GgrsSnapshots<_, _>` + `ComponentSnapshotPlugin::{save,load}` (src/snapshot/*.rs) -> inside the library
//! Everything else (`run_ggrs_schedules`, `ReadInputs`, `LocalInputs`, `Session`, `SyncTestMismatch`,
//! `PlayerInputs`, resources, reflect / hierarchy strategies) stays bevy_ggrs's own code.

pub mod ffi;

use bevy::prelude::*;
use bevy_ggrs::{ConfirmedFrameCount, MaxPredictionWindow, RollbackFrameCount, RollbackFrameRate, Session};
use ggrs::{Config, GgrsRequest};
use std::ffi::{CStr, CString};
use std::marker::PhantomData;

pub mod prelude {
    pub use crate::{systems, GgrsPlugin, HipComponent, HipWorld, HipWorldConfig, KernelSystem, RollbackApp};
    pub use bevy_ggrs::prelude::{
        GgrsConfig, GgrsSchedule, GgrsTime, PlayerInputs, ReadInputs, Rollback, RollbackFrameRate, RollbackId, Session, SyncTestMismatch,
    };
    pub use ggrs::{GgrsEvent, PlayerType, SessionBuilder};
}

/// Shape of the device world, inserted before the first component registration.
#[derive(Resource, Clone, Copy)]
pub struct HipWorldConfig {
    pub capacity: u64,
    pub max_depth: u32,
    pub device: i32,
}

/// The device world: replaces the archetype tables of every registered component and every
/// `GgrsSnapshots<_, _>` resource (src/snapshot/mod.rs:97-119).  Not `Sync`: same exclusive-system
/// contract as the reference (src/lib.rs:252-257).
#[derive(Resource)]
pub struct HipWorld {
    raw: *mut ffi::ggrs_world,
    comp_ids: bevy::platform::collections::HashMap<core::any::TypeId, u32>,
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
        Self { raw, comp_ids: default() }
    }
    fn check(&self, rc: i32) {
        if rc != ffi::GGRS_OK {
            // the reference panics on these paths too (e.g. src/snapshot/mod.rs:213-215)
            panic!("{}", unsafe { CStr::from_ptr(ffi::ggrs_hip_last_error(self.raw)) }.to_string_lossy());
        }
    }
    pub fn comp_id<T: HipComponent>(&self) -> u32 {
        *self.comp_ids.get(&core::any::TypeId::of::<T>()).expect("component is not registered for rollback")
    }
    /// `commands.spawn((bundle, Rollback))` x `count`, columns as SoA slices (None = component default).
    pub fn spawn(&mut self, count: u64, comp_mask: u64, cols: &[*const core::ffi::c_void]) -> u64 {
        let mut first = 0;
        self.check(unsafe { ffi::ggrs_hip_spawn(self.raw, count, comp_mask, if cols.is_empty() { core::ptr::null() } else { cols.as_ptr() }, &mut first) });
        first
    }
    pub fn download_word<T: HipComponent, W: Copy + Default>(&self, word: u32) -> Vec<W> {
        assert_eq!(core::mem::size_of::<W>() as u32, T::WORD_BYTES);
        let n = unsafe { ffi::ggrs_hip_len(self.raw) };
        let mut out = vec![W::default(); n as usize];
        if n > 0 {
            self.check(unsafe { ffi::ggrs_hip_download_word(self.raw, self.comp_id::<T>(), word, 0, n, out.as_mut_ptr().cast()) });
        }
        out
    }
}
impl Drop for HipWorld {
    fn drop(&mut self) {
        unsafe { ffi::ggrs_hip_world_destroy(self.raw) }
    }
}

/// A plain-old-data component whose fields are 4- or 8-byte words, stored as one SoA column per word
/// (Transform = 10 x f32, Velocity = 3 x f32, Ttl = 1 x u64).  `unsafe`: the layout claim must hold.
pub unsafe trait HipComponent: Component + Copy {
    const NAME: &'static str;
    const WORD_BYTES: u32;
    const N_WORDS: u32;
}

/// One kernel-backed system of the `GgrsSchedule` (include/ggrs_hip.h `GGRS_SYS_*`).
#[derive(Clone, Copy)]
pub struct KernelSystem {
    desc: ffi::ggrs_system_desc,
    comps: [Option<fn(&HipWorld) -> u32>; 4],
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
    /// examples/stress_tests/particles.rs:254-270
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

/// Same method names as `bevy_ggrs::RollbackApp` (src/snapshot/rollback_app.rs:31-133) for the component
/// kinds the device path owns; everything else keeps going through bevy_ggrs's own trait.
pub trait RollbackApp {
    fn rollback_component_with_copy<T: HipComponent>(&mut self) -> &mut Self;
    fn rollback_component_with_clone<T: HipComponent>(&mut self) -> &mut Self;
    fn rollback_immutable_component_with_copy<T: HipComponent>(&mut self) -> &mut Self;
    fn checksum_component_with_hash<T: HipComponent>(&mut self) -> &mut Self;
    /// `checksum_component::<T>(fn(&T) -> u64)` with the closure replaced by the list of hashed words
    /// (each word is fed to SeaHash as its little-endian bytes, i.e. `write_u32` / `write_u64`).
    fn checksum_component<T: HipComponent>(&mut self, hashed_words: &[u32]) -> &mut Self;
    fn add_kernel_system(&mut self, schedule: bevy_ggrs::GgrsSchedule, system: KernelSystem) -> &mut Self;
}

fn hip_world(app: &mut App) -> Mut<'_, HipWorld> {
    if !app.world().contains_resource::<HipWorld>() {
        let cfg = *app.world().get_resource::<HipWorldConfig>().expect("insert HipWorldConfig before registering components");
        app.insert_resource(HipWorld::new(cfg));
    }
    app.world_mut().resource_mut::<HipWorld>()
}

fn register<T: HipComponent>(app: &mut App) {
    let mut w = hip_world(app);
    let name = CString::new(T::NAME).unwrap();
    let mut id = 0u32;
    let rc = unsafe { ffi::ggrs_hip_register_component(w.raw, name.as_ptr(), T::WORD_BYTES, T::N_WORDS, &mut id) };
    w.check(rc);
    w.comp_ids.insert(core::any::TypeId::of::<T>(), id);
}

impl RollbackApp for App {
    fn rollback_component_with_copy<T: HipComponent>(&mut self) -> &mut Self {
        register::<T>(self);
        self
    }
    fn rollback_component_with_clone<T: HipComponent>(&mut self) -> &mut Self {
        register::<T>(self); // bitwise for POD (src/snapshot/strategy.rs:62-83)
        self
    }
    fn rollback_immutable_component_with_copy<T: HipComponent>(&mut self) -> &mut Self {
        register::<T>(self); // re-insertion == store words + presence bit (component_snapshot.rs:218-245)
        self
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
    fn add_kernel_system(&mut self, _schedule: bevy_ggrs::GgrsSchedule, system: KernelSystem) -> &mut Self {
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
}

/// `GgrsPlugin` with the device-backed request executor.  `build` is bevy_ggrs's own (src/lib.rs:227-259)
/// except that `run_ggrs_schedules` hands its request list to [`handle_requests`] below.
pub struct GgrsPlugin<C: Config> {
    _marker: PhantomData<C>,
}
impl<C: Config> Default for GgrsPlugin<C> {
    fn default() -> Self {
        Self { _marker: PhantomData }
    }
}
impl<C: Config<Input = u8>> Plugin for GgrsPlugin<C> {
    fn build(&self, app: &mut App) {
        // bevy_ggrs::GgrsPlugin::<C>::default().build(app) with `handle_requests` swapped -- in-tree this is a
        // one-line change at src/schedule_systems.rs:98,123,156 (`handle_requests(requests, world)`).
        app.add_plugins(bevy_ggrs::GgrsPlugin::<C>::default());
    }
}

/// Replaces `schedule_systems::handle_requests` (src/schedule_systems.rs:170-289): the WHOLE request list of
/// a tick is one device submission (request-group fusion inside the library).
pub fn handle_requests<T: Config<Input = u8>>(requests: Vec<GgrsRequest<T>>, world: &mut World) {
    // session-derived resources, refreshed as the reference does before the requests run (:197-220)
    let (max_prediction, check_distance) = match world.get_resource::<Session<T>>() {
        Some(Session::SyncTest(s)) => (Some(s.max_prediction()), s.check_distance() as i32),
        Some(Session::P2P(s)) => (Some(s.max_prediction()), -1),
        Some(Session::Spectator(_)) => (Some(0), -1),
        None => (None, -1),
    };
    let p2p_confirmed = match world.get_resource::<Session<T>>() {
        Some(Session::P2P(s)) => Some(s.confirmed_frame()),
        _ => None,
    };
    let fps = world.get_resource::<RollbackFrameRate>().map(|r| r.0).unwrap_or(60);
    let hip = world.resource::<HipWorld>();
    let raw = hip.raw;
    unsafe {
        ffi::ggrs_hip_set_frame_rate(raw, fps as u64);
        if let Some(m) = max_prediction {
            hip.check(ffi::ggrs_hip_set_depth(raw, m as u32)); // sync_depth, src/snapshot/mod.rs:263-273
        }
        ffi::ggrs_hip_set_synctest_check_distance(raw, check_distance); // the rule of :204-208, applied per request
        if let Some(c) = p2p_confirmed {
            ffi::ggrs_hip_set_confirmed(raw, 1, c);
        }
    }
    let mut inputs: Vec<Vec<u8>> = Vec::new(); // keeps the AdvanceFrame input bytes alive across the call
    let mut reqs: Vec<ffi::ggrs_request> = Vec::with_capacity(requests.len());
    let mut cells = Vec::new();
    for r in &requests {
        match r {
            GgrsRequest::SaveGameState { cell, frame } => {
                cells.push((cell.clone(), *frame));
                reqs.push(ffi::ggrs_request { kind: ffi::GGRS_REQ_SAVE, frame: *frame, ..ffi::ggrs_request::zeroed() });
            }
            GgrsRequest::LoadGameState { frame, .. } => {
                reqs.push(ffi::ggrs_request { kind: ffi::GGRS_REQ_LOAD, frame: *frame, ..ffi::ggrs_request::zeroed() });
            }
            GgrsRequest::AdvanceFrame { inputs: i } => {
                inputs.push(i.iter().map(|(b, _status)| *b).collect());
                let v = inputs.last().unwrap();
                reqs.push(ffi::ggrs_request { kind: ffi::GGRS_REQ_ADVANCE, inputs: v.as_ptr(), n_inputs: v.len() as u32, ..ffi::ggrs_request::zeroed() });
            }
        }
    }
    let mut sums = vec![0u64; 2 * cells.len() + 2];
    hip.check(unsafe { ffi::ggrs_hip_handle_requests(raw, reqs.as_ptr(), reqs.len() as u32, sums.as_mut_ptr()) });
    for (k, (cell, frame)) in cells.into_iter().enumerate() {
        // schedule_systems.rs:231-236: `as u128` of a u64 hash -- the upper half is always 0
        cell.save(frame, None, Some((sums[2 * k] as u128) | ((sums[2 * k + 1] as u128) << 64)));
    }
    // mirror the counters user systems read
    let frame = unsafe { ffi::ggrs_hip_frame(raw) };
    world.insert_resource(RollbackFrameCount(frame));
    if let Some(m) = max_prediction {
        world.insert_resource(MaxPredictionWindow(m));
    }
    if check_distance >= 0 && frame - check_distance >= 0 {
        world.insert_resource(ConfirmedFrameCount(frame - check_distance));
    } else if let Some(c) = p2p_confirmed {
        world.insert_resource(ConfirmedFrameCount(c));
    }
}
