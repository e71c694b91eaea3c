//! Raw bindings to `include/ggrs_hip.h` (ABI version 9) -- what `bindgen` emits, by hand.
//! UN-BUILT SOURCE: kept in lock-step with the header by tests/test_abi.py.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

pub const GGRS_HIP_ABI_VERSION: c_int = 9;

pub const GGRS_OK: c_int = 0;
pub const GGRS_E_INVALID: c_int = -1;
pub const GGRS_E_NO_SNAPSHOT: c_int = -2;
pub const GGRS_E_CAPACITY: c_int = -3;
pub const GGRS_E_HIP: c_int = -4;
pub const GGRS_E_NO_DEVICE: c_int = -5;

pub const GGRS_WORLD_DEFAULT: u32 = 0;
pub const GGRS_WORLD_UNFUSED: u32 = 2;
pub const GGRS_WORLD_NT_COPY: u32 = 4;
pub const GGRS_WORLD_NO_GROUPS: u32 = 8;
pub const GGRS_WORLD_LAYOUT_ONLY: u32 = 16;

pub const GGRS_COMP_ROLLBACK: u32 = 0;
pub const GGRS_COMP_NO_ROLLBACK: u32 = 1;

pub const GGRS_SYS_PARTICLES_UPDATE: u32 = 1;
pub const GGRS_SYS_TTL_DESPAWN: u32 = 2;
pub const GGRS_SYS_PARTICLES_SPAWN: u32 = 3;
pub const GGRS_SYS_ADD_U32: u32 = 4;
pub const GGRS_SYS_SAT_SUB_DESPAWN: u32 = 5;
pub const GGRS_SYS_BOX_MOVE: u32 = 6;
pub const GGRS_SYS_CUSTOM: u32 = 7;
pub const GGRS_SYS_SPAWN_CUSTOM: u32 = 8;
pub const GGRS_MAX_PLAYERS: usize = 16;
pub const GGRS_MAX_INPUT_BYTES: usize = 16;
pub const GGRS_INPUT_CONFIRMED: u8 = 0;
pub const GGRS_INPUT_PREDICTED: u8 = 1;
pub const GGRS_INPUT_DISCONNECTED: u8 = 2;
pub const GGRS_TIMELINE_FIELDS: usize = 7;
pub const GGRS_CUSTOM_MAX_BINDINGS: usize = 8;
pub const GGRS_DESPAWN_IMMEDIATE: i64 = 0;
pub const GGRS_DESPAWN_ROLLBACK: i64 = 1;

pub const GGRS_REQ_SAVE: u32 = 1;
pub const GGRS_REQ_LOAD: u32 = 2;
pub const GGRS_REQ_ADVANCE: u32 = 3;

pub const GGRS_KERNEL_CLASSES: usize = 5;

pub const GGRS_BRANCH_SAVE_LAST: u32 = 1;
pub const GGRS_BRANCH_RETAIN_NEWEST: u32 = 2;
pub const GGRS_BRANCH_RETAIN_ALL: u32 = 4;
pub const GGRS_ADOPT_RECOMPUTE: u32 = 0;
pub const GGRS_ADOPT_BROADCAST: u32 = 1;
/// `ggrs_spawn_system_desc::payload_stride`: the spawn system's counts and payloads come from the entities that called `e.spawn(n)` on the device.
pub const GGRS_SPAWN_PAYLOAD_PARENT: u32 = 0xFFFF_FFFF;

#[repr(C)]
pub struct ggrs_fanout {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ggrs_world {
    _opaque: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_world_desc {
    pub device: i32,
    pub max_depth: u32,
    pub capacity: u64,
    pub stream: *mut c_void,
    pub arena: *mut c_void,
    pub arena_bytes: u64,
    pub flags: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct ggrs_system_desc {
    pub kind: u32,
    pub comp: [u32; 4],
    pub word: [u32; 4],
    pub iparam: [i64; 2],
    pub fparam: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_custom_system_desc {
    pub name: *const c_char,
    pub source: *const c_char,
    pub n_bindings: u32,
    pub comp: [u32; GGRS_CUSTOM_MAX_BINDINGS],
    pub word: [u32; GGRS_CUSTOM_MAX_BINDINGS],
    pub iparam: [i64; 2],
    pub fparam: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_spawn_system_desc {
    pub name: *const c_char,
    pub source: *const c_char,
    pub bundle_mask: u64,
    pub payload_stride: u32,
    pub n_bindings: u32,
    pub comp: [u32; GGRS_CUSTOM_MAX_BINDINGS],
    pub word: [u32; GGRS_CUSTOM_MAX_BINDINGS],
    pub iparam: [i64; 2],
    pub fparam: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_request {
    pub kind: u32,
    pub frame: i32,
    pub dt_bits: u32,
    pub n_inputs: u32,
    pub inputs: *const u8,
    pub status: *const u8,
    pub spawn_count: u64,
    pub spawn_vx: *const f32,
    pub spawn_vy: *const f32,
    pub spawn_payload: *const c_void,
    pub spawn_payload_bytes: u64,
}

/// One entry of a branch step's spawn table: what the world's spawn system appends in a frame (shared by every branch that spawns there).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_branch_spawn {
    pub count: u64,
    pub vx: *const f32,
    pub vy: *const f32,
    pub payload: *const c_void,
    pub payload_bytes: u64,
}

/// `ggrs_hip_fanout_step_branches`: a prefix request list + n_branches x n_frames predicted inputs, expanded by the library.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ggrs_branch_step {
    pub prefix: *const ggrs_request,
    pub n_prefix: u32,
    pub n_branches: u32,
    pub n_frames: u32,
    pub n_inputs: u32,
    pub flags: u32,
    pub n_spawn_table: u32,
    pub inputs: *const u8,
    pub status: *const u8,
    pub spawn_table: *const ggrs_branch_spawn,
    pub spawn_sel: *const u16,
}

impl ggrs_request {
    pub const fn zeroed() -> Self {
        Self { kind: 0, frame: 0, dt_bits: 0, n_inputs: 0, inputs: core::ptr::null(), status: core::ptr::null(), spawn_count: 0, spawn_vx: core::ptr::null(), spawn_vy: core::ptr::null(),
               spawn_payload: core::ptr::null(), spawn_payload_bytes: 0 }
    }
}

unsafe extern "C" {
    // ---- world lifetime
    pub fn ggrs_hip_world_create(device: c_int, capacity: u64, max_depth: u32, out: *mut *mut ggrs_world) -> c_int;
    pub fn ggrs_hip_world_create_ex(desc: *const ggrs_world_desc, out: *mut *mut ggrs_world) -> c_int;
    pub fn ggrs_hip_arena_bytes(capacity: u64, max_depth: u32, n_components: u32, bytes_per_slot: u32) -> u64;
    pub fn ggrs_hip_world_destroy(w: *mut ggrs_world);
    pub fn ggrs_hip_last_error(w: *mut ggrs_world) -> *const c_char;
    pub fn ggrs_hip_abi_version() -> c_int;
    pub fn ggrs_hip_device_count() -> c_int;
    // ---- registration
    pub fn ggrs_hip_register_component(w: *mut ggrs_world, name: *const c_char, word_bytes: u32, n_words: u32, comp_id: *mut u32) -> c_int;
    pub fn ggrs_hip_register_component_ex(w: *mut ggrs_world, name: *const c_char, word_bytes: u32, n_words: u32, flags: u32, comp_id: *mut u32) -> c_int;
    pub fn ggrs_hip_set_component_default(w: *mut ggrs_world, comp_id: u32, words: *const c_void) -> c_int;
    pub fn ggrs_hip_checksum_component(w: *mut ggrs_world, comp_id: u32, word_idx: *const u32, n_idx: u32) -> c_int;
    pub fn ggrs_hip_checksum_component_custom(w: *mut ggrs_world, comp_id: u32, source: *const c_char) -> c_int;
    pub fn ggrs_hip_add_system(w: *mut ggrs_world, desc: *const ggrs_system_desc) -> c_int;
    pub fn ggrs_hip_add_custom_system(w: *mut ggrs_world, desc: *const ggrs_custom_system_desc) -> c_int;
    pub fn ggrs_hip_register_component_strategy(w: *mut ggrs_world, comp_id: u32, stored_word_bytes: u32, stored_n_words: u32, source: *const c_char) -> c_int;
    pub fn ggrs_hip_set_input_layout(w: *mut ggrs_world, input_bytes: u32, max_players: u32) -> c_int;
    pub fn ggrs_hip_add_spawn_system(w: *mut ggrs_world, desc: *const ggrs_spawn_system_desc) -> c_int;
    pub fn ggrs_hip_generated_kernel_source(w: *mut ggrs_world, form: u32, buf: *mut c_char, cap: u64, needed: *mut u64, compile: c_int) -> c_int;
    pub fn ggrs_hip_aot_object_name(source: *const c_char, buf: *mut c_char, cap: u64) -> c_int;
    pub fn ggrs_hip_set_frame_rate(w: *mut ggrs_world, fps: u64) -> c_int;
    // ---- entities and host <-> device column traffic
    pub fn ggrs_hip_spawn(w: *mut ggrs_world, count: u64, comp_mask: u64, cols: *const *const c_void, first_slot: *mut u64) -> c_int;
    pub fn ggrs_hip_despawn(w: *mut ggrs_world, slot: u64) -> c_int;
    pub fn ggrs_hip_despawn_rollback(w: *mut ggrs_world, slot: u64) -> c_int;
    pub fn ggrs_hip_download_disabled(w: *mut ggrs_world, host_dst: *mut u64, n_words64: u64) -> c_int;
    pub fn ggrs_hip_download_despawned_frames(w: *mut ggrs_world, first: u64, count: u64, frames: *mut i32) -> c_int;
    pub fn ggrs_hip_insert_component(w: *mut ggrs_world, comp_id: u32, slot: u64, words: *const c_void) -> c_int;
    pub fn ggrs_hip_remove_component(w: *mut ggrs_world, comp_id: u32, slot: u64) -> c_int;
    pub fn ggrs_hip_upload_word(w: *mut ggrs_world, comp_id: u32, word: u32, first: u64, count: u64, host_src: *const c_void) -> c_int;
    pub fn ggrs_hip_download_word(w: *mut ggrs_world, comp_id: u32, word: u32, first: u64, count: u64, host_dst: *mut c_void) -> c_int;
    pub fn ggrs_hip_download_alive(w: *mut ggrs_world, host_dst: *mut u64, n_words64: u64) -> c_int;
    pub fn ggrs_hip_download_present(w: *mut ggrs_world, comp_id: u32, host_dst: *mut u64, n_words64: u64) -> c_int;
    pub fn ggrs_hip_column_device_ptr(w: *mut ggrs_world, comp_id: u32, word: u32, dev_ptr: *mut *mut c_void, tile_stride: *mut u64) -> c_int;
    pub fn ggrs_hip_len(w: *mut ggrs_world) -> u64;
    pub fn ggrs_hip_active_count(w: *mut ggrs_world, out: *mut u64) -> c_int;
    // ---- frame counters and the snapshot ring
    pub fn ggrs_hip_frame(w: *mut ggrs_world) -> i32;
    pub fn ggrs_hip_set_frame(w: *mut ggrs_world, frame: i32) -> c_int;
    pub fn ggrs_hip_set_depth(w: *mut ggrs_world, depth: u32) -> c_int;
    pub fn ggrs_hip_set_confirmed(w: *mut ggrs_world, has: c_int, confirmed_frame: i32) -> c_int;
    pub fn ggrs_hip_has_snapshot(w: *mut ggrs_world, frame: i32) -> c_int;
    pub fn ggrs_hip_snapshot_count(w: *mut ggrs_world) -> u64;
    // ---- request execution
    pub fn ggrs_hip_save(w: *mut ggrs_world, checksum_out: *mut u64) -> c_int;
    pub fn ggrs_hip_load(w: *mut ggrs_world, frame: i32) -> c_int;
    pub fn ggrs_hip_advance(w: *mut ggrs_world, dt_bits: u32, inputs: *const u8, n_inputs: u32, spawn_count: u64, spawn_vx: *const f32, spawn_vy: *const f32) -> c_int;
    pub fn ggrs_hip_handle_requests(w: *mut ggrs_world, reqs: *const ggrs_request, n: u32, checksums_out: *mut u64) -> c_int;
    pub fn ggrs_hip_set_synctest_check_distance(w: *mut ggrs_world, check_distance: i32) -> c_int;
    pub fn ggrs_hip_enqueue_requests(w: *mut ggrs_world, reqs: *const ggrs_request, n: u32, n_saves_out: *mut u32) -> c_int;
    pub fn ggrs_hip_collect_checksums(w: *mut ggrs_world, checksums_out: *mut u64, max_saves: u32, n_saves_out: *mut u32) -> c_int;
    pub fn ggrs_hip_pending_batches(w: *mut ggrs_world) -> u32;
    pub fn ggrs_hip_synchronize(w: *mut ggrs_world) -> c_int;
    // ---- speculative fan-out support
    pub fn ggrs_hip_state_bytes(w: *mut ggrs_world) -> u64;
    pub fn ggrs_hip_live_state_ptr(w: *mut ggrs_world, dev_ptr: *mut *mut c_void) -> c_int;
    pub fn ggrs_hip_adopt_live_state(w: *mut ggrs_world) -> c_int;
    // ---- speculative fan-out across GPUs (RCCL behind the C ABI; the host carries the 128-byte ncclUniqueId between ranks)
    pub fn ggrs_hip_fanout_unique_id(id_out: *mut u8) -> c_int;
    pub fn ggrs_hip_fanout_init(w: *mut ggrs_world, id: *const u8, rank: c_int, world_size: c_int, out: *mut *mut ggrs_fanout) -> c_int;
    pub fn ggrs_hip_fanout_sync_confirmed(f: *mut ggrs_fanout, root: c_int) -> c_int;
    pub fn ggrs_hip_fanout_step(f: *mut ggrs_fanout, reqs: *const ggrs_request, n: u32, n_saves_out: *mut u32) -> c_int;
    pub fn ggrs_hip_fanout_set_interval(f: *mut ggrs_fanout, steps_per_all_gather: u32) -> c_int;
    pub fn ggrs_hip_fanout_collect(f: *mut ggrs_fanout, checksums_out: *mut u64, max_u128_per_rank: u32, n_steps_out: *mut u32, n_saves_out: *mut u32) -> c_int;
    pub fn ggrs_hip_fanout_destroy(f: *mut ggrs_fanout);
    pub fn ggrs_hip_fanout_last_error(f: *mut ggrs_fanout) -> *const c_char;
    pub fn ggrs_hip_fanout_comm_info(f: *mut ggrs_fanout, rank_out: *mut c_int, size_out: *mut c_int, device_out: *mut c_int) -> c_int;
    pub fn ggrs_hip_fanout_step_branches(f: *mut ggrs_fanout, step: *const ggrs_branch_step, n_saves_out: *mut u32) -> c_int;
    pub fn ggrs_hip_fanout_adopt(f: *mut ggrs_fanout, branch: u32, frame: i32, mode: u32, replay: *const ggrs_request, n_replay: u32, checksums_out: *mut u64, n_checksums_out: *mut u32) -> c_int;
    // ---- measurement hooks
    pub fn ggrs_hip_profile_enable(w: *mut ggrs_world, on: c_int) -> c_int;
    pub fn ggrs_hip_profile_read(w: *mut ggrs_world, ms_out: *mut f64, launches_out: *mut u64) -> c_int;
    pub fn ggrs_hip_profile_read_launches(w: *mut ggrs_world, kernel_class: u32, us_out: *mut f32, cap: u32, n_out: *mut u32) -> c_int;
    pub fn ggrs_hip_profile_read_bytes(w: *mut ggrs_world, bytes_out: *mut u64) -> c_int;
    pub fn ggrs_hip_host_timeline(w: *mut ggrs_world, enable: c_int, us_out: *mut f64, counts_out: *mut u64) -> c_int;
    pub fn ggrs_hip_specialise_wait(w: *mut ggrs_world) -> c_int;
    pub fn ggrs_hip_world_kernel_info(w: *mut ggrs_world, buf: *mut c_char, cap: u64, needed: *mut u64) -> c_int;
}
