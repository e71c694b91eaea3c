//! Dumps what the REAL bevy_ggrs computes on this repo's synthetic workloads: every `Checksum(u128)` that SaveWorld
//! produces (in save order, with its RollbackFrameCount) for the particles stress_test at 10 k and 1 M entities under a
//! SyncTest session of check distance 8, and the per-tick cube state of box_game under SyncTest (2 players, check
//! distance 7, input delay 2).  The harness is the reference's own (tests/common/mod.rs:26-55: MinimalPlugins, manual
//! 60 fps time step, single SyncTest session, GgrsPlugin); the game systems restate examples/stress_tests/particles.rs
//! :272-289 and are registered exactly as the example registers them (:187-240) minus the render-only components.
//!
//! Inputs come from rust/fixtures/inputs/*.bin (tests/golden/make_reference_inputs.py): the same numpy-generated
//! velocities / ttls / input bytes the Python and C++ sides use, so the dump is comparable bit for bit.
use bevy::{platform::collections::HashMap, prelude::*, time::TimeUpdateStrategy};
use bevy_ggrs::{prelude::*, *};
use core::hash::{Hash, Hasher};
use core::time::Duration;
use ggrs::*;
use std::{fs, path::PathBuf};

pub struct Cfg;
impl Config for Cfg {
    type Input = u8;
    type State = u8;
    type Address = usize;
    type InputPredictor = ggrs::PredictRepeatLast;
}

#[derive(Component, Clone, Copy, Default)]
struct Velocity(Vec3);
impl Hash for Velocity {
    fn hash<H: Hasher>(&self, s: &mut H) {
        self.0.x.to_bits().hash(s);
        self.0.y.to_bits().hash(s);
        self.0.z.to_bits().hash(s);
    }
}
#[derive(Component, Clone, Copy, Default)]
struct Ttl(usize);

/// Scripted inputs: `table[frame % len][handle]` (all zeros for the particles configs).
#[derive(Resource, Default)]
struct InputScript(Vec<Vec<u8>>);
#[derive(Resource, Default)]
struct Recorded(Vec<(i32, u128)>);

fn read_inputs(mut commands: Commands, players: Res<LocalPlayers>, script: Res<InputScript>, frame: Res<RollbackFrameCount>) {
    let mut inputs = HashMap::new();
    for &h in &players.0 {
        let v = if script.0.is_empty() { 0 } else { script.0[(frame.0.max(0) as usize) % script.0.len()][h] };
        inputs.insert(h, v);
    }
    commands.insert_resource(LocalInputs::<Cfg>(inputs));
}
fn record_checksum(frame: Res<RollbackFrameCount>, checksum: Res<Checksum>, mut rec: ResMut<Recorded>) {
    rec.0.push((frame.0, checksum.0));
}
fn update_particles(mut q: Query<(&mut Transform, &mut Velocity)>, time: Res<Time>) {
    let dt = time.delta_secs();
    let gravity = Vec3::NEG_Y * 200.0;
    for (mut t, mut v) in &mut q {
        v.0 += gravity * dt;
        t.translation += v.0 * dt;
    }
}
fn despawn_particles(mut commands: Commands, mut q: Query<(Entity, &mut Ttl)>) {
    for (e, mut ttl) in &mut q {
        ttl.0 -= 1;
        if ttl.0 == 0 {
            commands.entity(e).despawn();
        }
    }
}

fn inputs_dir() -> PathBuf { PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("inputs") }
fn read_f32(name: &str) -> Vec<f32> {
    fs::read(inputs_dir().join(name)).expect("run tests/golden/make_reference_inputs.py first")
        .chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}
fn read_u64(name: &str) -> Vec<u64> {
    fs::read(inputs_dir().join(name)).unwrap().chunks_exact(8).map(|b| u64::from_le_bytes(b.try_into().unwrap())).collect()
}

fn synctest_app(num_players: usize, check_distance: usize, input_delay: usize) -> App {
    let mut b = SessionBuilder::<Cfg>::new()
        .with_num_players(num_players).unwrap()
        .with_max_prediction_window(check_distance + 1)
        .with_input_delay(input_delay)
        .with_check_distance(check_distance);
    for h in 0..num_players {
        b = b.add_player(PlayerType::Local, h).unwrap();
    }
    let mut app = App::new();
    app.add_plugins(MinimalPlugins)
        .insert_resource(TimeUpdateStrategy::ManualDuration(Duration::from_secs_f64(1.0 / 60.0)))
        .insert_resource(Session::SyncTest(b.start_synctest_session().unwrap()))
        .add_plugins(GgrsPlugin::<Cfg>::default())
        .init_resource::<InputScript>()
        .init_resource::<Recorded>()
        .add_systems(ReadInputs, read_inputs)
        .add_systems(SaveWorld, record_checksum.after(SaveWorldSystems::Checksum));
    app
}

fn particles(tag: &str, ticks: usize) -> serde_json::Value {
    let vel = read_f32(&format!("{tag}_vel.bin"));          // n x 2
    let ttl = read_u64(&format!("{tag}_ttl.bin"));
    let n = ttl.len();
    let mut app = synctest_app(1, 8, 0);
    app.rollback_component_with_clone::<Transform>()
        .rollback_component_with_copy::<Velocity>()
        .rollback_component_with_copy::<Ttl>()
        .checksum_component_with_hash::<Velocity>()
        .checksum_component::<Transform>(|t| {
            let mut h = checksum_hasher();
            t.translation.x.to_bits().hash(&mut h);
            t.translation.y.to_bits().hash(&mut h);
            t.translation.z.to_bits().hash(&mut h);
            h.finish()
        })
        .add_systems(GgrsSchedule, (update_particles, despawn_particles).chain());
    for i in 0..n {
        app.world_mut().spawn((Transform::default(), Velocity(Vec3::new(vel[2 * i], vel[2 * i + 1], 0.0)), Ttl(ttl[i] as usize), Rollback));
    }
    for _ in 0..ticks {
        app.update();
    }
    let rec = &app.world().resource::<Recorded>().0;
    serde_json::json!({ "entities": n, "check_distance": 8, "ticks": ticks,
        "saves": rec.iter().map(|(f, c)| serde_json::json!([f, format!("{c:#x}")])).collect::<Vec<_>>() })
}

#[test]
fn dump() {
    let mut out = serde_json::Map::new();
    out.insert("bevy_ggrs".into(), "0.22.0".into());
    out.insert("config2_particles_10k".into(), particles("config2", 24));
    out.insert("config3_particles_1m".into(), particles("config3", 12));
    // box_game (config 1) needs the example's move_cube_system, which is not part of the published crate: add it here from a
    // checkout (`#[path = ".../examples/box_game/box_game.rs"] mod box_game;`) and dump translation / velocity bits per tick
    // under synctest_app(2, 7, 2) with InputScript = inputs/config1_inputs.bin (40 x 2 bytes).
    let path = PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden/reference_checksums.json");
    fs::write(&path, serde_json::to_string_pretty(&serde_json::Value::Object(out)).unwrap()).unwrap();
    println!("wrote {}", path.display());
}
