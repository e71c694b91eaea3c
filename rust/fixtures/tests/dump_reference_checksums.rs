//! Dumps what the REAL bevy_ggrs computes on this repo's synthetic workloads, so that the CPU oracle (and through it the HIP
//! path) can be pinned to the reference itself: every SaveWorld of every scenario is recorded as
//!     { frame, checksum (the `Checksum(u128)` handed to `cell.save`), parts: { <label>: ChecksumPart } }
//! in request order, plus component bits where a scenario's checksums do not cover the state (box_game).
//!
//! Scenarios (BASELINE.json configs; tests/test_reference_fixtures.py consumes every one of them):
//!   config2_particles_10k / config3_particles_1m   stress_test systems (examples/stress_tests/particles.rs:272-289) under a SyncTest
//!                                                  session of check distance 8
//!   config4_rollback_<r>, r = 1..=7                the same world at 100 k entities under check distance r: every tick is the request
//!                                                  list of an r-frame rollback, [Load(F-r), Adv, (Save, Adv) x (r-1), Save(F), Adv] --
//!                                                  the shapes a P2P session at 120 ms RTT sends (handle_requests is pub(crate) and a
//!                                                  P2P session's rollback lengths depend on socket timing, so the shapes are produced
//!                                                  one length per session)
//!   despawn_immediate / despawn_rollback           tests/synctest.rs:26-75 (Health counts down, the entity is despawned at 0) over 64
//!                                                  entities with staggered health, with `despawn()` and with `despawn_rollback()`
//!   config1_box_game                               (cargo feature `box_game`) examples/box_game: 2 players, check distance 7, input
//!                                                  delay 2, scripted inputs; translation / velocity bits of both cubes after every tick
//!
//! The harness is the reference's own (tests/common/mod.rs:26-55: MinimalPlugins, manual 60 fps time step, one SyncTest session,
//! GgrsPlugin).  Inputs come from rust/fixtures/inputs/*.bin (tests/golden/make_reference_inputs.py): the same numpy-generated
//! velocities / ttls / input bytes the Python and C++ sides use, so the dump is comparable bit for bit.
//!
//! WHERE the checksum is read (What's weak 1).  `Checksum` is written by `ChecksumPlugin::update`, which is ordered
//! `.after(SaveWorldSystems::Checksum).before(SaveWorldSystems::Snapshot)` (src/snapshot/checksum.rs:119-124).  A recorder ordered only
//! `.after(SaveWorldSystems::Checksum)` is UNORDERED with respect to it and may read the previous frame's value; the recorder here runs
//! `.after(SaveWorldSystems::Snapshot)`, i.e. behind everything SaveWorld does -- it sees exactly what `handle_requests` reads right after
//! `save_world_schedule.run(world)` (src/schedule_systems.rs:223-237).  Frame 0 is the one place a real run may differ from a
//! restatement: the `ChecksumPart` entities are created through `Commands` on the first SaveWorld (component_checksum.rs:103-107) and
//! reach `ChecksumPlugin::update` only through the schedule's automatic sync point -- hence the per-part dump, and a separate check
//! of frame 0 on the consuming side.
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
#[derive(Component, Hash, Debug, Clone, Copy)]
struct Health(u32);

/// Scripted inputs: `table[frame % len][handle]` (all zeros when empty).
#[derive(Resource, Default)]
struct InputScript(Vec<Vec<u8>>);

/// One SaveWorld as the reference saw it.
struct SaveRecord {
    frame: i32,
    checksum: u128,
    parts: Vec<(&'static str, u128)>,
}
#[derive(Resource, Default)]
struct Recorded(Vec<SaveRecord>);

fn read_inputs(mut commands: Commands, players: Res<LocalPlayers>, script: Res<InputScript>, frame: Res<RollbackFrameCount>) {
    let mut inputs = HashMap::new();
    for &h in &players.0 {
        let v = if script.0.is_empty() { 0 } else { script.0[(frame.0.max(0) as usize) % script.0.len()][h] };
        inputs.insert(h, v);
    }
    commands.insert_resource(LocalInputs::<Cfg>(inputs));
}

/// Runs behind the WHOLE SaveWorld schedule (see the module docs): the total and every part, labelled by the `ChecksumFlag<T>` that
/// tags it (component_checksum.rs:103-107, entity_checksum.rs:46-50, resource_checksum.rs:75-79).
#[allow(clippy::type_complexity)]
fn record_checksum(
    frame: Res<RollbackFrameCount>,
    checksum: Res<Checksum>,
    parts: Query<(
        &ChecksumPart,
        Option<&ChecksumFlag<Velocity>>,
        Option<&ChecksumFlag<Transform>>,
        Option<&ChecksumFlag<Health>>,
        Option<&ChecksumFlag<Entity>>,
    )>,
    mut rec: ResMut<Recorded>,
) {
    let mut out = Vec::new();
    for (p, v, t, h, e) in &parts {
        let label = if v.is_some() {
            "Velocity"
        } else if t.is_some() {
            "Transform"
        } else if h.is_some() {
            "Health"
        } else if e.is_some() {
            "Entity"
        } else {
            "other"
        };
        out.push((label, p.0));
    }
    out.sort();
    rec.0.push(SaveRecord { frame: frame.0, checksum: checksum.0, parts: out });
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
/// tests/synctest.rs:37-44
fn decrease_health(mut commands: Commands, mut players: Query<(Entity, &mut Health)>) {
    for (entity, mut health) in &mut players {
        health.0 = health.0.saturating_sub(1);
        if health.0 == 0 {
            commands.entity(entity).despawn();
        }
    }
}
/// the same with the deferred despawn of src/snapshot/despawn.rs:114-143
fn decrease_health_rollback(mut commands: Commands, mut players: Query<(Entity, &mut Health)>) {
    for (entity, mut health) in &mut players {
        health.0 = health.0.saturating_sub(1);
        if health.0 == 0 {
            commands.entity(entity).despawn_rollback();
        }
    }
}

fn inputs_dir() -> PathBuf {
    PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("inputs")
}
fn read_f32(name: &str) -> Vec<f32> {
    fs::read(inputs_dir().join(name))
        .expect("run tests/golden/make_reference_inputs.py first")
        .chunks_exact(4)
        .map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]]))
        .collect()
}
fn read_u64(name: &str) -> Vec<u64> {
    fs::read(inputs_dir().join(name)).unwrap().chunks_exact(8).map(|b| u64::from_le_bytes(b.try_into().unwrap())).collect()
}

/// tests/common/mod.rs:46-55 with the session parameters open (`start_synctest_session` wants check_distance < max_prediction).
fn synctest_app(num_players: usize, check_distance: usize, input_delay: usize) -> App {
    let mut b = SessionBuilder::<Cfg>::new()
        .with_num_players(num_players)
        .unwrap()
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
        .add_systems(SaveWorld, record_checksum.after(SaveWorldSystems::Snapshot));
    app
}

fn saves_json(app: &App) -> Vec<serde_json::Value> {
    app.world()
        .resource::<Recorded>()
        .0
        .iter()
        .map(|r| {
            let parts: serde_json::Map<String, serde_json::Value> =
                r.parts.iter().map(|(k, v)| (k.to_string(), serde_json::Value::String(format!("{v:#x}")))).collect();
            serde_json::json!({ "frame": r.frame, "checksum": format!("{:#x}", r.checksum), "parts": parts })
        })
        .collect()
}

/// The stress_test registration (examples/stress_tests/particles.rs:187-240) minus the render-only components.
fn particles(tag: &str, n_take: usize, check_distance: usize, ticks: usize) -> serde_json::Value {
    let vel = read_f32(&format!("{tag}_vel.bin")); // n x 2
    let ttl = read_u64(&format!("{tag}_ttl.bin"));
    let n = ttl.len().min(n_take);
    let mut app = synctest_app(1, check_distance, 0);
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
    serde_json::json!({ "entities": n, "check_distance": check_distance, "ticks": ticks, "saves": saves_json(&app) })
}

/// tests/synctest.rs:46-75 over 64 entities with health 1 + (i % 10): entities die on ten different frames, inside and outside
/// resimulated windows (check distance 5).
fn despawn_scenario(deferred: bool, ticks: usize) -> serde_json::Value {
    let n = 64usize;
    let mut app = synctest_app(1, 5, 0);
    app.rollback_component_with_copy::<Health>().checksum_component_with_hash::<Health>();
    if deferred {
        app.add_systems(GgrsSchedule, decrease_health_rollback);
    } else {
        app.add_systems(GgrsSchedule, decrease_health);
    }
    for i in 0..n {
        app.world_mut().spawn((Health(1 + (i as u32 % 10)), Rollback));
    }
    let mut alive_after_tick = Vec::new();
    for _ in 0..ticks {
        app.update();
        // default query filters hide RollbackDespawned entities: this is what game systems see
        let alive = app.world_mut().query::<&Health>().iter(app.world()).count();
        alive_after_tick.push(alive);
    }
    serde_json::json!({ "entities": n, "check_distance": 5, "ticks": ticks, "deferred": deferred, "health": "1 + (i % 10)",
        "alive_after_tick": alive_after_tick, "saves": saves_json(&app) })
}

#[cfg(feature = "box_game")]
mod box_game_fixture {
    //! `tests/support/box_game.rs` is the reference's examples/box_game/box_game.rs, linked in by hand (rust/fixtures/README.md);
    //! it needs bevy's default features (meshes, materials, keyboard input) to compile.
    use super::*;
    #[path = "support/box_game.rs"]
    #[allow(dead_code)]
    mod box_game;
    use box_game::{BoxConfig, BoxInput, FrameCount, Player, increase_frame_system, move_cube_system};

    #[derive(Resource, Default)]
    struct BoxRecorded(Vec<(i32, u128)>);

    fn read_box_inputs(mut commands: Commands, players: Res<LocalPlayers>, script: Res<InputScript>, frame: Res<RollbackFrameCount>) {
        let mut inputs = HashMap::new();
        for &h in &players.0 {
            let v = script.0[(frame.0.max(0) as usize) % script.0.len()][h];
            // BoxInput is `#[repr(C)] struct BoxInput(u8)` with a private field (box_game.rs:28-30)
            inputs.insert(h, unsafe { core::mem::transmute::<u8, BoxInput>(v) });
        }
        commands.insert_resource(LocalInputs::<BoxConfig>(inputs));
    }
    fn record_box_checksum(frame: Res<RollbackFrameCount>, checksum: Res<Checksum>, mut rec: ResMut<BoxRecorded>) {
        rec.0.push((frame.0, checksum.0));
    }

    pub fn box_game(ticks: usize) -> serde_json::Value {
        let raw = fs::read(inputs_dir().join("config1_inputs.bin")).unwrap();
        let script: Vec<Vec<u8>> = raw.chunks_exact(2).map(|c| c.to_vec()).collect();
        let num_players = 2usize;
        let mut b = SessionBuilder::<BoxConfig>::new().with_num_players(num_players).unwrap().with_check_distance(7).with_input_delay(2);
        for h in 0..num_players {
            b = b.add_player(PlayerType::Local, h).unwrap();
        }
        let mut app = App::new();
        app.add_plugins(MinimalPlugins)
            .insert_resource(TimeUpdateStrategy::ManualDuration(Duration::from_secs_f64(1.0 / 60.0)))
            .insert_resource(Session::SyncTest(b.start_synctest_session().unwrap()))
            .add_plugins(GgrsPlugin::<BoxConfig>::default())
            .insert_resource(InputScript(script))
            .init_resource::<BoxRecorded>()
            .add_systems(ReadInputs, read_box_inputs)
            // box_game_synctest.rs:44-53
            .rollback_resource_with_copy::<FrameCount>()
            .rollback_component_with_copy::<box_game::Velocity>()
            .rollback_component_with_clone::<Transform>()
            .checksum_resource_with_hash::<FrameCount>()
            .add_systems(GgrsSchedule, (move_cube_system, increase_frame_system))
            .insert_resource(FrameCount { frame: 0 })
            .add_systems(SaveWorld, record_box_checksum.after(SaveWorldSystems::Snapshot));
        // setup_system (box_game.rs:89-143) without the meshes: the cubes on a circle of radius PLANE_SIZE / 4
        let r = 5.0f32 / 4.0;
        let mut initial = Vec::new();
        for handle in 0..num_players {
            let rot = handle as f32 / num_players as f32 * 2.0 * std::f32::consts::PI;
            let mut transform = Transform::default();
            transform.translation.x = r * rot.cos();
            transform.translation.y = 0.2 / 2.0;
            transform.translation.z = r * rot.sin();
            initial.push(serde_json::json!([transform.translation.x.to_bits(), transform.translation.y.to_bits(), transform.translation.z.to_bits()]));
            app.world_mut().spawn((transform, Player { handle }, box_game::Velocity::default()));
        }
        let mut per_tick = Vec::new();
        for _ in 0..ticks {
            app.update();
            let frame = app.world().resource::<RollbackFrameCount>().0;
            let mut cubes = Vec::new();
            let mut q = app.world_mut().query::<(&Player, &Transform, &box_game::Velocity)>();
            let mut rows: Vec<(usize, [u32; 6])> = q
                .iter(app.world())
                .map(|(p, t, v)| (p.handle, [t.translation.x.to_bits(), t.translation.y.to_bits(), t.translation.z.to_bits(), v.x.to_bits(), v.y.to_bits(), v.z.to_bits()]))
                .collect();
            rows.sort_by_key(|r| r.0);
            for (_, bits) in rows {
                cubes.push(serde_json::json!(bits));
            }
            per_tick.push(serde_json::json!({ "frame": frame, "frame_count": app.world().resource::<FrameCount>().frame, "cubes": cubes }));
        }
        let saves: Vec<serde_json::Value> =
            app.world().resource::<BoxRecorded>().0.iter().map(|(f, c)| serde_json::json!({ "frame": f, "checksum": format!("{c:#x}") })).collect();
        serde_json::json!({ "players": num_players, "check_distance": 7, "input_delay": 2, "ticks": ticks, "initial_translation_bits": initial,
            "after_tick": per_tick, "saves": saves })
    }
}

#[test]
fn dump() {
    let mut out = serde_json::Map::new();
    out.insert("bevy_ggrs".into(), "0.22.0".into());
    out.insert("recorder".into(), "after SaveWorldSystems::Snapshot (behind ChecksumPlugin::update)".into());
    out.insert("config2_particles_10k".into(), particles("config2", usize::MAX, 8, 24));
    out.insert("config3_particles_1m".into(), particles("config3", usize::MAX, 8, 12));
    for r in 1..=7usize {
        // BASELINE config 4's entity count on the first 100 k rows of config 3's inputs
        out.insert(format!("config4_rollback_{r}"), particles("config3", 100_000, r, 16));
    }
    out.insert("despawn_immediate".into(), despawn_scenario(false, 30));
    out.insert("despawn_rollback".into(), despawn_scenario(true, 30));
    #[cfg(feature = "box_game")]
    out.insert("config1_box_game".into(), box_game_fixture::box_game(40));
    let path = PathBuf::from(env!("CARGO_MANIFEST_DIR")).join("../../tests/golden/reference_checksums.json");
    fs::write(&path, serde_json::to_string_pretty(&serde_json::Value::Object(out)).unwrap()).unwrap();
    println!("wrote {}", path.display());
}
