// host_groups.hpp -- fused request groups.  handle_requests (schedule_systems.rs:170-289) receives the WHOLE request list of
// a tick, and every request of a kernel-backed world is slot-local, so a run  [Load?] (Save | Advance)*  executes as ONE
// pass: the source block is read once (the ring slot being loaded, or live), the ops are replayed in request order with the
// slot's words in registers, the live block is written once.  The host does, in request order and while the group is
// assembled, everything the reference's systems do outside the per-entity loops: frame counters, ring push / confirm /
// rollback (exact mirror of mod.rs:121-243 over slot indices), row versions, dirty extents.
// One kernel serves groups: the one generated for the world (kernel_gen.hpp: one 256-slot workgroup per tile).
// Part of the single translation unit ggrs_hip.hip.
#pragma once

namespace {

// value tags: the ids a launch of `members` groups of `n_steps` steps may hand out (tag_base of its argument block).  When the 32-bit counter starts over, no tag of the
// old numbering survives anywhere: every block's tags are ZEROED on the stream (0 = no identity) and its tag_ok bits cleared -- a tag left in a unit that no launch covers
// for a while (a LoadWorld of a frame before a spawn shortens the world) must not meet the same number again.  Callers reserve BEFORE they read any block's tag_ok.
// (Every world crosses its first start-over within its first few dozen launches -- host_seal.hpp, like jiffies -- so the path is run by every test of a tag-keeping world.)
inline int vtags_reserve(ggrs_world* w, uint32_t n_steps, uint32_t members, uint32_t* base) {
    const uint64_t need = (uint64_t)members * (n_steps + 2u);          // per member: one id for "unknown at the load", one per possible count of steps before a store (0 .. n_steps)
    if ((uint64_t)w->tag_counter + need >= 0xFFFFFFF0ull) {
        const uint64_t tag_bytes = w->state_bytes - w->off_tags;
        auto forget = [&](Block& b) -> hipError_t { b.tag_ok = 0; return (b.ptr && tag_bytes) ? hipMemsetAsync(b.ptr + w->off_tags, 0, tag_bytes, w->stream) : hipSuccess; };
        HIPCHK(w, forget(w->live));
        for (auto& b : w->slots) HIPCHK(w, forget(b));
        for (auto& b : w->spec_blocks) HIPCHK(w, forget(b));
        w->tag_counter = 1; ++w->tag_wraps;
    }
    *base = w->tag_counter; w->tag_counter += (uint32_t)need;
    return GGRS_OK;
}
// the tags of a block as a SOURCE / DESTINATION of a launch: columns somebody else may write behind the library's back never count
inline uint64_t block_tagok(const ggrs_world* w, const Block& b) {
    uint64_t m = b.tag_ok & w->tag_cols;
    if (&b == &w->live && w->live_handed_out) m = 0;
    for (uint32_t c = 0; c < w->n_tcols && c < 64; ++c) if (w->col_ext[c]) m &= ~(1ull << c);
    return m;
}
struct GroupState {
    Block* src; uint64_t cover; uint32_t src_is_live;
    Block* dsts[MAX_TICK_SAVES];
    uint64_t save_rows[MAX_TICK_SAVES];               // bit c: column c is stored with Save k (row versions)
    uint32_t save_pmask[MAX_TICK_SAVES];              // bit c: component c's presence mask is stored with Save k
    uint64_t live_rows = 0;                           // columns the group's live write handles
    // (the versions Save k's slot holds once the group has run live in w->group_save_ver, [k][column]: no allocation per group)
};
// LoadGameState opens a group: the ring slot becomes the source (schedule_systems.rs:238-250)
int group_open(ggrs_world* w, const ggrs_request* reqs, uint32_t& i, GroupState& g) {
    g.src = &w->live; g.cover = w->live.dirty_len; g.src_is_live = 1;
    if (reqs[i].kind != GGRS_REQ_LOAD) return GGRS_OK;
    trace_request(w, reqs[i]);
    apply_synctest_confirmed(w);
    w->frame = reqs[i].frame;
    if (!ring_rollback(w, reqs[i].frame))
        return w->fail(GGRS_E_NO_SNAPSHOT, "Could not rollback to %d: no snapshot at that moment could be found.", reqs[i].frame);
    g.src = &w->slots[w->ring_slot.front()];
    w->live_stale.valid = false;                                          // LoadWorld overwrites the live world: what a lazy group left unwritten is never needed
    int rc = launch_load_reconcile(w, *g.src); if (rc) return rc;        // EntityResurrect: before the group rewrites live liveness
    w->len = g.src->len;
    w->cur_ver = g.src->ver;                                              // the logical live state is the snapshot from here on
    // ... except for components under a Strategy: what LoadWorld puts into the world is load(store(x)), which need not be the x the live
    // block (or any other slot) holds under the same version (a lossy Stored form) -- the loaded words are NEW values
    if (w->has_strategy) for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].s_n_words && !w->comps[c].no_rollback) for (uint32_t k = 0; k < w->comps[c].n_words; ++k) ver_touch(w, w->comps[c].col_base + k);
    g.cover = std::max(g.cover, g.src->dirty_len);
    g.src_is_live = 0;
    ++i;
    return GGRS_OK;
}
// column mask -> what it costs per slot.  ring_side: the block is a snapshot -- a component under a Strategy moves as its Stored words there
// (all of them as soon as one of its columns is in the mask), as its own words in the live block
uint64_t rows_bytes_per_slot(const ggrs_world* w, uint64_t mask, bool ring_side) {
    uint64_t b = 0;
    for (const Comp& cc : w->comps) {
        if (cc.no_rollback) continue;
        uint64_t cm = 0;
        for (uint32_t k = 0; k < cc.n_words && cc.col_base + k < 64; ++k) cm |= 1ull << (cc.col_base + k);
        if (ring_side && cc.s_n_words) { if (mask & cm) b += (uint64_t)cc.s_n_words * cc.s_word_bytes; }
        else b += (uint64_t)__builtin_popcountll(mask & cm) * cc.word_bytes;
    }
    return b;
}
// columns of `dst` that differ from the logical live state
uint64_t rows_to_store(const ggrs_world* w, const Block& dst) {
    uint64_t m = 0;
    for (uint32_t c = 0; c < w->n_tcols && c < 64; ++c) if (w->col_rb[c] && ver_differs(w, dst, w->cur_ver, c)) m |= 1ull << c;
    return m;
}
// SaveGameState inside a group: discard_old_snapshots + GgrsSnapshots::push (mod.rs:147-202); the copy itself is an op of the kernel
int group_save(ggrs_world* w, GroupState& g, uint32_t k, uint8_t** save_dst, int32_t* save_frame) {
    if (tracer().on()) { ggrs_request r{}; r.kind = GGRS_REQ_SAVE; r.frame = w->frame; trace_request(w, r); }
    apply_synctest_confirmed(w);
    if (w->has_confirmed) ring_confirm(w, w->confirmed);
    int sl = -1;
    int rc = ring_push(w, w->frame, &sl); if (rc) return rc;
    Block* d = sl >= 0 ? &w->slots[sl] : nullptr;
    g.dsts[k] = d;
    save_dst[k] = d ? d->ptr : nullptr;
    save_frame[k] = w->frame;
    g.save_rows[k] = 0; g.save_pmask[k] = 0;
    if (d) {
        g.cover = std::max(g.cover, d->dirty_len); d->len = w->len;
        g.save_rows[k] = rows_to_store(w, *d);
        g.save_pmask[k] = pmask_differs(w, *d, w->cur_ver);
        const size_t nc = w->cur_ver.size();
        if (w->group_save_ver.size() < (size_t)MAX_TICK_SAVES * nc) w->group_save_ver.resize((size_t)MAX_TICK_SAVES * nc);
        std::copy(w->cur_ver.begin(), w->cur_ver.end(), w->group_save_ver.begin() + (size_t)k * nc);
    }
    return GGRS_OK;
}
// AdvanceFrame inside a group: RollbackFrameCount += 1 (schedule_systems.rs:254-259), DespawnConfirmed, Time<GgrsTime>.
// DespawnConfirmed only touches the live-only marker mask, which no op inside a group reads or writes: queueing it
// ahead of the group's launch keeps request order.
// marks_flags != nullptr: the group kernel keeps the RollbackDespawned markers itself (generated kernel with markers);
// it receives bit 0 = DespawnConfirmed is due before this step (its Local<ConfirmedFrameCount> changed, despawn.rs:92-99),
// bit 1 = the step's frame is unconfirmed, i.e. despawn_rollback() defers (despawn.rs:129-137).
int group_step(ggrs_world* w, const ggrs_request& r, uint32_t* dt_bits_out, uint8_t* marks_flags = nullptr) {
    trace_request(w, r);
    apply_synctest_confirmed(w);
    w->frame += 1;
    if (marks_flags) {
        uint8_t f = 0;
        if (w->confirmed != w->dc_local) { w->dc_local = w->confirmed; f |= 1; }
        if (w->confirmed < w->frame) { f |= 2; w->marks_possible = true; }
        *marks_flags = f;
    } else {
        int rc = step_despawn_confirmed(w); if (rc) return rc;
    }
    ver_step(w);                                                        // every system may have written its write set
    *dt_bits_out = r.dt_bits ? r.dt_bits : dt_bits_for_frame(w->fps, w->frame);
    return GGRS_OK;
}
// dead: the group ran checksum-only (dead-snapshot elimination) -- neither its ring slots nor the live block were written, so
// their dirty extents and row versions still describe what they hold: lowering the extents here would leave mask bits beyond
// the new extent that no later pass cleans (ghost entities once len grows back into those words)
void group_close(ggrs_world* w, GroupState& g, uint32_t n_saves, bool dead, bool wrote_live) {
    w->pending_valid = false;
    if (dead) return;
    const uint64_t new_dirty = std::max(g.src->dirty_len, w->len);
    const size_t nc = w->cur_ver.size();
    for (uint32_t k = 0; k < n_saves; ++k) if (g.dsts[k]) {
        g.dsts[k]->dirty_len = new_dirty;
        std::copy(w->group_save_ver.begin() + (size_t)k * nc, w->group_save_ver.begin() + (size_t)(k + 1) * nc, g.dsts[k]->ver.begin());
        // value tags: every column this Save handled now carries the tag of what it holds (stored with it, or found equal); without the feature a store leaves stale tags behind
        if (w->vtags) g.dsts[k]->tag_ok |= g.save_rows[k] & w->tag_cols; else g.dsts[k]->tag_ok &= ~g.save_rows[k];
    }
    if (wrote_live) { w->live.dirty_len = new_dirty; ver_sync_live(w); if (w->vtags) w->live.tag_ok |= g.live_rows & w->tag_cols; else w->live.tag_ok &= ~g.live_rows; }
}

// Dead-snapshot elimination.  A request group whose NEXT request is a LoadGameState of a frame older than everything the
// group saved leaves nothing behind: that rollback pops every one of its snapshots from the ring (mod.rs:210-226) before
// anything could load them, and LoadWorld overwrites the live world.  Only the group's Checksum(u128)s are observable --
// exactly what a speculative branch of the fan-out is ([Load(C), Adv, Save, ...] x B in one list: every branch but the last).
// Such a group runs checksum-only: no snapshot stores, no live write.  The host ring bookkeeping is done as usual.
// Not applied when something else reads the live world in between: a spawn system the generator could NOT fuse (it then fires as its own
// launches on the live block; a fused spawn runs inside the group and does not count), live-only components or RollbackDespawned markers,
// whose reconcile pass reads the live liveness mask.
bool group_is_dead(const ggrs_world* w, const ggrs_request* reqs, uint32_t i, uint32_t n, const int32_t* save_frame, uint32_t n_saves, bool spawn_pending) {
    if (spawn_pending || n_saves == 0 || i >= n || reqs[i].kind != GGRS_REQ_LOAD || w->has_nr || w->marks_possible) return false;
    bool present = false;
    for (int32_t f : w->ring_frame) present |= f == reqs[i].frame;
    if (!present) return false;                                        // that Load is going to fail: change nothing
    for (uint32_t k = 0; k < n_saves; ++k) {
        const int64_t d = (int64_t)save_frame[k] - (int64_t)reqs[i].frame;
        if (d <= 0 || d > (1 << 30)) return false;                       // not newer (or i32 wrap-around in play): keep it
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// the generated kernel
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint64_t JIT_NT_MIN_SLOTS = 416 * 1024;      // snapshot stores of bigger groups are non-temporal: written once, read a tick later, and the ring does not fit the Infinity Cache
constexpr uint64_t JIT_BATCH_MAX_SLOTS = 400 * 1024;   // identical checksum-only groups ride in one launch while the world is this small
// grid of the per-tile form: 8 x ceil(tiles / 8) workgroups, mapped to tiles XCD by XCD inside the kernel (kernel_gen.hpp)
inline uint32_t jit_grid(uint32_t tiles) { return 8u * ((tiles + 7u) / 8u); }

// FOLD-FORWARD.  Every Save of a group leaves one partial per workgroup and checksummed part (the XOR of a component's entity hashes,
// the live count).  Up to GGRS_FOLD_FORWARD_MIN_WGS workgroups the rows go to pinned memory and the host XORs them at collect time; beyond, that
// fold -- 94 k values = 750 KB per tick at 1 M entities, ~30 us of a host core -- sat between collect(k) and enqueue(k + 2) of a session that
// keeps one tick in flight, i.e. on the path that must stay shorter than one kernel (60 us per step around a 48.7 us kernel).
// Now such a launch leaves its rows in DEVICE memory (two buffers, used alternately) and the NEXT launch on the stream starts with
// saves x parts extra workgroups that fold one row each and write the value, then a tag, into the pinned ring: no atomics, no ticket, no
// second launch, nothing at the end of the producing kernel -- the fold runs beside the next tick's tiles.  collect(k) waits for the batch's
// event, then for the tags (they arrive a few microseconds after kernel k + 1 starts), and hashes 24 values.  When nothing follows on the
// stream by the time the batch is collected, k_ff_fold does the same as its own launch (ff_flush).
// What the previous launch left behind rides along with THIS one:
inline void ff_attach(ggrs_world* w, GgrsJitArgs& j) {
    ggrs_world::FfPending& p = w->ff_pending;
    if (!p.valid) return;
    j.ff_rows = reinterpret_cast<const ggrs_u64*>(w->d_ff_rows[p.buf]); j.ff_out = reinterpret_cast<ggrs_u64*>(w->d_rows + p.out_off); j.ff_seq = p.seq;
    j.ff_nvals = p.nvals; j.ff_blocks = (p.nvals + 7u) & ~7u; j.ff_g = p.g; j.ff_stride = p.stride; j.ff_istride = p.istride; j.ff_split = p.split;
    w->ff_done_id = p.id; p.valid = false;
    w->ff_mark_id = p.id;                                             // launch_jit records the group's event right behind this launch
}

// The shape of a steady SyncTest tick of this world at full length -- [Load(F - D), Advance, (Save, Advance) x D] with D = max_depth - 1, every
// slot live, the rows the systems write -- under the same store / load / role policies run_request_groups_gen applies: what
// ggrs_hip_generated_kernel_source(GGRS_KERNEL_FORM_STEADY) specialises for and what `make aot` ships, so that a session's 16th steady tick finds its kernel
// among the shipped objects when there is no run-time compiler.
inline bool lazy_live_allowed(const ggrs_world* w);
JitSig jit_steady_sig(const ggrs_world* w) {
    // the group caps come from the world's argument-block layout -- computed here, not read from w->cap_*: a GGRS_WORLD_LAYOUT_ONLY world (`make aot` on a
    // build machine) is never sealed, and the defaults (16 / 24) would describe a tick its device struct has no room for
    const JitLayout L = jit_layout(w);
    const uint32_t d = std::min<uint32_t>(std::max<uint32_t>(w->max_depth, 2) - 1, std::min<uint32_t>(L.cap_saves, L.cap_steps - 1));
    const uint64_t cover = w->capacity;
    JitSig g;
    g.n_saves = d; g.n_steps = d + 1; g.n_ops = 2 * d + 1;
    for (uint32_t k = 0; k <= d; ++k) g.op_bits |= 1ull << (2 * k);                       // Advance, Save, Advance, ..., Save, Advance
    g.save_rows = g.live_rows = jit_hot_cols(w); g.load_rows = g.save_rows | jit_static_reads(w);
    g.nt = cover > JIT_NT_MIN_SLOTS ? 1u : 0u;
    g.cached_saves = (g.nt && d >= 2 && rows_bytes_per_slot(w, g.save_rows, true) * cover <= JIT_CACHED_SAVE_MAX_BYTES) ? 1u : 0u;
    g.nt_loads = (g.nt && !g.cached_saves) ? 1u : 0u;
    const JitNeeds need = jit_needs(w);
    if (d >= 2 && !need.marks) g.dp_s = cover <= JIT_DP_MAX_SLOTS ? 1u : (cover <= 2 * JIT_DP_MAX_SLOTS ? 2u : (cover <= 6 * JIT_DP_MAX_SLOTS ? 3u : 0u));
    // an HBM-sized steady session leaves the live block unwritten (lazy live block, below): the next tick's LoadGameState never reads it
    if (g.nt && !need.marks && lazy_live_allowed(w)) { g.skip_live = 1; g.live_rows = 0; }
    g.vtags = vtags_policy(w) ? 1u : 0u;
    return g;
}

// Launch of a generated kernel: the host-side argument block is packed into the world's device layout (kernel_gen.hpp jit_pack).  With
// profiling on, the event pair rides on the dispatch itself (hipExtModuleLaunchKernel's start / stop events: the kernel's own begin and end,
// what rocprofv3's kernel trace reports) instead of bracketing it with two marker packets, which read ~3 us more.
// `done`: an event that completes with THIS launch (enqueue's batch event riding on the list's last kernel instead of a marker packet behind it).
int launch_jit(ggrs_world* w, hipFunction_t fn, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t lds, GgrsJitArgs& j, uint64_t bytes, hipEvent_t done = nullptr) {
    w->spin_n = 0;                                               // a finalize before this launch is no longer the list's last GPU operation (arm_spin)
    ff_attach(w, j);
    j.skip_count = (w->prof && j.vtags) ? reinterpret_cast<ggrs_u64*>(w->d_skip) : nullptr;
    jit_pack(*w->jl, j, w->jit_argbuf.data());
    void* params[] = {w->jit_argbuf.data()};
    gx += j.ff_blocks;
    const double t0 = w->tl.on ? tl_now_us() : 0;
    if (w->dev_spawn) {
        // spawns decided on the device: the workgroups meet inside -- a COOPERATIVE launch (every workgroup resident, the runtime lets no second one in beside it)
        hipEvent_t a = nullptr, b = nullptr;
        if (w->prof) { a = w->prof_event(); b = w->prof_event(); if (!a || !b) return w->fail(GGRS_E_HIP, "hipEventCreate failed"); w->prof_bytes[GGRS_KERNEL_TICK] += bytes; HIPCHK(w, hipEventRecord(a, w->stream)); }
        HIPCHK(w, hipModuleLaunchCooperativeKernel(fn, gx, gy, gz, TPB, 1, 1, lds, w->stream, params));
        if (w->prof) { HIPCHK(w, hipEventRecord(b, w->stream)); w->prof_events.push_back({a, b, GGRS_KERNEL_TICK}); }
        if (done) HIPCHK(w, hipEventRecord(done, w->stream));
        w->len_stale = true;
    } else if (!w->prof) {
        if (done) HIPCHK(w, hipExtModuleLaunchKernel(fn, gx * TPB, gy, gz, TPB, 1, 1, lds, w->stream, params, nullptr, nullptr, done, 0));
        else HIPCHK(w, hipModuleLaunchKernel(fn, gx, gy, gz, TPB, 1, 1, lds, w->stream, params, nullptr));
    } else {
        hipEvent_t a = w->prof_event(), b = w->prof_event();
        if (!a || !b) return w->fail(GGRS_E_HIP, "hipEventCreate failed");
        w->prof_bytes[GGRS_KERNEL_TICK] += bytes;
        HIPCHK(w, hipExtModuleLaunchKernel(fn, gx * TPB, gy, gz, TPB, 1, 1, lds, w->stream, params, nullptr, a, b, 0));
        w->prof_events.push_back({a, b, GGRS_KERNEL_TICK});
    }
    if (w->tl.on) { w->tl.launch_us += tl_now_us() - t0; ++w->tl.n_launches; }
    if (w->ff_mark_id) { const uint64_t id = w->ff_mark_id; w->ff_mark_id = 0; if (w->knobs.spin_wait_us <= 0) return ff_mark_folded(w, id); }   // (polling sessions never wait for it: no marker packet per tick for them)
    return GGRS_OK;
}

// The kernel specialised for this group's shape, once the session has sent the shape often enough and the build is done (kernel_gen.hpp
// jit_specialise); nullptr: use the generic kernel.  Plain launches of every size qualify (depth-parallel roles are part of the shape); batches of
// checksum-only branches and groups with an eliminated Save do not.  Shapes are counted one by one (host_world.hpp JitSpecSlot): a SyncTest
// session has one, a P2P session one per rollback length; the row masks right after a spawn make a few more that never reach the threshold.
// members: the launch carries batch members with records (run_branch_step) -- where a Save lands and which rows it moves are per member, so only the op sequence,
// the load mask and the store policies are literals of that copy.
hipFunction_t jit_spec_for(ggrs_world* w, const GgrsJitArgs& j, bool members = false) {
    if (!w->knobs.jit_specialise_after || w->jit_src.empty() || !j.n_saves || !j.n_ops) return nullptr;
    if (w->dev_spawn) return nullptr;              // (a cooperative launch must be resident as a whole: the world was sized against the GENERAL kernel's occupancy at seal)
    for (uint32_t k = 0; k < j.n_saves && !members; ++k)
        if (!j.save_dst[k] || j.save_rows[k] != j.save_rows[0] || j.save_pmask[k] != j.save_pmask[0]) return nullptr;
    JitSig g; g.members = members ? 1u : 0u; g.vtags = j.vtags; g.op_bits = j.op_bits; g.save_rows = j.save_rows[0]; g.live_rows = j.live_rows; g.load_rows = j.load_rows; g.n_ops = j.n_ops; g.n_saves = j.n_saves;
    g.n_steps = j.n_steps; g.src_is_live = j.src_is_live; g.skip_live = j.skip_live; g.nt = j.nt; g.cached_saves = j.cached_saves; g.save_pmask = j.save_pmask[0]; g.live_pmask = j.live_pmask; g.dp_s = j.dp_s; g.nt_loads = j.nt_loads;
    auto building = [](const JitSpecSlot& s) { return s.spec && s.spec->state.load(std::memory_order_acquire) == 1; };
    JitSpecSlot* s = nullptr;
    for (auto& t : w->spec_tab) if (t.sig == g) { s = &t; break; }
    if (!s) {
        if (w->spec_tab.size() < (size_t)w->spec_shapes) { w->spec_tab.emplace_back(); s = &w->spec_tab.back(); }
        else {                                                                   // least recently used out: a shape without a kernel if there is one
            for (auto& t : w->spec_tab) {
                if (building(t)) continue;
                if (!s || (!t.spec && s->spec) || ((!t.spec) == (!s->spec) && t.last_use < s->last_use)) s = &t;
            }
            if (!s) return nullptr;
            jit_spec_drop(w, *s);
        }
        *s = JitSpecSlot{}; s->sig = g;
    }
    s->last_use = ++w->spec_clock;
    w->spec_last_slot = (int)(s - w->spec_tab.data());
    if (s->spec) return s->spec->state.load(std::memory_order_acquire) == 2 ? s->spec->fn : nullptr;
    if (++s->seen < (uint32_t)w->knobs.jit_specialise_after) return nullptr;
    if (w->spec_builds >= JIT_SPEC_MAX_BUILDS) return nullptr;
    for (auto& t : w->spec_tab) if (building(t)) return nullptr;                 // one build at a time; this shape asks again with its next group
    ++w->spec_builds;
    s->spec = new JitSpec; s->spec->sig = g;
    const std::string src = jit_specialise(w->jit_src, g);
    if (src.empty()) { s->spec->why = "the generated kernel's text could not be specialised"; s->spec->state.store(3, std::memory_order_release); return nullptr; }
    s->spec->state.store(1, std::memory_order_relaxed);
    if (w->knobs.jit_specialise_sync) jit_spec_build(s->spec, w->device, src, w->knobs.jit_cache_dir, w->knobs.aot_dir, w->knobs.no_hiprtc);
    else { s->spec->th = std::thread(jit_spec_build, s->spec, w->device, src, w->knobs.jit_cache_dir, w->knobs.aot_dir, w->knobs.no_hiprtc); jit_spec_worker_register(s->spec); }
    return s->spec->state.load(std::memory_order_acquire) == 2 ? s->spec->fn : nullptr;
}

inline ggrs_world::HostFold make_host_fold(const GgrsJitArgs& j, uint32_t res_slot, uint32_t rows, uint32_t n_cks, uint32_t members, uint64_t rows_off) {
    ggrs_world::HostFold f{}; f.res_slot = res_slot; f.n_saves = j.n_saves; f.g = rows; f.n_cks = n_cks; f.members = members; f.rows_off = rows_off;
    for (uint32_t k = 0; k < j.n_saves && k < (uint32_t)MAX_TICK_SAVES; ++k) f.save_len[k] = j.save_len[k];
    return f;
}
// res: the result slot of the group's first Save -- the pinned ring's cell and, when a consumer asked for one (ggrs_world::dev_results_dst), the device copy's
inline GenFinArgs make_gen_fin(const ggrs_world* w, const GgrsJitArgs& j, uint32_t rows, uint32_t n_cks, uint32_t res) {
    GenFinArgs f; memset(&f, 0, sizeof f);
    f.parts = reinterpret_cast<uint64_t*>(j.parts); f.part_stride = j.part_stride; f.n_parts = rows; f.n_cks = n_cks; f.n_saves = std::max(1u, j.n_saves);
    for (uint32_t k = 0; k < j.n_saves && k < (uint32_t)MAX_TICK_SAVES; ++k) f.save_len[k] = j.save_len[k];
    f.out = w->d_results + 2 * (uint64_t)res;
    if (w->dev_results_dst && res >= w->dev_results_first) f.out2 = w->dev_results_dst + 2 * (uint64_t)(res - w->dev_results_first);
    if (w->dev_spawn) f.dev_save_len = w->d_sp_len + 2;
    return f;
}

// A blocking call whose LAST GPU operation is this k_gen_finalize polls the tags the kernel leaves behind its results instead of asking the
// runtime for the stream (read_back): every later launch of the list disarms it again (launch_jit, the spawn systems)
inline void arm_spin(ggrs_world* w, GenFinArgs& f, uint32_t n_wgs, bool blocking) {
    w->spin_n = 0;
    if (!blocking || w->knobs.spin_wait_us <= 0 || !w->d_done || n_wgs > ggrs_world::SPIN_TAGS || w->prof || w->device_results_only) return;
    f.done = w->d_done; f.seq = ++w->spin_seq; w->spin_n = n_wgs;
}

// Identical checksum-only groups off the same source block (speculative branches: same ops, same frames, same length) are
// launched TOGETHER: one grid of tiles x K members (blockIdx.z) and one finalize of saves x K, instead of K launch pairs.
struct JitBatch {
    bool active = false; GgrsJitArgs j; uint32_t g = 0, k = 0, res_first = 0, n_cks = 0;
    void start(const GgrsJitArgs& j_, uint32_t g_, uint32_t res, uint32_t n_cks_) { active = true; j = j_; g = g_; k = 1; res_first = res; n_cks = n_cks_; }
    bool try_add(const ggrs_world* w, const GgrsJitArgs& b, uint32_t g_, uint32_t res) {
        if (!active || g_ != g || b.src != j.src || b.len != j.len || b.op_bits != j.op_bits || b.n_ops != j.n_ops || b.n_saves != j.n_saves ||
            b.n_steps != j.n_steps || b.load_rows != j.load_rows || memcmp(b.dt_bits, j.dt_bits, sizeof b.dt_bits) != 0 || memcmp(b.aux_bits, j.aux_bits, sizeof b.aux_bits) != 0 ||
            memcmp(b.step_frame, j.step_frame, sizeof b.step_frame) != 0 || memcmp(b.step_confirmed, j.step_confirmed, sizeof b.step_confirmed) != 0 ||
            memcmp(b.step_flags, j.step_flags, sizeof b.step_flags) != 0) return false;
        if (w->jit_reads_inputs) {
            if (memcmp(b.n_inputs, j.n_inputs, sizeof b.n_inputs) != 0) return false;
            const size_t row = (size_t)w->max_players * (w->input_bytes + 1);            // what pack_inputs wrote of every step's row
            for (uint32_t q = 0; q < j.n_steps; ++q) if (memcmp(b.inputs[q], j.inputs[q], row) != 0) return false;
        }
        // fused spawns: the same rows from the same staged payload at the same steps (one request list stages a payload it is handed twice
        // -- the same host arrays: every branch that spawns in frame f -- only once, so identical spawning branches share pointers)
        if (memcmp(b.spawn_count, j.spawn_count, sizeof b.spawn_count) != 0 || memcmp(b.save_len, j.save_len, sizeof b.save_len) != 0 ||
            memcmp(b.spawn_first, j.spawn_first, sizeof b.spawn_first) != 0 || memcmp(b.spawn_payload, j.spawn_payload, sizeof b.spawn_payload) != 0) return false;
        if ((k + 1) * j.n_saves > w->gen_parts_saves || res != res_first + k * j.n_saves) return false;
        ++k;
        return true;
    }
    bool blocking = false;                                           // the synchronous API: see host_fold_rows
    int flush(ggrs_world* w) {
        if (!active) return GGRS_OK;
        active = false;
        w->batch_ev_attached = false;                                // this launch comes after whatever carried the batch event
        bool host_fold = false; uint64_t rows_off = 0;
        {
            if (k > 1) j.dp_s = 0;
            host_fold = host_fold_rows(w, g, j.n_saves, n_cks, k, &rows_off, blocking);
            if (host_fold) { j.parts = reinterpret_cast<ggrs_u64*>(w->d_rows + rows_off); j.part_stride = g; }
            { const int lrc = launch_jit(w, w->jit_fn, jit_grid(g), j.dp_s ? (j.n_saves + j.dp_s) / j.dp_s : 1u, k, jit_lane_fold_bytes(w, w->cks_args.n_cks, j.n_saves), j,
                                         rows_bytes_per_slot(w, j.load_rows, !j.src_is_live) * j.len * k); if (lrc) return lrc; }
        }
        if (host_fold) { w->folds.push_back(make_host_fold(j, res_first, g, n_cks, k, rows_off)); return GGRS_OK; }
        GenFinArgs f = make_gen_fin(w, j, g, n_cks, res_first);   // one row per workgroup
        arm_spin(w, f, j.n_saves * k, blocking);
        {
            ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
            hipLaunchKernelGGL(k_gen_finalize, dim3(j.n_saves * k), dim3(FIN_TPB), 0, w->stream, f);
        }
        HIPCHK(w, hipGetLastError());
        return GGRS_OK;
    }
};

// ---- lazy live block -------------------------------------------------------------------------------------------------------
// A SyncTest tick ends  [.., Save(F), Advance]  and the next one opens with LoadGameState(F + 1 - d): the live block the tick would write
// (frame F + 1: 32 of the stress_test's 320 B per entity-tick) is never read -- the next group's source is a ring slot.  A session whose lists keep
// opening with a Load (LAZY_LIVE_STREAK in a row) therefore stops writing it at the end of a list: the live world is then DEFINED as
// Advance(slot of F) with the recorded inputs, and whoever needs its bytes -- a list that opens without a Load, a download, a spawn, a despawn,
// an upload, a handed-out column pointer, the fan-out -- gets them from ONE launch first (materialise_live: load the slot, one step, store live).
// Only for HBM-sized worlds (bytes are what bounds them), worlds without live-only state (markers, non-rollback components) and without
// externally held column pointers.  Snapshots, checksums, ring and frame counters are untouched: every Save of the tick was made.
constexpr uint32_t LAZY_LIVE_STREAK = 8;
inline bool lazy_live_allowed(const ggrs_world* w) {                 // (what a layout-only world -- `make aot` on a machine without a GPU -- can tell)
    if (!w->lazy_live_on || w->live_handed_out || w->has_nr || w->marks_possible || w->device_results_only || jit_dev_spawn(w)) return false;
    for (uint8_t e : w->col_ext) if (e) return false;
    return true;
}
inline bool lazy_live_possible(const ggrs_world* w) { return w->gen_ok && !w->jit_marks && lazy_live_allowed(w); }
int materialise_live(ggrs_world* w) {
    ggrs_world::LiveStale& st = w->live_stale;
    if (!st.valid) return GGRS_OK;
    GgrsJitArgs j; memset(&j, 0, offsetof(GgrsJitArgs, inputs));
    j.src = st.src->ptr; j.live = w->live.ptr; j.len = st.len; j.src_is_live = 0;
    j.n_ops = 1; j.op_bits = 1; j.n_steps = 1;
    j.dt_bits[0] = st.dt_bits; j.aux_bits[0] = st.aux_bits; j.step_frame[0] = st.step_frame; j.step_confirmed[0] = st.step_confirmed; j.n_inputs[0] = st.n_inputs;
    memcpy(j.inputs[0], st.inputs, sizeof st.inputs);
    j.live_rows = rows_to_store(w, w->live); j.live_pmask = pmask_differs(w, w->live, w->cur_ver);
    j.load_rows = jit_static_reads(w) | j.live_rows;
    j.vtags = w->vtags ? 1u : 0u;
    if (j.vtags) { const int trc = vtags_reserve(w, 1, 1, &j.tag_base); if (trc) return trc; j.src_tagok = block_tagok(w, *st.src); j.live_tagok = block_tagok(w, w->live); }
    const uint64_t cover = std::max(std::max(st.src->dirty_len, w->live.dirty_len), w->len);
    j.parts = reinterpret_cast<ggrs_u64*>(w->d_gen_parts); j.part_stride = w->gen_part_stride; j.part_tstride = 1;
    j.n_units = std::max<uint32_t>(1, (uint32_t)((cover + 63) / 64));
    const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + 255) / 256));
    j.nt = 0;                                                        // (the live block is read again by whoever asked for it)
    j.nt_loads = cover > JIT_NT_MIN_SLOTS ? 1u : 0u;                  // the slot was stored around the caches one launch ago
    const int rc = launch_jit(w, w->jit_fn, jit_grid(g), 1, 1, 0, j, (rows_bytes_per_slot(w, j.load_rows, true) + rows_bytes_per_slot(w, j.live_rows, false)) * w->len);
    if (rc) return rc;
    w->live.dirty_len = std::max(st.src->dirty_len, w->len);
    ver_sync_live(w);
    if (w->vtags) w->live.tag_ok |= j.live_rows & w->tag_cols; else w->live.tag_ok &= ~j.live_rows;
    st.valid = false; ++w->lazy_materialised;
    return GGRS_OK;
}

int run_request_groups_gen(ggrs_world* w, const ggrs_request* reqs, uint32_t n, uint64_t* checksums_out,
                           uint32_t res_base = 0, bool wait = true, uint32_t* n_saves_out = nullptr) {
    uint32_t i = 0, ns = 0;
    int rc = GGRS_OK;
    JitBatch batch; batch.blocking = wait;
    w->spin_n = 0;
    // payloads this list has staged already (by the caller's host pointer): every branch of a fan-out step that spawns in frame f hands in the same
    // arrays.  Offsets into the ring die with the ring's generation (a reset: the ring was full and the stream was waited for; a result page was read back)
    struct Staged { const void* host; const void* host2; uint64_t bytes; const unsigned char* dev; };
    std::vector<Staged> staged; uint64_t staged_gen = w->stage_gen;
    const uint32_t n_cks = w->cks_args.n_cks;
    const uint64_t static_reads = jit_static_reads(w);
    if (n) {
        // a list that opens with a LoadGameState replaces the live world: a lazily skipped live block needs no bytes; any other list reads it first
        if (reqs[0].kind == GGRS_REQ_LOAD) ++w->load_open_streak; else { w->load_open_streak = 0; rc = materialise_live(w); if (rc) return rc; }
    }
    while (i < n) {
        w->batch_ev_attached = false;                                   // only the list's LAST launch may carry the batch event
        GgrsJitArgs j; memset(&j, 0, offsetof(GgrsJitArgs, inputs));
        GroupState gs;
        const ggrs_request* spawn_req = nullptr;
        rc = group_open(w, reqs, i, gs); if (rc) return rc;
        j.src_is_live = gs.src_is_live;
        const uint64_t len_start = w->len;                              // the source block's len: a fused spawn grows w->len while the group is assembled
        while (i < n && j.n_ops < (uint32_t)MAX_TICK_OPS) {
            const ggrs_request& r = reqs[i];
            if (r.kind == GGRS_REQ_LOAD) break;
            if (r.kind == GGRS_REQ_SAVE) {
                if (j.n_saves == w->cap_saves || (wait && ns + j.n_saves == w->max_results)) break;
                rc = group_save(w, gs, j.n_saves, j.save_dst, j.save_frame); if (rc) return rc;
                j.save_len[j.n_saves] = w->len;
                ++j.n_ops; ++j.n_saves;
            } else if (r.kind == GGRS_REQ_ADVANCE) {
                if (j.n_steps == w->cap_steps) break;
                // A spawn system fires in this frame.  Fused (kernel_gen.hpp): the step's launch appends the rows itself -- the payload is staged
                // now (pinned, device-mapped ring: no copy command per spawning step), the host does in request order what run_spawn_systems does
                // around its kernel (capacity, versions of the bundle, len) and the group goes on.  Worlds whose spawn system the generator cannot
                // fuse end the group after the step and run the spawn as its own launches, as Bevy's Commands flush ends the schedule.
                const bool spawns = advance_spawns(w, r);
                const bool fused_spawn = spawns && w->jit_spawn_sys >= 0;
                const unsigned char* payload_dev = nullptr;
                if (fused_spawn) {
                    const ggrs_system_desc& sd = w->systems[w->jit_spawn_sys];
                    const bool custom = sd.kind == GGRS_SYS_SPAWN_CUSTOM;
                    if (r.spawn_count > 0xFFFFFFFFull || w->len + r.spawn_count > w->capacity)
                        return w->fail(GGRS_E_CAPACITY, "spawn of %llu exceeds capacity %llu", (unsigned long long)r.spawn_count, (unsigned long long)w->capacity);
                    // what the spawner reads: particles -- count f32 of vx, then count f32 of vy; user-written -- the request's payload blob
                    const ggrs_world::SpawnSys* sp = custom ? &w->spawn_customs[sd.comp[0]] : nullptr;
                    const uint64_t pbytes = custom ? (sp->payload_stride ? (uint64_t)sp->payload_stride * r.spawn_count : r.spawn_payload_bytes) : 8 * r.spawn_count;
                    const void* key = custom ? r.spawn_payload : (const void*)r.spawn_vx;
                    const void* key2 = custom ? nullptr : (const void*)r.spawn_vy;
                    if (staged_gen != w->stage_gen) { staged.clear(); staged_gen = w->stage_gen; }
                    const Staged* hit = nullptr;
                    if (pbytes) for (auto& st : staged) if (st.host == key && st.host2 == key2 && st.bytes == pbytes) { hit = &st; break; }
                    if (hit) payload_dev = hit->dev;
                    else if (pbytes) {
                        uint64_t soff = 0;
                        if (!stage_ring_alloc(w, pbytes, &soff)) {
                            // the ring is full of payloads that launches already queued -- or the steps of THIS group -- still have to read
                            if (pbytes > w->stage_bytes) return w->fail(GGRS_E_CAPACITY, "spawn payload of %llu bytes exceeds the staging buffer (%llu bytes: GGRS_STAGE_BYTES)", (unsigned long long)pbytes, (unsigned long long)w->stage_bytes);
                            if (j.n_ops) break;                              // the group ends BEFORE this frame: it is launched, then the frame opens the next group
                            rc = batch.flush(w); if (rc) return rc;
                            HIPCHK(w, hipStreamSynchronize(w->stream));      // let everything queued run, then start the ring over
                            stage_ring_reset(w); staged.clear(); staged_gen = w->stage_gen;
                            if (!stage_ring_alloc(w, pbytes, &soff)) return w->fail(GGRS_E_CAPACITY, "spawn payload of %llu bytes does not fit the staging buffer", (unsigned long long)pbytes);
                        }
                        // (the caller's arrays are free again when the call returns; the bytes sit in the ring until this list's batch is collected)
                        if (custom) memcpy(w->h_stage + soff, r.spawn_payload, pbytes);
                        else { memcpy(w->h_stage + soff, r.spawn_vx, r.spawn_count * 4); memcpy(w->h_stage + soff + r.spawn_count * 4, r.spawn_vy, r.spawn_count * 4); }
                        payload_dev = w->d_hstage + soff;
                        if (staged.size() < 256) staged.push_back({key, key2, pbytes, payload_dev});
                    }
                }
                uint32_t dtb = 0;
                rc = group_step(w, r, &dtb, w->jit_marks ? &j.step_flags[j.n_steps] : nullptr); if (rc) return rc;
                if (w->dev_spawn) {                                            // any frame may append rows of the bundle: its columns and presence masks are new after every step
                    const ggrs_world::SpawnSys& spd = w->spawn_customs[w->systems[w->jit_spawn_sys].comp[0]];
                    for (uint32_t c = 0; c < w->comps.size(); ++c) if ((spd.bundle_mask >> c) & 1ull) ver_touch_comp(w, c);
                }
                j.dt_bits[j.n_steps] = dtb;
                j.step_frame[j.n_steps] = w->frame; j.step_confirmed[j.n_steps] = w->confirmed;
                if (w->jit_box_sys >= 0) {                                     // FRICTION.powf(dt), platform libm (box_game.rs:189-195)
                    float dtf; memcpy(&dtf, &dtb, 4);
                    const float fp = powf(w->systems[w->jit_box_sys].fparam[2], dtf);
                    memcpy(&j.aux_bits[j.n_steps], &fp, 4);
                }
                j.n_inputs[j.n_steps] = (uint8_t)std::min<uint32_t>(r.n_inputs, w->max_players);
                if (w->jit_reads_inputs) pack_inputs(w, r, j.inputs[j.n_steps]);      // PlayerInputs<T>: (T::Input, InputStatus) per player (src/lib.rs:98)
                const uint32_t step = j.n_steps;
                ++j.n_steps;
                j.op_bits |= 1ULL << j.n_ops; ++j.n_ops;
                if (fused_spawn) {
                    const ggrs_system_desc& sd = w->systems[w->jit_spawn_sys];
                    j.spawn_payload[step] = payload_dev;
                    j.spawn_first[step] = w->len; j.spawn_count[step] = (uint32_t)r.spawn_count;
                    // new rows in every column (and the presence mask) of the bundle
                    if (sd.kind == GGRS_SYS_SPAWN_CUSTOM) { const ggrs_world::SpawnSys& sp = w->spawn_customs[sd.comp[0]]; for (uint32_t c = 0; c < w->comps.size(); ++c) if ((sp.bundle_mask >> c) & 1ull) ver_touch_comp(w, c); }
                    else { ver_touch_comp(w, sd.comp[0]); ver_touch_comp(w, sd.comp[1]); ver_touch_comp(w, sd.comp[2]); }
                    w->len += r.spawn_count;
                } else if (spawns) { spawn_req = &r; ++i; break; }
            } else {
                return w->fail(GGRS_E_INVALID, "unknown request kind %u", r.kind);
            }
            ++i;
        }
        const bool dead = !w->dev_spawn && group_is_dead(w, reqs, i, n, j.save_frame, j.n_saves, spawn_req != nullptr);
        if (dead) { for (uint32_t k = 0; k < j.n_saves; ++k) j.save_dst[k] = nullptr; j.skip_live = 1; }
        const uint64_t cover = w->dev_spawn ? w->capacity : std::max(gs.cover, w->len);      // (device-decided spawns: the host only knows a bound of len)
        // lazy live block: the LAST group of a list that ends [.., Save(F), Advance] in a session whose lists keep opening with a Load
        if (!dead && !spawn_req && i >= n && ((w->load_open_streak >= LAZY_LIVE_STREAK && cover > JIT_NT_MIN_SLOTS) || w->lazy_live_on == 2) && j.n_ops >= 2 && j.n_saves && j.n_steps &&
            ((j.op_bits >> (j.n_ops - 1)) & 1ull) && !((j.op_bits >> (j.n_ops - 2)) & 1ull) && gs.dsts[j.n_saves - 1] && !j.spawn_count[j.n_steps - 1] &&
            gs.dsts[j.n_saves - 1] != gs.src && lazy_live_possible(w)) {
            j.skip_live = 1;
            ggrs_world::LiveStale& st = w->live_stale;
            const uint32_t q = j.n_steps - 1;
            st.valid = true; st.src = gs.dsts[j.n_saves - 1]; st.len = j.save_len[j.n_saves - 1];
            st.dt_bits = j.dt_bits[q]; st.aux_bits = j.aux_bits[q]; st.step_frame = j.step_frame[q]; st.step_confirmed = j.step_confirmed[q]; st.n_inputs = j.n_inputs[q];
            if (w->jit_reads_inputs) memcpy(st.inputs, j.inputs[q], sizeof st.inputs); else memset(st.inputs, 0, sizeof st.inputs);
            ++w->lazy_skips;
        }
        const bool wrote_live = (!j.src_is_live || j.n_steps) && !j.skip_live;
        // ---- row versions -> store masks; what must be in registers = everything stored + everything a step or checksum reads
        uint64_t bytes_slot = 0;
        j.load_rows = j.n_ops ? static_reads : 0;
        for (uint32_t k = 0; k < j.n_saves; ++k) {
            j.save_rows[k] = j.save_dst[k] ? gs.save_rows[k] : 0;
            j.save_pmask[k] = j.save_dst[k] ? gs.save_pmask[k] : 0;
            j.load_rows |= j.save_rows[k];
            bytes_slot += rows_bytes_per_slot(w, j.save_rows[k], true);
        }
        if (wrote_live) { j.live_rows = rows_to_store(w, w->live); j.live_pmask = pmask_differs(w, w->live, w->cur_ver); j.load_rows |= j.live_rows; bytes_slot += rows_bytes_per_slot(w, j.live_rows, false); }
        gs.live_rows = j.live_rows;
        // value tags: a group that stores nothing (a dead branch) has no use for them
        j.vtags = (w->vtags && !dead) ? 1u : 0u;
        if (j.vtags) {
            rc = vtags_reserve(w, j.n_steps, 1, &j.tag_base); if (rc) return rc;
            j.src_tagok = block_tagok(w, *gs.src); j.live_tagok = wrote_live ? block_tagok(w, w->live) : 0;
            for (uint32_t k = 0; k < j.n_saves; ++k) j.save_tagok[k] = (j.save_dst[k] && gs.dsts[k]) ? block_tagok(w, *gs.dsts[k]) : 0;
        }
        bytes_slot += rows_bytes_per_slot(w, j.load_rows, !j.src_is_live);
        j.src = gs.src->ptr; j.live = w->live.ptr; j.len = len_start;
        if (w->dev_spawn) {
            if (w->sp_epoch > 0xF0000000u) {         // epochs never repeat: long before the 32-bit counter wraps (~65 M launches) the mailboxes start over, in stream order
                HIPCHK(w, hipMemsetAsync(w->d_sp_sums, 0, (3 * (size_t)w->sp_tiles + 32) * 8, w->stream)); w->sp_epoch = 0;
            }
            j.sp_sums = reinterpret_cast<ggrs_u64*>(w->d_sp_sums); j.sp_epoch = w->sp_epoch; w->sp_epoch += 2u * MAX_TICK_STEPS + 2u;
            j.sp_prec = w->d_sp_prec; j.sp_link = reinterpret_cast<ggrs_u64*>(w->d_sp_link);
            j.sp_len = reinterpret_cast<ggrs_u64*>(w->d_sp_len); j.sp_cap = w->capacity; j.sp_tiles = w->sp_tiles;
        }
        j.parts = reinterpret_cast<ggrs_u64*>(w->d_gen_parts); j.part_stride = w->gen_part_stride; j.part_tstride = 1;
        j.n_units = std::max<uint32_t>(1, (uint32_t)((cover + 63) / 64));
        const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + 255) / 256));
        j.nt = (w->nt_copy || cover > JIT_NT_MIN_SLOTS) ? 1u : 0u;
        // A rollback group's FIRST Save is the oldest frame it produces -- what the next rollback loads (SyncTest: always; P2P with a steady
        // rollback depth: likewise).  Storing it through the L2 instead of around it lets the next launch's loads hit there: 72.5 -> 64.6 us
        // per depth-8 tick at 1 M in the ring-walking harness (profiles/r03n), every other row still streams past the caches.
        // Past ~2.5 M particles the rows no longer survive in the caches until the next launch and only displace the stream (4 M: +3 %; r04c).
        j.cached_saves = (j.nt && !j.src_is_live && j.n_saves >= 2 && rows_bytes_per_slot(w, j.save_rows[0], true) * cover <= JIT_CACHED_SAVE_MAX_BYTES) ? 1u : 0u;
        // the source block of an HBM-sized rollback group is in the caches only if the previous group kept a Save there (the steady session: the
        // same decision as this group's): otherwise its lines come from HBM once and are dead after the load (4 M: -3.5 %, allhot 4 M -8 %; it costs
        // 3 % where the loads DO hit: profiles/r04o)
        j.nt_loads = (j.nt && !j.cached_saves && !j.src_is_live) ? 1u : 0u;
        const bool launch = j.n_ops || !j.src_is_live;
        {
            // ---- per-tile grid: 256-slot workgroups, depth-parallel roles, batches; the partial rows go to the host, forward, or to k_gen_finalize
            // Depth-parallel roles: the group's outputs (Saves + live world) are split over grid.y roles of dp_s outputs.  Every role
            // reads the source block while the others write theirs, so the source must be none of the destinations; below ~2 Saves
            // there is no chain to split.  Crossovers: profiles/r02dp/ab2.txt, profiles/r02jit/jit_dp.txt.
            if (j.n_saves >= 2 && !w->jit_marks && !w->dev_spawn) {
                bool ok = !(wrote_live && j.src == j.live);
                for (uint32_t k = 0; k < j.n_saves; ++k) ok = ok && j.save_dst[k] != j.src;
                // ... and the destinations pairwise distinct: a ring shallower than the group's Saves hands an evicted slot to a later Save, and two
                // roles writing one block concurrently could leave the OLDER frame's rows there (in op order on one lane the newer one wins)
                for (uint32_t k = 1; k < j.n_saves && ok; ++k) for (uint32_t q = 0; q < k; ++q) ok = ok && (!j.save_dst[k] || j.save_dst[k] != j.save_dst[q]);
                const uint64_t m = JIT_DP_MAX_SLOTS;
                if (ok) j.dp_s = cover <= m ? 1u : (cover <= 2 * m ? 2u : (cover <= 6 * m ? 3u : 0u));
            }
            // identical checksum-only groups (speculative branches) ride in one launch; a batch already fills the chip, so no roles
            const bool batchable = dead && j.n_saves > 0 && !w->jit_marks && cover <= JIT_BATCH_MAX_SLOTS;
            if (batchable && batch.active) {
                GgrsJitArgs jb = j; jb.dp_s = 0;
                if (batch.try_add(w, jb, g, res_base + ns)) { batch.j.dp_s = 0; group_close(w, gs, j.n_saves, dead, wrote_live); ns += j.n_saves; goto group_done; }
            }
            rc = batch.flush(w); if (rc) return rc;
            if (batchable) { batch.start(j, g, res_base + ns, n_cks); group_close(w, gs, j.n_saves, dead, wrote_live); ns += j.n_saves; goto group_done; }
            // who folds this launch's partial rows: the next launch on the stream (fold-forward: enqueued lists of more than
            // GGRS_FOLD_FORWARD_MIN_WGS workgroups), the launch itself (self-fold: blocking calls of that size), the host from pinned rows (smaller groups), or k_gen_finalize (blocking calls of more than 1024
            // workgroups, results that stay on the device, no room in the pinned ring)
            uint64_t rows_off = 0;
            const uint32_t ff_split = (g + FF_CHUNK - 1u) / FF_CHUNK;                      // chunks of <= 1024 entries per row: one fold-forward workgroup each
            const uint32_t nvals = j.n_saves * (n_cks + 1) * ff_split;
            bool ff = launch && j.n_saves && !wait && !w->device_results_only && !w->dev_spawn && g > (uint32_t)w->knobs.fold_forward_min_wgs && w->d_ff_rows[0] &&
                      rows_ring_alloc(w, 2ull * nvals, &rows_off);
            // a BLOCKING call of that size: the launch folds its own rows (self-fold: its fold workgroups read the tile workgroups' tagged cells as they arrive) -- no k_gen_finalize
            // between the kernel and the caller.  Not beside a pending fold-forward (the role folds one set of rows), not with depth-parallel roles (plain grids only),
            // and only while the fold workgroups -- resident and waiting from the launch's start -- are few beside the 2048 the device holds (8 M entities and more
            // keep k_gen_finalize: thousands of waiting workgroups would leave the tiles no room, at 32 M none at all)
            const bool ffs = !ff && launch && j.n_saves && wait && nvals <= SELF_FOLD_MAX_WGS && !w->device_results_only && !w->dev_spawn && !j.dp_s && !w->ff_pending.valid &&
                             g > (uint32_t)w->knobs.fold_forward_min_wgs && w->d_ff_rows[0] && !spawn_req && rows_ring_alloc(w, 2ull * nvals, &rows_off);
            if (ffs) ff = true;
            const bool host_fold = !ff && launch && host_fold_rows(w, g, j.n_saves, n_cks, 1, &rows_off, wait);
            uint32_t ff_buf = 0;
            const uint32_t rows_n = j.n_saves * (n_cks + 1);
            // fold-forward rows are TILE-major: a workgroup's values of all its Saves and parts sit side by side (one coalesced store of 192 B for the
            // stress_test; row-major they were 24 eight-byte stores into 24 different lines, and the 1 M launch took 51 us instead of 48.9: profiles/r05c)
            if (ff && !ffs) { ff_buf = w->ff_cur; w->ff_cur ^= 1u; j.parts = reinterpret_cast<ggrs_u64*>(w->d_ff_rows[ff_buf]); j.part_stride = 1; j.part_tstride = rows_n; }
            // (self-fold: 16-byte cells {value, tag} instead of 8-byte entries -- the two fold-forward buffers, which lie side by side and hold nothing pending, as one)
            if (ffs) { ff_buf = 0; j.parts = reinterpret_cast<ggrs_u64*>(w->d_ff_rows[0]); j.part_stride = 1; j.part_tstride = rows_n; }
            uint64_t ffs_id = 0, ffs_seq = 0;
            if (ffs) {
                ffs_id = w->ff_next_id++; ffs_seq = (0xA5ull << 56) | ++w->ff_seq;
                memset(w->h_rows + rows_off, 0, (size_t)nvals * 16);
                j.ff_rows = reinterpret_cast<const ggrs_u64*>(w->d_ff_rows[ff_buf]); j.ff_out = reinterpret_cast<ggrs_u64*>(w->d_rows + rows_off); j.ff_seq = ffs_seq;
                j.ff_nvals = nvals; j.ff_blocks = (nvals + 7u) & ~7u; j.ff_g = g; j.ff_stride = 1; j.ff_istride = rows_n; j.ff_split = ff_split;
                j.ff_self = 1;
                w->ff_done_id = ffs_id; w->ff_mark_id = ffs_id;       // the fold is on the stream with this very launch; launch_jit records its event behind it
            }
            if (host_fold) { j.parts = reinterpret_cast<ggrs_u64*>(w->d_rows + rows_off); j.part_stride = g; }
            if (launch) {
                hipFunction_t fn = jit_spec_for(w, j);
                if (!fn) fn = w->jit_fn;
                // nothing is queued behind this kernel when its rows are folded later (or there is nothing to fold) and no spawn system follows:
                // the batch event of an enqueued list then completes WITH it (no marker packet between this tick's kernel and the next one's)
                const bool last_gpu_op = (host_fold || ff || !j.n_saves) && !spawn_req && !w->prof;
                hipEvent_t done = last_gpu_op ? w->batch_ev : nullptr;
                rc = launch_jit(w, fn, jit_grid(g), j.dp_s ? (j.n_saves + j.dp_s) / j.dp_s : 1u, 1, jit_lane_fold_bytes(w, n_cks, j.n_saves), j, bytes_slot * w->len, done); if (rc) return rc;
                w->batch_ev_attached = done != nullptr;
            }
            group_close(w, gs, j.n_saves, dead, wrote_live);
            if (ff) {
                ggrs_world::HostFold f = make_host_fold(j, res_base + ns, ff_split, n_cks, 1u, rows_off);   // (the host XORs / adds the row's chunks)
                if (ffs) { f.ff_id = ffs_id; f.ff_seq = ffs_seq; w->folds.push_back(f); ns += j.n_saves; }
                else {
                f.ff_id = w->ff_next_id++; f.ff_seq = (0xA5ull << 56) | ++w->ff_seq;      // (a tag no live count and -- but for 2^-64 -- no hash equals)
                memset(w->h_rows + rows_off, 0, (size_t)nvals * 16);                    // the {value, tag} cells: whatever an earlier fold left there is gone
                w->folds.push_back(f);
                ggrs_world::FfPending& p = w->ff_pending;
                p.valid = true; p.id = f.ff_id; p.seq = f.ff_seq; p.buf = ff_buf; p.nvals = nvals; p.g = g; p.stride = 1; p.istride = rows_n; p.split = ff_split; p.out_off = rows_off;
                ns += j.n_saves;
                }
            } else if (host_fold) { w->folds.push_back(make_host_fold(j, res_base + ns, g, n_cks, 1u, rows_off)); ns += j.n_saves; }
            else if (j.n_saves) {
                GenFinArgs f = make_gen_fin(w, j, g, n_cks, res_base + ns);   // one row per workgroup
                arm_spin(w, f, j.n_saves, wait);
                {
                    ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
                    hipLaunchKernelGGL(k_gen_finalize, dim3(j.n_saves), dim3(FIN_TPB), 0, w->stream, f);
                }
                HIPCHK(w, hipGetLastError());
                ns += j.n_saves;
            }
        }
        group_done:
        if (spawn_req) {
            rc = batch.flush(w); if (rc) return rc;
            w->spin_n = 0;
            rc = run_spawn_systems(w, spawn_req->inputs, spawn_req->n_inputs, spawn_req->spawn_count, spawn_req->spawn_vx, spawn_req->spawn_vy);
            if (rc) return rc;
        }
        if (wait && ns == w->max_results) {
            rc = batch.flush(w); if (rc) return rc;
            rc = read_back(w, ns, checksums_out); if (rc) return rc;
            checksums_out += 2 * (uint64_t)ns; ns = 0;
        }
    }
    rc = batch.flush(w); if (rc) return rc;
    if (n_saves_out) *n_saves_out = ns;
    if (!wait) return GGRS_OK;
    return read_back(w, ns, checksums_out);
}

// ---------------------------------------------------------------------------------------------------------------------
// BRANCH STEPS (ggrs_hip_fanout_step_branches).  B speculative branches off ONE snapshot, each  (AdvanceFrame, SaveGameState) x n_frames  with its own
// PlayerInputs and spawns, in ONE launch of the generated kernel: gridDim.z = B, member z's differences in a record in device memory (kernel_gen.hpp
// JitLayout::Member).  Branches are speculation, not history: nothing here touches the ring, the frame counters or the live block.  What a branch produces is its
// Checksum(u128)s and -- when asked to be kept -- its frames in blocks of their own (ggrs_world::spec_blocks), under row versions like every other block.
// ---------------------------------------------------------------------------------------------------------------------
struct BranchKeep {          // what the last branch step retained: output o of branch b (frame base_frame + 1 + o) lives in spec_blocks[blk[b * n_out + o]] (-1: not kept)
    bool valid = false; uint32_t n_branches = 0, n_out = 0, n_frames = 0; int32_t base_frame = 0;
    std::vector<int> blk;
};
constexpr uint32_t BRANCH_MAX = 4096;
// spec_blocks[0 .. n): allocated in chunks, header + masks zeroed (the invariant every block keeps: mask words beyond its dirty_len are zero)
int spec_blocks_reserve(ggrs_world* w, size_t n) {
    const uint64_t head = ALIGN + (uint64_t)w->plan.n_masks * align_up(w->cap_pad / 8, ALIGN);
    while (w->spec_blocks.size() < n) {
        const size_t want = n - w->spec_blocks.size();
        const size_t chunk = std::max<size_t>(1, std::min<size_t>(want, (1ull << 30) / std::max<uint64_t>(w->state_bytes, 1)));
        uint8_t* p = nullptr;
        if (hipMalloc((void**)&p, chunk * w->state_bytes) != hipSuccess) { (void)hipGetLastError(); return w->fail(GGRS_E_HIP, "hipMalloc of %zu branch state blocks (%llu bytes each) failed", chunk, (unsigned long long)w->state_bytes); }
        w->spec_allocs.push_back(p);
        for (size_t k = 0; k < chunk; ++k) {
            Block b; b.ptr = p + k * w->state_bytes; b.ver.assign(w->cur_ver.size(), VER_NONE);
            if (w->knobs.debug_poison) HIPCHK(w, hipMemsetAsync(b.ptr, 0xA5, w->state_bytes, w->stream));
            HIPCHK(w, hipMemsetAsync(b.ptr, 0, head, w->stream));
            HIPCHK(w, hipMemsetAsync(b.ptr + w->off_tags, 0, w->state_bytes - w->off_tags, w->stream));      // value tags: 0 = no identity
            w->spec_blocks.push_back(std::move(b));
        }
    }
    return GGRS_OK;
}
// everything that can be checked before the prefix runs
int validate_branch_step(ggrs_world* w, const ggrs_branch_step& st) {
    if (!w->gen_ok) return w->fail(GGRS_E_INVALID, "branch steps need the generated request-group kernel, which this world does not have: %s", w->jit_status.c_str());
    if (w->dev_spawn) return w->fail(GGRS_E_INVALID, "branch steps are not available for worlds whose systems spawn on the device (every launch is one cooperative grid): use ggrs_hip_fanout_step");
    if (w->jit_marks || w->has_nr || w->marks_possible) return w->fail(GGRS_E_INVALID, "branch steps are not available for worlds with live-only state (RollbackDespawned markers -- a system that can call despawn_rollback() --, non-rollback components): use ggrs_hip_fanout_step");
    if (st.n_branches == 0 || st.n_branches > BRANCH_MAX) return w->fail(GGRS_E_INVALID, "a branch step holds 1..%u branches, not %u", BRANCH_MAX, st.n_branches);
    const uint32_t S = (st.flags & GGRS_BRANCH_SAVE_LAST) ? st.n_frames : st.n_frames - 1;
    if (st.n_frames == 0 || st.n_frames > w->cap_steps || S > w->cap_saves) return w->fail(GGRS_E_INVALID, "a branch covers 1..%u frames (%u SaveGameStates) in this world, not %u", w->cap_steps, w->cap_saves, st.n_frames);
    if (st.flags & ~(GGRS_BRANCH_SAVE_LAST | GGRS_BRANCH_RETAIN_NEWEST | GGRS_BRANCH_RETAIN_ALL)) return w->fail(GGRS_E_INVALID, "unknown branch step flags %x", st.flags);
    // the state after the last AdvanceFrame would be kept in the LIVE form (the component itself), a snapshot holds Strategy::Stored: such a block could not become a ring slot
    if (w->has_strategy && (st.flags & (GGRS_BRANCH_RETAIN_NEWEST | GGRS_BRANCH_RETAIN_ALL)) && !(st.flags & GGRS_BRANCH_SAVE_LAST))
        return w->fail(GGRS_E_INVALID, "a world with a component under a Strategy keeps branch states only as snapshots: add GGRS_BRANCH_SAVE_LAST");
    if (st.n_inputs > w->max_players) return w->fail(GGRS_E_INVALID, "%u player inputs (at most %u: ggrs_hip_set_input_layout)", st.n_inputs, w->max_players);
    if (st.n_inputs && !st.inputs) return w->fail(GGRS_E_INVALID, "n_inputs = %u but inputs is NULL", st.n_inputs);
    if (st.status) for (uint64_t k = 0; k < (uint64_t)st.n_branches * st.n_frames * st.n_inputs; ++k) if (st.status[k] > GGRS_INPUT_DISCONNECTED) return w->fail(GGRS_E_INVALID, "InputStatus %u is none of Confirmed / Predicted / Disconnected", st.status[k]);
    if (st.spawn_sel) {
        if (w->jit_spawn_sys < 0) { for (uint64_t k = 0; k < (uint64_t)st.n_branches * st.n_frames; ++k) if (st.spawn_sel[k]) return w->fail(GGRS_E_INVALID, "spawn_sel names a spawn but the world has no spawn system its generated kernel runs"); }
        else {
            const ggrs_system_desc& sd = w->systems[w->jit_spawn_sys];
            const bool custom = sd.kind == GGRS_SYS_SPAWN_CUSTOM;
            for (uint64_t k = 0; k < (uint64_t)st.n_branches * st.n_frames; ++k) if (st.spawn_sel[k] > st.n_spawn_table) return w->fail(GGRS_E_INVALID, "spawn_sel[%llu] = %u names no entry of spawn_table (%u entries)", (unsigned long long)k, st.spawn_sel[k], st.n_spawn_table);
            if (st.n_spawn_table && !st.spawn_table) return w->fail(GGRS_E_INVALID, "spawn_table is NULL");
            for (uint32_t t = 0; t < st.n_spawn_table; ++t) {
                const ggrs_branch_spawn& e = st.spawn_table[t];
                if (e.count > 0xFFFFFFFFull) return w->fail(GGRS_E_CAPACITY, "spawn_table[%u]: %llu entities", t, (unsigned long long)e.count);
                if (!custom && e.count && (!e.vx || !e.vy)) return w->fail(GGRS_E_INVALID, "spawn_table[%u]: a spawn of %llu but vx / vy is NULL", t, (unsigned long long)e.count);
                if (custom) {
                    const ggrs_world::SpawnSys& sp = w->spawn_customs[sd.comp[0]];
                    const uint64_t need = sp.payload_stride ? (uint64_t)sp.payload_stride * e.count : e.payload_bytes;
                    if (need && !e.payload) return w->fail(GGRS_E_INVALID, "spawn_table[%u]: the spawn system reads %llu payload bytes but payload is NULL", t, (unsigned long long)need);
                    if (sp.payload_stride && e.payload_bytes && e.payload_bytes < need) return w->fail(GGRS_E_INVALID, "spawn_table[%u]: payload_bytes = %llu, the spawn of %llu needs %llu", t, (unsigned long long)e.payload_bytes, (unsigned long long)e.count, (unsigned long long)need);
                }
            }
        }
    }
    return GGRS_OK;
}
// res_first: result slot of branch 0's first Save.  keep (may be null): filled with what was retained.
int run_branch_step(ggrs_world* w, const ggrs_branch_step& st, uint32_t res_first, BranchKeep* keep) {
    const JitLayout& L = *w->jl;
    const uint32_t B = st.n_branches, T = st.n_frames, S = (st.flags & GGRS_BRANCH_SAVE_LAST) ? T : T - 1;
    const bool keep_all = (st.flags & GGRS_BRANCH_RETAIN_ALL) != 0, keep_any = keep_all || (st.flags & GGRS_BRANCH_RETAIN_NEWEST);
    const bool tail_adv = S < T;                                     // the last AdvanceFrame has no SaveGameState behind it: its result is the branch's "live" output
    const uint32_t n_out = S + (tail_adv ? 1u : 0u);
    if (keep) keep->valid = false;
    if (w->ring_frame.empty() || w->ring_frame.front() != w->frame)
        return w->fail(GGRS_E_NO_SNAPSHOT, "a branch step starts from the snapshot of the current frame %d, and the ring's newest snapshot is %s (end the prefix with SaveGameState)", w->frame,
                       w->ring_frame.empty() ? "none" : std::to_string(w->ring_frame.front()).c_str());
    int rc = materialise_live(w); if (rc) return rc;
    Block& src = w->slots[w->ring_slot.front()];
    const int32_t F = w->frame;
    const uint32_t n_cks = w->cks_args.n_cks, ib = w->input_bytes;
    const int spawn_sys = w->jit_spawn_sys;
    const ggrs_system_desc* sd = spawn_sys >= 0 ? &w->systems[spawn_sys] : nullptr;
    const bool custom = sd && sd->kind == GGRS_SYS_SPAWN_CUSTOM;
    const ggrs_world::SpawnSys* sp = custom ? &w->spawn_customs[sd->comp[0]] : nullptr;
    // ---- the shape every member shares
    GgrsJitArgs j; memset(&j, 0, offsetof(GgrsJitArgs, inputs));
    j.src = src.ptr; j.live = nullptr; j.len = src.len; j.src_is_live = 0; j.n_steps = T; j.n_saves = S;
    for (uint32_t i = 0; i < T; ++i) {
        j.op_bits |= 1ull << j.n_ops; ++j.n_ops;
        j.dt_bits[i] = dt_bits_for_frame(w->fps, F + 1 + (int32_t)i);
        j.step_frame[i] = F + 1 + (int32_t)i; j.step_confirmed[i] = w->confirmed;
        if (w->jit_box_sys >= 0) { float dtf; memcpy(&dtf, &j.dt_bits[i], 4); const float fp = powf(w->systems[w->jit_box_sys].fparam[2], dtf); memcpy(&j.aux_bits[i], &fp, 4); }
        if (i < S) { j.save_frame[i] = F + 1 + (int32_t)i; ++j.n_ops; }
    }
    j.skip_live = (keep_any && tail_adv) ? 0u : 1u;
    // ---- spawn payloads + member records in the staging ring (pinned; one copy to its device twin below)
    std::vector<uint64_t> pay_off(st.n_spawn_table, 0), pay_bytes(st.n_spawn_table, 0);
    uint64_t total = (uint64_t)B * L.m.bytes;
    for (uint32_t t = 0; t < st.n_spawn_table && st.spawn_sel && sd; ++t) {
        const ggrs_branch_spawn& e = st.spawn_table[t];
        pay_bytes[t] = custom ? (sp->payload_stride ? (uint64_t)sp->payload_stride * e.count : e.payload_bytes) : 8 * e.count;
        pay_off[t] = total; total += (pay_bytes[t] + 15u) & ~15ull;
    }
    uint64_t soff = 0;
    if (!stage_ring_alloc(w, total, &soff)) {
        if (total > w->stage_bytes) return w->fail(GGRS_E_CAPACITY, "a branch step of %llu bytes of member records and spawn payloads exceeds the staging buffer (%llu bytes: GGRS_STAGE_BYTES)", (unsigned long long)total, (unsigned long long)w->stage_bytes);
        HIPCHK(w, hipStreamSynchronize(w->stream));
        stage_ring_reset(w);
        if (!stage_ring_alloc(w, total, &soff)) return w->fail(GGRS_E_CAPACITY, "the staging buffer cannot hold a branch step of %llu bytes", (unsigned long long)total);
    }
    for (uint32_t t = 0; t < st.n_spawn_table && st.spawn_sel && sd; ++t) {
        const ggrs_branch_spawn& e = st.spawn_table[t];
        if (!pay_bytes[t]) continue;
        unsigned char* dst = w->h_stage + soff + pay_off[t];
        if (custom) memcpy(dst, e.payload, pay_bytes[t]);
        else { memcpy(dst, e.vx, e.count * 4); memcpy(dst + e.count * 4, e.vy, e.count * 4); }
    }
    // ---- members: inputs, spawns, lens; with retention the blocks their frames land in and the rows each store moves (row versions)
    size_t n_keep = 0;
    if (keep_any) { n_keep = (size_t)B * (keep_all ? n_out : 1u); rc = spec_blocks_reserve(w, n_keep); if (rc) return rc; }
    // value tags: the launch's ids are reserved BEFORE any block's tag_ok is read or set below (a start-over of the numbering clears them all)
    if (w->vtags && keep_any) { rc = vtags_reserve(w, T, B, &j.tag_base); if (rc) return rc; }
    if (keep) { keep->n_branches = B; keep->n_out = n_out; keep->n_frames = T; keep->base_frame = F; keep->blk.assign((size_t)B * n_out, -1); }
    uint64_t cover = std::max(src.dirty_len, src.len), max_len = src.len, load_rows = jit_static_reads(w), store_bytes = 0;
    if (keep_any) for (size_t k = 0; k < n_keep; ++k) cover = std::max(cover, w->spec_blocks[k].dirty_len);
    std::vector<ver_t> cv;
    std::vector<unsigned char> rec((size_t)B * L.m.bytes);
    std::vector<Block*> touched;
    const size_t in_row = (size_t)w->max_players * (ib + 1);
    size_t next_blk = 0;
    for (uint32_t b = 0; b < B; ++b) {
        uint64_t len_b = src.len;
        if (keep_any) {
            cv = src.ver;
            if (w->has_strategy) for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].s_n_words && !w->comps[c].no_rollback) for (uint32_t k = 0; k < w->comps[c].n_words; ++k) cv[w->comps[c].col_base + k] = ++w->ver_counter;
        }
        auto keep_into = [&](uint32_t o, ggrs_u64* rows_out, ggrs_u32* pm_out, ggrs_u64* tagok_out) -> Block* {
            if (!keep_any || !(keep_all || o == n_out - 1)) return nullptr;
            Block* d = &w->spec_blocks[next_blk];
            if (keep) keep->blk[(size_t)b * n_out + o] = (int)next_blk;
            ++next_blk;
            uint64_t m = 0;
            for (uint32_t c = 0; c < w->n_tcols && c < 64; ++c) if (w->col_rb[c] && ver_differs(w, *d, cv, c)) m |= 1ull << c;
            *rows_out = m; *pm_out = pmask_differs(w, *d, cv);
            *tagok_out = block_tagok(w, *d);
            if (w->vtags) d->tag_ok |= m & w->tag_cols; else d->tag_ok &= ~m;
            d->ver = cv; d->len = len_b;
            load_rows |= m; store_bytes += rows_bytes_per_slot(w, m, o < S) * len_b;
            touched.push_back(d);
            return d;
        };
        uint32_t k_save = 0;
        for (uint32_t i = 0; i < T; ++i) {
            const size_t bi = (size_t)b * T + i;
            // PlayerInputs of the frame
            unsigned char* row = j.inputs[i];
            memset(row, 0, in_row);
            const uint32_t np = std::min<uint32_t>(st.n_inputs, w->max_players);
            if (np) memcpy(row, st.inputs + bi * (size_t)st.n_inputs * ib, (size_t)np * ib);
            if (np && st.status) memcpy(row + (size_t)w->max_players * ib, st.status + bi * st.n_inputs, np);
            j.n_inputs[i] = (unsigned char)np;
            if (keep_any) for (auto& cols : w->sys_writes) for (uint32_t c : cols) cv[c] = ++w->ver_counter;       // ver_step on the branch's own versions
            // the spawn system (the host decided: spawn_sel), after the frame's other systems
            j.spawn_count[i] = 0; j.spawn_first[i] = 0; j.spawn_payload[i] = nullptr;
            const uint32_t sel = (st.spawn_sel && sd) ? st.spawn_sel[bi] : 0u;
            if (sel) {
                const ggrs_branch_spawn& e = st.spawn_table[sel - 1];
                const bool fires = e.count && (custom || (np && spawn_pressed(w, *sd, row, np)));                   // spawn_pressed, particles.rs:254-256 (advance_spawns)
                if (fires) {
                    if (len_b + e.count > w->capacity) return w->fail(GGRS_E_CAPACITY, "branch %u: spawn of %llu exceeds capacity %llu", b, (unsigned long long)e.count, (unsigned long long)w->capacity);
                    j.spawn_count[i] = (uint32_t)e.count; j.spawn_first[i] = len_b; j.spawn_payload[i] = pay_bytes[sel - 1] ? w->d_stage + soff + pay_off[sel - 1] : nullptr;
                    len_b += e.count;
                    if (keep_any) {
                        auto touch = [&](uint32_t c) { for (uint32_t k = 0; k < w->comps[c].n_words; ++k) cv[w->comps[c].col_base + k] = ++w->ver_counter; cv[ver_presence(w, c)] = ++w->ver_counter; };
                        if (custom) { for (uint32_t c = 0; c < w->comps.size(); ++c) if ((sp->bundle_mask >> c) & 1ull) touch(c); }
                        else { touch(sd->comp[0]); touch(sd->comp[1]); touch(sd->comp[2]); }
                    }
                }
            }
            if (i < S) {
                j.save_len[k_save] = len_b; j.save_rows[k_save] = 0; j.save_pmask[k_save] = 0;
                j.save_tagok[k_save] = 0;
                Block* d = keep_into(k_save, &j.save_rows[k_save], &j.save_pmask[k_save], &j.save_tagok[k_save]);
                j.save_dst[k_save] = d ? d->ptr : nullptr;
                ++k_save;
            }
        }
        j.live = nullptr; j.live_rows = 0; j.live_pmask = 0; j.live_tagok = 0;
        if (tail_adv) { Block* d = keep_into(n_out - 1, &j.live_rows, &j.live_pmask, &j.live_tagok); j.live = d ? d->ptr : nullptr; }
        max_len = std::max(max_len, len_b);
        jit_pack_member(L, j, rec.data() + (size_t)b * L.m.bytes);
    }
    memcpy(w->h_stage + soff, rec.data(), rec.size());
    HIPCHK(w, hipMemcpyAsync(w->d_stage + soff, w->h_stage + soff, total, hipMemcpyHostToDevice, w->stream));
    cover = std::max(cover, max_len);
    for (Block* d : touched) d->dirty_len = std::max(src.dirty_len, max_len);
    // ---- one launch for all members, one k_gen_finalize for all their Saves
    const uint32_t g = std::max<uint32_t>(1, (uint32_t)((cover + 255) / 256));
    const uint64_t parts_need = (uint64_t)B * std::max(1u, S) * (n_cks + 1) * g;
    if (parts_need > w->branch_parts_cap) {
        if (w->d_branch_parts) { HIPCHK(w, hipStreamSynchronize(w->stream)); (void)hipFree(w->d_branch_parts); w->d_branch_parts = nullptr; w->branch_parts_cap = 0; }
        HIPCHK(w, hipMalloc((void**)&w->d_branch_parts, parts_need * 8));
        w->branch_parts_cap = parts_need;
    }
    memset(j.save_dst, 0, sizeof j.save_dst); memset(j.save_rows, 0, sizeof j.save_rows); memset(j.save_pmask, 0, sizeof j.save_pmask);
    j.live = w->live.ptr; j.live_rows = 0; j.live_pmask = 0;           // (never used: every member's record says where its live output goes)
    j.mtab = w->d_stage + soff;
    j.load_rows = load_rows;
    j.parts = reinterpret_cast<ggrs_u64*>(w->d_branch_parts); j.part_stride = g; j.part_tstride = 1;
    j.n_units = std::max<uint32_t>(1, (uint32_t)((cover + 63) / 64));
    // branch blocks are written once and not read before an adoption: past what the caches hold they stream around them
    j.nt = (cover > JIT_NT_MIN_SLOTS || store_bytes > (128ull << 20)) ? 1u : 0u;
    j.cached_saves = 0; j.nt_loads = 0; j.dp_s = 0;
    j.vtags = (w->vtags && keep_any) ? 1u : 0u;
    if (j.vtags) j.src_tagok = block_tagok(w, src);
    w->batch_ev_attached = false;
    hipFunction_t fn = jit_spec_for(w, j, true);                      // the copy of the kernel built for this op sequence, once the session has sent it often enough
    if (!fn) fn = w->jit_fn;
    rc = launch_jit(w, fn, jit_grid(g), 1, B, jit_lane_fold_bytes(w, n_cks, S), j, rows_bytes_per_slot(w, load_rows, true) * src.len * B + store_bytes); if (rc) return rc;
    if (S) {
        GenFinArgs f = make_gen_fin(w, j, g, n_cks, res_first);
        f.mtab = w->d_stage + soff; f.mstride = L.m.bytes; f.moff_save_len = L.m.save_len;
        {
            ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
            hipLaunchKernelGGL(k_gen_finalize, dim3(S * B), dim3(FIN_TPB), 0, w->stream, f);
        }
        HIPCHK(w, hipGetLastError());
    }
    if (keep) keep->valid = keep_any;
    return GGRS_OK;
}

// which runner serves this world's request lists right now (nullptr: one launch per request)
typedef int (*GroupRunner)(ggrs_world*, const ggrs_request*, uint32_t, uint64_t*, uint32_t, bool, uint32_t*);
GroupRunner group_runner(const ggrs_world* w) {
    if (w->gen_ok) return run_request_groups_gen;
    return nullptr;
}

// Requests are validated BEFORE any host bookkeeping (frame counters, ring) is touched: a malformed list fails with
// GGRS_E_INVALID and leaves the world exactly as it was.
int validate_requests(ggrs_world* w, const ggrs_request* reqs, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        const ggrs_request& r = reqs[i];
        if (r.kind != GGRS_REQ_SAVE && r.kind != GGRS_REQ_LOAD && r.kind != GGRS_REQ_ADVANCE)
            return w->fail(GGRS_E_INVALID, "request %u: unknown request kind %u", i, r.kind);
        if (r.kind != GGRS_REQ_ADVANCE) continue;
        if (r.n_inputs > w->max_players) return w->fail(GGRS_E_INVALID, "request %u: %u player inputs (at most %u: ggrs_hip_set_input_layout)", i, r.n_inputs, w->max_players);
        if (r.n_inputs && !r.inputs) return w->fail(GGRS_E_INVALID, "request %u: n_inputs = %u but inputs is NULL", i, r.n_inputs);
        if (r.status) for (uint32_t k = 0; k < r.n_inputs; ++k) if (r.status[k] > GGRS_INPUT_DISCONNECTED) return w->fail(GGRS_E_INVALID, "request %u: InputStatus %u of player %u is none of Confirmed / Predicted / Disconnected", i, r.status[k], k);
        if (!advance_spawns(w, r)) continue;
        bool custom = false; const ggrs_world::SpawnSys* sp = nullptr;
        for (auto& s : w->systems) if (s.kind == GGRS_SYS_SPAWN_CUSTOM) { custom = true; sp = &w->spawn_customs[s.comp[0]]; }
        if (!custom && (!r.spawn_vx || !r.spawn_vy)) return w->fail(GGRS_E_INVALID, "request %u: a spawn of %llu fires but spawn_vx / spawn_vy is NULL", i, (unsigned long long)r.spawn_count);
        if (custom) {
            const uint64_t need = sp->payload_stride ? (uint64_t)sp->payload_stride * r.spawn_count : r.spawn_payload_bytes;
            if (need && !r.spawn_payload) return w->fail(GGRS_E_INVALID, "request %u: the spawn system reads %llu payload bytes but spawn_payload is NULL", i, (unsigned long long)need);
            if (sp->payload_stride && r.spawn_payload_bytes && r.spawn_payload_bytes < need) return w->fail(GGRS_E_INVALID, "request %u: spawn_payload_bytes = %llu, the spawn of %llu needs %llu", i, (unsigned long long)r.spawn_payload_bytes, (unsigned long long)r.spawn_count, (unsigned long long)need);
        }
    }
    return GGRS_OK;
}
inline bool range_ok(uint64_t first, uint64_t count, uint64_t capacity) { return first <= capacity && count <= capacity - first; }

}  // namespace
