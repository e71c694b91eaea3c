// host_seal.hpp -- sealing a world: the layout is fixed, the fused paths are recognised / generated, the arena is carved.
// Part of the single translation unit ggrs_hip.hip.
#pragma once

namespace {

int seal_impl(ggrs_world* w);
// Sealing fixes the layout and carves the arena, lazily, on the first call that needs device state.  It is
// failure-atomic: whatever a failed attempt allocated is released, and the failure LATCHES -- every later call
// reports the same error instead of carving a second arena over half-initialised bookkeeping.
// A library-owned arena goes back to the runtime when its world closes or fails to seal.
void arena_release(ggrs_world* w) {
    if (!(w->own_arena && w->arena)) return;
    (void)hipFree(w->arena);
    w->arena = nullptr; w->arena_bytes = 0; w->own_arena = false;
}

void sp_release(ggrs_world* w) {
    if (w->d_sp_sums) (void)hipFree(w->d_sp_sums);
    if (w->d_sp_prec) (void)hipFree(w->d_sp_prec);
    if (w->d_sp_link) (void)hipFree(w->d_sp_link);
    if (w->h_sp_len) (void)hipHostFree((void*)w->h_sp_len);
    w->d_sp_sums = nullptr; w->d_sp_prec = nullptr; w->d_sp_link = nullptr; w->h_sp_len = nullptr; w->d_sp_len = nullptr;
}
int seal(ggrs_world* w) {
    if (w->layout_only) return w->fail(GGRS_E_NO_DEVICE, "GGRS_WORLD_LAYOUT_ONLY world: there is no device behind it");
    if (w->sealed) return GGRS_OK;
    if (w->seal_error) return w->seal_error;
    const int rc = seal_impl(w);
    if (rc == GGRS_OK) return rc;
    const std::string why = w->err;
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    if (w->d_gen_parts) { (void)hipFree(w->d_gen_parts); w->d_gen_parts = nullptr; w->d_ff_rows[0] = w->d_ff_rows[1] = nullptr; }
    if (w->d_skip) { (void)hipFree(w->d_skip); w->d_skip = nullptr; }
    sp_release(w);
    if (w->h_results) { (void)hipHostFree(w->h_results); w->h_results = nullptr; w->d_results = nullptr; }
    if (w->h_stage) { (void)hipHostFree(w->h_stage); w->h_stage = nullptr; w->d_hstage = nullptr; }
    if (w->h_rows) { (void)hipHostFree(w->h_rows); w->h_rows = nullptr; w->d_rows = nullptr; }
    arena_release(w);
    (void)hipGetLastError();
    w->slots.clear(); w->free_slots.clear(); w->live = Block{};
    w->sealed = false; w->seal_error = rc;
    w->err = "world could not be sealed (permanent): " + why;
    return rc;
}

// ---- recognise the particles schedule: [PARTICLES_UPDATE, TTL_DESPAWN] (+ optional SPAWN) over three distinct components.
// fused_ok: the per-request path steps it with ONE kernel (k_particles_step, checksum partials of the post-step state).
void recognise_particles(ggrs_world* w) {
    w->fused_ok = false; w->f_spawn = -1; w->fused_cks = false; w->f_cksT = w->f_cksV = false;
    int upd = -1, ttl = -1, other = 0;
    for (size_t i = 0; i < w->systems.size(); ++i) {
        switch (w->systems[i].kind) {
        case GGRS_SYS_PARTICLES_UPDATE: if (upd < 0) upd = (int)i; else ++other; break;
        case GGRS_SYS_TTL_DESPAWN: if (ttl < 0) ttl = (int)i; else ++other; break;
        case GGRS_SYS_PARTICLES_SPAWN: w->f_spawn = (int)i; break;
        default: ++other;
        }
    }
    if (upd < 0 || ttl < 0 || other != 0 || (w->flags & GGRS_WORLD_UNFUSED)) return;
    const ggrs_system_desc& u = w->systems[upd]; const ggrs_system_desc& l = w->systems[ttl];
    const Comp& T = w->comps[u.comp[0]]; const Comp& V = w->comps[u.comp[1]]; const Comp& L = w->comps[l.comp[0]];
    if (!(T.word_bytes == 4 && V.word_bytes == 4 && L.word_bytes == 8 && u.word[0] + 3 <= T.n_words && u.word[1] + 3 <= V.n_words &&
          !T.no_rollback && !V.no_rollback && !L.no_rollback)) return;
    w->fused_ok = true;
    w->f_T = (int)u.comp[0]; w->f_V = (int)u.comp[1]; w->f_L = (int)l.comp[0];
    w->f_tw = u.word[0]; w->f_vw = u.word[1]; w->f_lw = l.word[0];
    for (int k = 0; k < 3; ++k) w->f_g[k] = u.fparam[k];
    // does the fused step cover every checksum spec?  (a word list naming exactly the three stepped words, in order)
    bool all = true;
    for (uint32_t c : w->cks_comp) {
        const Comp& cc = w->comps[c];
        const uint32_t base = ((int)c == w->f_T) ? w->f_tw : w->f_vw;
        const bool is3 = cc.cks_source.empty() && cc.cks_words.size() == 3 && cc.cks_words[0] == base && cc.cks_words[1] == base + 1 && cc.cks_words[2] == base + 2;
        if ((int)c == w->f_T && is3 && w->f_T != w->f_V) w->f_cksT = true;
        else if ((int)c == w->f_V && is3 && w->f_T != w->f_V) w->f_cksV = true;
        else all = false;
    }
    w->fused_cks = all;
    if (!all) w->f_cksT = w->f_cksV = false;

}

int seal_impl(ggrs_world* w) {
    if (total_rows(w) > (uint32_t)MAX_ROWS) return w->fail(GGRS_E_INVALID, "too many registered words (%u rows > %d)", total_rows(w), MAX_ROWS);
    // The particles kernels address their columns with the tile stride of the ROLLBACK columns; a live-only
    // (GGRS_COMP_NO_ROLLBACK) column is a plain array with a different stride.  The flag is set after registration
    // (register_component_ex), so the check lives here rather than in add_system.
    for (auto& sd : w->systems) {
        if (sd.kind != GGRS_SYS_PARTICLES_UPDATE && sd.kind != GGRS_SYS_TTL_DESPAWN && sd.kind != GGRS_SYS_PARTICLES_SPAWN) continue;
        const uint32_t nc = sd.kind == GGRS_SYS_PARTICLES_UPDATE ? 2u : (sd.kind == GGRS_SYS_TTL_DESPAWN ? 1u : 3u);
        for (uint32_t k = 0; k < nc; ++k)
            if (sd.comp[k] >= w->comps.size() || w->comps[sd.comp[k]].no_rollback)
                return w->fail(GGRS_E_INVALID, "system %u runs over component %u, which is not registered for rollback (GGRS_COMP_NO_ROLLBACK): unsupported", sd.kind, sd.comp[k]);
    }
    HIPCHK(w, hipSetDevice(w->device));
    build_layout(w);

    // ---- checksum specs (the per-request k_checksum's view: one UnitDesc per hashed word)
    w->cks_comp.clear(); w->custom_hashers = false;
    std::vector<UnitDesc> units;
    memset(&w->cks_args, 0, sizeof w->cks_args);
    for (uint32_t c = 0; c < w->comps.size(); ++c) {
        Comp& cc = w->comps[c];
        if (!cc.checksummed) continue;
        const uint32_t k = (uint32_t)w->cks_comp.size();
        w->cks_comp.push_back(c);
        w->custom_hashers |= !cc.cks_source.empty();
        w->cks_args.off_present[k] = w->off_present[c];
        w->cks_args.unit_base[k] = (uint32_t)units.size();
        for (uint32_t wi : cc.cks_words) units.push_back({w->col_off[cc.col_base + wi], cc.word_bytes, w->col_ts[cc.col_base + wi]});
        w->cks_args.n_units[k] = (uint32_t)units.size() - w->cks_args.unit_base[k];
        if (w->cks_args.n_units[k] > (uint32_t)MAX_UNITS) return w->fail(GGRS_E_INVALID, "checksum spec too long");
    }
    w->cks_args.n_cks = (uint32_t)w->cks_comp.size();
    w->cks_args.off_alive = w->off_alive;

    recognise_particles(w);

    // ---- the kernel generated for this world (kernel_gen.hpp): every world it covers, unless groups are off
    w->gen_ok = false; w->jit_box_sys = -1; w->jit_marks = false; w->jit_reads_inputs = false; w->jit_spawn_sys = -1;
    if (!(w->flags & (GGRS_WORLD_NO_GROUPS | GGRS_WORLD_UNFUSED)) && w->ts > 0 && !w->knobs.tick_jit) w->jit_status = "disabled (GGRS_TICK_JIT=0)";
    if (!(w->flags & (GGRS_WORLD_NO_GROUPS | GGRS_WORLD_UNFUSED)) && w->ts > 0 && w->knobs.tick_jit) {
        std::string src;
        if (!jit_source(w, src)) w->jit_status = "not covered by the generator (a system writes a live-only component, or too many words per entity)";
        else {
            if (w->knobs.debug_jit > 1) fprintf(stderr, "%s\n", src.c_str());
            const std::string keep = w->err;
            w->jit_src = src;
            if (jit_cached(w, src, &w->jit_fn, &w->jit_entry, &w->jit_origin) != GGRS_OK) {
                if (w->knobs.debug_jit) fprintf(stderr, "[ggrs_hip] generated request-group kernel rejected: %s\n", w->err.c_str());
                w->jit_status = (hiprtc_for(w).lib ? "rejected: " : "no run-time compiler and no shipped code object for this world: ") + w->err.substr(0, 300);
                w->jit_fn = nullptr; w->err = keep;
            } else w->jit_status = "ok";
        }
        if (w->jit_fn) {
            for (size_t i = 0; i < w->systems.size(); ++i) {
                const ggrs_system_desc& d = w->systems[i];
                w->jit_reads_inputs |= d.kind == GGRS_SYS_CUSTOM || d.kind == GGRS_SYS_BOX_MOVE || d.kind == GGRS_SYS_SPAWN_CUSTOM;
                w->jit_marks |= (d.kind == GGRS_SYS_CUSTOM && w->customs[d.comp[0]].may_defer) || (d.kind == GGRS_SYS_SAT_SUB_DESPAWN && d.iparam[1] == GGRS_DESPAWN_ROLLBACK);
                if (d.kind == GGRS_SYS_BOX_MOVE) w->jit_box_sys = (int)i;
            }
            w->gen_ok = true;
            w->jit_spawn_sys = jit_fused_spawn_system(w);              // the spawn system runs inside request groups
            w->dev_spawn = jit_dev_spawn(w);
            delete w->jl; w->jl = new JitLayout(jit_layout(w));
            w->cap_saves = w->jl->cap_saves; w->cap_steps = w->jl->cap_steps;
            w->jit_argbuf.assign(w->jl->bytes, 0);
        }
    }
    if (w->custom_hashers && !w->gen_ok)
        return w->fail(GGRS_E_INVALID, "a user-written checksum hasher needs the generated request-group kernel, which this world does not have: %s", w->jit_status.c_str());
    if (w->has_strategy && !w->gen_ok)
        return w->fail(GGRS_E_INVALID, "a component under a Strategy (ggrs_hip_register_component_strategy) needs the generated request-group kernel, which this world does not have: %s", w->jit_status.c_str());
    for (auto& sd : w->systems) if (sd.kind == GGRS_SYS_SPAWN_CUSTOM && !(w->gen_ok && w->jit_spawn_sys >= 0))
        return w->fail(GGRS_E_INVALID, "a user-written spawn system (ggrs_hip_add_spawn_system) runs inside the generated request-group kernel, which this world does not have "
                                       "(or the schedule holds a second spawn system): %s", w->jit_status.c_str());

    // ---- arena carve
    const uint32_t n_tiles = (uint32_t)(w->cap_pad / TILE);
    w->gen_part_stride = (uint32_t)(w->cap_pad / 256);            // one partial row entry per 256-slot workgroup of the generated kernel
    w->gen_parts_saves = w->cap_pad <= 512 * 1024 ? 8 * MAX_TICK_SAVES : MAX_TICK_SAVES;   // small worlds: room for a batch of 16 eight-Save groups
    w->part_stride = n_tiles + 4096 / 1;            // + room for spawn partial blocks
    const uint64_t parts_bytes = align_up((uint64_t)(w->cks_args.n_cks + 1) * w->part_stride * 8, ALIGN);
    w->max_results = 16384;                             // pinned result ring (256 KiB): a fan-out step of 256 branches x 8 frames alone is 2048
    const uint64_t units_bytes = align_up((units.size() + 1) * sizeof(UnitDesc), ALIGN);
    w->stage_bytes = w->knobs.stage_bytes; w->stage_used = w->stage_tail = 0;
    const uint64_t stage_bytes = w->stage_bytes;
    const uint64_t need = (uint64_t)(w->max_depth + 1) * w->state_bytes + w->side_bytes + parts_bytes + units_bytes + ALIGN + stage_bytes + ALIGN;
    if (w->arena) {
        if (w->arena_bytes < need) return w->fail(GGRS_E_INVALID, "arena too small: need %llu bytes, have %llu", (unsigned long long)need, (unsigned long long)w->arena_bytes);
    } else {
        // plain hipMalloc pages: the generated kernel is faster on them than on a physically contiguous (write-through) arena
        // (1 M: 169 vs 149 G entity-frames/s, r03y3) -- the opt-in contiguous arena of rounds 2-4 and its parking list are gone
        uint8_t* pa = nullptr;
        if (hipMalloc((void**)&pa, need) != hipSuccess) { (void)hipGetLastError(); return w->fail(GGRS_E_HIP, "hipMalloc of %llu bytes failed", (unsigned long long)need); }
        w->arena = pa; w->arena_bytes = need; w->own_arena = true;
    }
    // GGRS_DEBUG_POISON=1: fill a library-owned arena with a garbage pattern before anything is initialised -- a read of memory the
    // library never wrote (hidden by whatever a previous allocation left there) then fails the parity tests every time
    if (w->knobs.debug_poison && w->own_arena) HIPCHK(w, hipMemsetAsync(w->arena, 0xA5, need, w->stream));
    uint8_t* p = w->arena;
    const uint32_t ncols = (uint32_t)w->col_off.size();
    // value tags by default where a steady Save is bound by bytes: what the systems write x the world's slots (the knob: ggrs_dbg_set_value_tags)
    w->vtags = w->gen_ok && vtags_policy(w);
    // (like jiffies: the ids' 32-bit numbering starts over within a world's first few dozen launches -- host_groups.hpp vtags_reserve --, so that path is run by every
    // test of a tag-keeping world instead of once per ~4e8 launches)
    if (w->vtags) w->tag_counter = 0xFFFFFFF0u - 400u;
    w->live.ptr = p; p += w->state_bytes;
    w->live.ver.assign(ncols + w->comps.size(), 0);                                   // == cur_ver: nothing has been written yet
    w->slots.resize(w->max_depth);
    for (uint32_t i = 0; i < w->max_depth; ++i) {
        w->slots[i].ptr = p; p += w->state_bytes; w->slots[i].ver.assign(ncols + w->comps.size(), VER_NONE);
        w->free_slots.push_back((int)(w->max_depth - 1 - i));
    }
    uint8_t* const side = p; p += w->side_bytes;          // == live.ptr + side_off (build_layout)
    w->d_parts = (uint64_t*)p; p += parts_bytes;
    w->d_units = (UnitDesc*)p; p += units_bytes;
    w->d_maskoffs = (uint64_t*)p; p += ALIGN;
    w->d_stage = p; p += stage_bytes;
    w->cks_args.parts = w->d_parts;
    w->cks_args.part_cnt = w->d_parts + (uint64_t)w->cks_args.n_cks * w->part_stride;
    w->cks_args.part_stride = w->part_stride;

    // Checksum(u128) results are written by the kernels straight into pinned, device-mapped host memory:
    // no device->host copy node per request list, one stream sync makes them visible.
    HIPCHK(w, hipHostMalloc((void**)&w->h_results, (size_t)w->max_results * 16 + ggrs_world::SPIN_TAGS * 8, hipHostMallocMapped));
    HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_results, w->h_results, 0));
    w->h_done = w->h_results + 2 * (size_t)w->max_results; w->d_done = w->d_results + 2 * (size_t)w->max_results;
    memset((void*)w->h_done, 0, ggrs_world::SPIN_TAGS * 8); w->spin_seq = 0; w->spin_n = 0;
    HIPCHK(w, hipHostMalloc((void**)&w->h_stage, stage_bytes, hipHostMallocMapped));       // pinned AND device-mapped: a fused spawn's payload is read by the group's launch straight from here
    HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_hstage, w->h_stage, 0));
    if (w->jit_fn) {
        w->rows_cap = 1u << 20;                                    // 8 MiB of partial rows between two collects
        HIPCHK(w, hipHostMalloc((void**)&w->h_rows, w->rows_cap * 8, hipHostMallocMapped));
        HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_rows, w->h_rows, 0));
    }
    if (w->knobs.debug_poison) { memset(w->h_results, 0xA5, (size_t)w->max_results * 16); memset(w->h_stage, 0xA5, stage_bytes); }
    // zero header + masks of EVERY block (columns need no init: masked by liveness).  Invariant
    // relied on by k_copy_state: mask words beyond a block's dirty_len are zero.
    {
        const uint64_t head = ALIGN + (uint64_t)w->plan.n_masks * align_up(w->cap_pad / 8, ALIGN);   // header + every mask
        HIPCHK(w, hipMemsetAsync(w->live.ptr, 0, head, w->stream));
        for (auto& b : w->slots) HIPCHK(w, hipMemsetAsync(b.ptr, 0, head, w->stream));
        HIPCHK(w, hipMemsetAsync(side, 0, w->side_bytes, w->stream));     // no markers, no non-rollback components yet
        // value tags: 0 = "no identity" in every block (a poisoned or recycled arena must not carry tags that happen to match)
        const uint64_t tag_bytes = w->state_bytes - w->off_tags;
        HIPCHK(w, hipMemsetAsync(w->live.ptr + w->off_tags, 0, tag_bytes, w->stream));
        for (auto& b : w->slots) HIPCHK(w, hipMemsetAsync(b.ptr + w->off_tags, 0, tag_bytes, w->stream));
    }
    if (!units.empty()) HIPCHK(w, hipMemcpyAsync(w->d_units, units.data(), units.size() * sizeof(UnitDesc), hipMemcpyHostToDevice, w->stream));
    if (w->gen_ok) {
        // k_gen_finalize's row buffer [saves][n_cks + 1][one row per 256-slot workgroup]; then the two row buffers of the fold-forward path
        // ([cap_saves][n_cks + 1][one row per workgroup] each, used alternately by consecutive launches)
        const size_t bytes = align_up((size_t)w->gen_parts_saves * (w->cks_args.n_cks + 1) * w->gen_part_stride * 8, ALIGN);
        const size_t ff_bytes = align_up((size_t)MAX_TICK_SAVES * (w->cks_args.n_cks + 1) * w->gen_part_stride * 8, ALIGN);
        HIPCHK(w, hipMalloc((void**)&w->d_gen_parts, bytes + 2 * ff_bytes));
        uint8_t* const base = reinterpret_cast<uint8_t*>(w->d_gen_parts);
        w->d_ff_rows[0] = reinterpret_cast<uint64_t*>(base + bytes);
        w->d_ff_rows[1] = reinterpret_cast<uint64_t*>(base + bytes + ff_bytes);
        w->ff_cur = 0; w->ff_pending = ggrs_world::FfPending{};
        if (w->knobs.debug_poison) HIPCHK(w, hipMemsetAsync(w->d_gen_parts, 0xA5, bytes + 2 * ff_bytes, w->stream));
        // self-fold reads these buffers as {value, tag} cells WHILE the launch that writes them runs: a cell must never carry a tag this world will use before the
        // launch that owns it has written it.  Tags count up per world, so memory recycled from an earlier world of the process could (the fuzz under
        // GGRS_FOLD_FORWARD_MIN_WGS=0 found it: stale cells of the previous test's world, same tags) -- the buffers start zeroed (no tag is 0)
        else HIPCHK(w, hipMemsetAsync(w->d_ff_rows[0], 0, 2 * ff_bytes, w->stream));
    }
    if (w->vtags) { HIPCHK(w, hipMalloc((void**)&w->d_skip, 8)); HIPCHK(w, hipMemsetAsync(w->d_skip, 0, 8, w->stream)); }
    if (w->dev_spawn) {
        // every launch of such a world covers its whole capacity and must be resident as a whole (grid barriers inside): what the device holds of this kernel bounds the world
        w->sp_tiles = (uint32_t)(((w->capacity + 63) / 64 + 3) / 4);                 // == the workgroups that own a tile when a launch covers `capacity` slots
        int per_cu = 0;
        HIPCHK(w, hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, w->jit_fn, TPB, jit_lane_fold_bytes(w, w->cks_args.n_cks, w->cap_saves)));
        // the occupancy query is not enough (resident_wgs_per_cu): the register files bound it too, by the counts the code object's own note states; when the
        // note cannot be read, one workgroup per CU less than the query says
        const uint32_t regs = w->jit_entry ? w->jit_entry->vgprs : 0, sregs = w->jit_entry ? w->jit_entry->sgprs : 0;
        per_cu = (regs && sregs) ? std::min(per_cu, resident_wgs_per_cu(regs, sregs)) : per_cu - 1;
        w->sp_sregs = (int)sregs;
        w->sp_regs = (int)regs; w->sp_per_cu = per_cu;
        const uint64_t max_wgs = (uint64_t)std::max(per_cu, 0) * (uint64_t)w->n_cu;
        const uint32_t grid = 8u * ((w->sp_tiles + 7u) / 8u);
        if (grid > max_wgs)
            return w->fail(GGRS_E_CAPACITY, "a world whose systems spawn on the device runs as ONE resident launch: %u workgroups are needed for %llu slots, the device holds %llu of this kernel (at most %llu slots)",
                           grid, (unsigned long long)w->capacity, (unsigned long long)max_wgs, (unsigned long long)(max_wgs / 8 * 8 * 256));
        HIPCHK(w, hipMalloc((void**)&w->d_sp_sums, (3 * (size_t)w->sp_tiles + 32) * 8));                 // the mailbox words {epoch, value}: counts, prefixes, done per tile; total; go
        HIPCHK(w, hipMemsetAsync(w->d_sp_sums, 0, (3 * (size_t)w->sp_tiles + 32) * 8, w->stream));
        w->sp_epoch = 0xF0000000u - 8u * (2u * MAX_TICK_STEPS + 2u) + 1u;      // (like jiffies: every world crosses the epochs' start-over on its 9th launch, so that path is run by every test and session)
        HIPCHK(w, hipMalloc((void**)&w->d_sp_prec, 2 * (size_t)std::max<uint64_t>(w->cap_pad, (uint64_t)w->sp_tiles * 256u) * 64));       // the parents' records, one set per step parity
        HIPCHK(w, hipMalloc((void**)&w->d_sp_link, (size_t)w->cap_pad * 16));
        HIPCHK(w, hipHostMalloc((void**)&w->h_sp_len, (2 + MAX_TICK_SAVES) * 8, hipHostMallocMapped));
        HIPCHK(w, hipHostGetDevicePointer((void**)&w->d_sp_len, (void*)w->h_sp_len, 0));
        for (int k = 0; k < 2 + MAX_TICK_SAVES; ++k) w->h_sp_len[k] = 0;
    }
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->sealed = true;
    return GGRS_OK;
}

}  // namespace
