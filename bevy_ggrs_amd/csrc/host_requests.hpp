// host_requests.hpp -- one launch per request: SaveWorld / LoadWorld / AdvanceWorld as the reference schedules them, the ring,
// spawn bookkeeping, the host-side checksum fold and tracing.  Part of the single translation unit ggrs_hip.hip.
#pragma once

namespace {

Header header_of(const ggrs_world* w) {
    Header h; memset(&h, 0, sizeof h);
    h.len = w->len; h.frame = w->frame;
    return h;
}

FinalizeArgs no_finalize() { FinalizeArgs f; memset(&f, 0, sizeof f); return f; }

// `want`: the versions of the state being copied (src's own, or the logical live state's).  Only the rows whose column differs
// in dst move; the masks and the header always do.  dst holds `want` afterwards.
int launch_copy(ggrs_world* w, const Block& src, Block& dst, const std::vector<ver_t>& want, uint64_t len, uint32_t cls, const FinalizeArgs& fin) {
    const uint64_t cover = std::max(std::max(src.dirty_len, dst.dirty_len), len);
    const uint32_t g = std::max(1u, tiles_for(cover));
    CopyPlan plan = w->plan;
    plan.n_rows = 0; plan.n_wide = 0;
    uint64_t bytes_per_tile = 0;
    const bool same = src.ptr == dst.ptr;                             // depth 0: the copy only carries the fold
    for (uint32_t r = 0; r < w->plan.n_rows && !same; ++r) {
        if (!ver_differs(w, dst, want, w->row_col[r])) continue;
        plan.row[plan.n_rows++] = w->plan.row[r];
        if (r < w->plan.n_wide) plan.n_wide = plan.n_rows;
        bytes_per_tile += w->plan.row[r].bytes;
    }
    {
        ProfScope ps(w, cls, 2 * bytes_per_tile * tiles_for(len));
        if (w->nt_copy)
            hipLaunchKernelGGL((k_copy_state<true>), dim3(g), dim3(TPB), 0, w->stream, (const uint8_t*)src.ptr, dst.ptr, plan, len, header_of(w), fin);
        else
            hipLaunchKernelGGL((k_copy_state<false>), dim3(g), dim3(TPB), 0, w->stream, (const uint8_t*)src.ptr, dst.ptr, plan, len, header_of(w), fin);
    }
    HIPCHK(w, hipGetLastError());
    if (!same) { dst.dirty_len = src.dirty_len; dst.len = len; dst.ver = want; }
    return GGRS_OK;
}

// Generic checksum pass over the live block -> partials
int launch_checksum(ggrs_world* w) {
    const uint32_t g = std::max(1u, tiles_for(w->live.dirty_len));
    CksArgs a = w->cks_args; a.state = w->live.ptr;
    {
        ProfScope ps(w, GGRS_KERNEL_CHECKSUM);
        hipLaunchKernelGGL(k_checksum, dim3(g, std::max(1u, a.n_cks)), dim3(TPB), 0, w->stream, a, (const UnitDesc*)w->d_units);
    }
    HIPCHK(w, hipGetLastError());
    w->pending_valid = true; w->pending_parts = g;
    return GGRS_OK;
}

FinalizeArgs finalize_args(ggrs_world* w, uint32_t result_idx) {
    FinalizeArgs f; memset(&f, 0, sizeof f);
    f.parts = w->d_parts; f.part_cnt = w->cks_args.part_cnt;
    f.n_cks = w->cks_args.n_cks; f.part_stride = w->part_stride; f.n_parts = w->pending_parts; f.enabled = 1;
    f.total_len = w->len;
    f.out = w->d_results + 2 * (uint64_t)result_idx;
    f.live_hdr = (Header*)w->live.ptr;
    return f;
}

// ---- ring: exact mirror of GgrsSnapshots::{push,confirm,rollback} over slot indices
void ring_pop_front(ggrs_world* w) { w->free_slots.push_back(w->ring_slot.front()); w->ring_slot.pop_front(); w->ring_frame.pop_front(); }
void ring_pop_back(ggrs_world* w) { w->free_slots.push_back(w->ring_slot.back()); w->ring_slot.pop_back(); w->ring_frame.pop_back(); }

void ring_confirm(ggrs_world* w, int32_t confirmed) {          // mod.rs:185-202
    while (!w->ring_frame.empty() && w->ring_frame.back() < confirmed) ring_pop_back(w);
}
int ring_push(ggrs_world* w, int32_t frame, int* slot_out) {    // mod.rs:147-181
    while (!w->ring_frame.empty()) {
        const int32_t current = w->ring_frame.front();
        const uint32_t ad = current >= frame ? (uint32_t)current - (uint32_t)frame : (uint32_t)frame - (uint32_t)current;
        const bool wrapped = ad > (UINT32_MAX / 2);
        if ((current >= frame && !wrapped) || (frame >= current && wrapped)) ring_pop_front(w); else break;
    }
    // evict from the back first so the new slot can reuse the oldest block (same end state as
    // push_front followed by pop_back while len > depth)
    while (!w->ring_frame.empty() && w->ring_frame.size() + 1 > w->depth) ring_pop_back(w);
    if (w->depth == 0) { *slot_out = -1; return GGRS_OK; }
    if (w->free_slots.empty()) return w->fail(GGRS_E_INVALID, "ring depth %zu exceeds provisioned max_depth %u", w->depth, w->max_depth);
    const int s = w->free_slots.back(); w->free_slots.pop_back();
    w->ring_slot.push_front(s); w->ring_frame.push_front(frame);
    *slot_out = s;
    return GGRS_OK;
}
bool ring_rollback(ggrs_world* w, int32_t frame) {             // mod.rs:210-226
    for (;;) {
        if (w->ring_frame.empty()) return false;
        if (w->ring_frame.front() != frame) ring_pop_front(w); else return true;
    }
}

// ---- RollbackDespawned (snapshot/despawn.rs)
inline uint32_t blocks_for_slots(uint64_t n) { return (uint32_t)((align_up(std::max<uint64_t>(n, 1), 64) + TPB - 1) / TPB); }

// LoadWorldSystems::EntityResurrect + the non-rollback side of the entity reconcile; must be queued
// before the kernel that overwrites the live liveness mask.  w->frame is already the loaded frame.
int launch_load_reconcile(ggrs_world* w, const Block& snap) {
    if (!w->has_nr && !w->marks_possible) return GGRS_OK;
    ReconcileArgs a; memset(&a, 0, sizeof a);
    a.live = w->live.ptr; a.snap = snap.ptr; a.off_alive = w->off_alive; a.dm = w->marks; a.frame = w->frame;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback) a.nr_present_off[a.n_nr++] = w->off_present[c];
    const uint64_t cover = std::max(std::max(w->live.dirty_len, snap.dirty_len), w->len);
    a.n_slots_pad64 = align_up(std::max<uint64_t>(cover, 1), 64);
    hipLaunchKernelGGL(k_load_reconcile, dim3(blocks_for_slots(cover)), dim3(TPB), 0, w->stream, a);
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}
// AdvanceWorldSystems::DespawnConfirmed (despawn.rs:89-112), with its Local<ConfirmedFrameCount>
int step_despawn_confirmed(ggrs_world* w) {
    if (w->confirmed == w->dc_local) return GGRS_OK;          // "No work necessary"
    w->dc_local = w->confirmed;
    if (!w->marks_possible) return GGRS_OK;                   // no marker was ever set: nothing to free
    const uint64_t cover = std::max(w->live.dirty_len, w->len);
    hipLaunchKernelGGL(k_despawn_confirmed, dim3(blocks_for_slots(cover)), dim3(TPB), 0, w->stream, w->live.ptr, w->marks,
                       w->confirmed, align_up(std::max<uint64_t>(cover, 1), 64));
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

// ---- SaveWorld
int do_save(ggrs_world* w, uint32_t result_idx) {
    int rc = seal(w); if (rc) return rc;
    // SaveWorldSystems::Checksum -> ChecksumPlugin::update
    if (!w->pending_valid) { rc = launch_checksum(w); if (rc) return rc; }
    // ChecksumPlugin::update (fold) runs inside workgroup 0 of the snapshot copy kernel
    const FinalizeArgs fin = finalize_args(w, result_idx);
    // SaveWorldSystems::Snapshot: sync_depth (caller) -> discard_old_snapshots -> save
    if (w->has_confirmed) ring_confirm(w, w->confirmed);
    int s = -1;
    rc = ring_push(w, w->frame, &s); if (rc) return rc;
    if (s >= 0) { rc = launch_copy(w, w->live, w->slots[s], w->cur_ver, w->len, GGRS_KERNEL_SAVE, fin); if (rc) return rc; }
    else {
        // depth 0: nothing is stored, but the checksum is still due -> copy live onto itself
        // (no rows) just to run the fold
        rc = launch_copy(w, w->live, w->live, w->cur_ver, 0, GGRS_KERNEL_SAVE, fin); if (rc) return rc;
    }
    return GGRS_OK;
}

// ---- LoadWorld
int do_load(ggrs_world* w, int32_t frame) {
    int rc = seal(w); if (rc) return rc;
    w->frame = frame;                                           // schedule_systems.rs:244-247
    if (!ring_rollback(w, frame))
        return w->fail(GGRS_E_NO_SNAPSHOT, "Could not rollback to %d: no snapshot at that moment could be found.", frame);
    Block& s = w->slots[w->ring_slot.front()];
    rc = launch_load_reconcile(w, s); if (rc) return rc;        // LoadWorldSystems::EntityResurrect
    // entity.rs:55-99 + component_snapshot.rs:95-123 + RollbackOrdered restore (mod.rs:342):
    // masks, columns and len of the live block := the snapshot's
    w->len = s.len;
    rc = launch_copy(w, s, w->live, s.ver, s.len, GGRS_KERNEL_LOAD, no_finalize()); if (rc) return rc;
    w->cur_ver = w->live.ver;                                   // the logical live state IS the snapshot now
    w->pending_valid = false;
    return GGRS_OK;
}

// ---- spawn bookkeeping shared by the API call and the in-schedule spawn system
int set_masks_for_range(ggrs_world* w, uint64_t first, uint64_t count, uint64_t comp_mask) {
    if (count == 0) return GGRS_OK;
    MaskOffs mo, mc; uint32_t n = 0, nc = 0;
    mo.off[n++] = w->off_alive;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if ((comp_mask >> c) & 1ULL) mo.off[n++] = w->off_present[c];
    // a fresh entity carries no RollbackDespawned marker and only the non-rollback components of its
    // bundle (those masks are live-only: no LoadWorld copy ever cleans them)
    if (w->marks_possible) mc.off[nc++] = w->marks.off_disabled;
    for (uint32_t c = 0; c < w->comps.size(); ++c) if (w->comps[c].no_rollback && !((comp_mask >> c) & 1ULL)) mc.off[nc++] = w->off_present[c];
    const uint64_t words = ((first + count - 1) >> 6) - (first >> 6) + 1;
    hipLaunchKernelGGL(k_set_mask_range, dim3((uint32_t)((words + TPB - 1) / TPB)), dim3(TPB), 0, w->stream,
                       w->live.ptr, first, count, n, mo, nc, mc);
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

// Host <-> device copy of `count` words of one column starting at slot `first` (tile-major columns: a head
// piece, the full tiles as one pitched 2D copy, a tail piece; plain arrays: one copy).
int copy_column(ggrs_world* w, uint32_t col, uint64_t first, uint64_t count, void* host, bool to_device) {
    if (count == 0) return GGRS_OK;
    const uint32_t wb = w->col_wb[col], ts = w->col_ts[col];
    uint8_t* h = (uint8_t*)host;
    auto dev = [&](uint64_t slot) { return w->live.ptr + col_at(w->col_off[col], ts, wb, slot); };
    auto piece = [&](uint64_t slot, uint64_t n) -> hipError_t {
        return to_device ? hipMemcpyAsync(dev(slot), h + (slot - first) * wb, n * wb, hipMemcpyHostToDevice, w->stream)
                         : hipMemcpyAsync(h + (slot - first) * wb, dev(slot), n * wb, hipMemcpyDeviceToHost, w->stream);
    };
    if (ts == LAYOUT_TILE * wb) { HIPCHK(w, piece(first, count)); return GGRS_OK; }
    uint64_t s0 = first, end = first + count;
    if (s0 % LAYOUT_TILE) { const uint64_t n = std::min<uint64_t>(end - s0, LAYOUT_TILE - s0 % LAYOUT_TILE); HIPCHK(w, piece(s0, n)); s0 += n; }
    const uint64_t full = (end - s0) / LAYOUT_TILE;
    if (full) {
        const size_t width = (size_t)LAYOUT_TILE * wb;
        if (to_device) HIPCHK(w, hipMemcpy2DAsync(dev(s0), ts, h + (s0 - first) * wb, width, width, full, hipMemcpyHostToDevice, w->stream));
        else HIPCHK(w, hipMemcpy2DAsync(h + (s0 - first) * wb, width, dev(s0), ts, width, full, hipMemcpyDeviceToHost, w->stream));
        s0 += full * LAYOUT_TILE;
    }
    if (s0 < end) HIPCHK(w, piece(s0, end - s0));
    return GGRS_OK;
}

int fill_defaults(ggrs_world* w, uint32_t c, uint64_t first, uint64_t count) {
    const Comp& cc = w->comps[c];
    for (uint32_t k = 0; k < cc.n_words; ++k) {
        uint64_t v = 0; memcpy(&v, &cc.defaults[(size_t)k * cc.word_bytes], cc.word_bytes);   // 1, 2, 4 or 8 bytes
        hipLaunchKernelGGL(k_fill_col, dim3((uint32_t)((count + TPB - 1) / TPB)), dim3(TPB), 0, w->stream,
                           w->live.ptr, w->col_off[cc.col_base + k], w->col_ts[cc.col_base + k], cc.word_bytes, first, count, v);
    }
    HIPCHK(w, hipGetLastError());
    return GGRS_OK;
}

// The spawn-payload ring (host_world.hpp): n contiguous bytes (16-byte granules), or false when the region in front of the oldest uncollected
// batch's payloads is too small (the caller then waits for the stream, which frees everything).
bool stage_ring_alloc(ggrs_world* w, uint64_t n, uint64_t* off) {
    n = (n + 15u) & ~15ull;
    uint64_t& head = w->stage_used; const uint64_t tail = w->stage_tail, cap = w->stage_bytes;
    if (n > cap) return false;
    if (head >= tail) {
        if (head + n <= cap) { *off = head; head += n; return true; }
        if (n < tail) { *off = 0; head = n; return true; }              // wrap: the front of the buffer has been consumed
        return false;
    }
    if (head + n < tail) { *off = head; head += n; return true; }
    return false;
}
// every launch that could read a staged payload has completed (the stream was waited for / every batch collected).  Offsets handed out before
// this point are dead: stage_gen tells whoever cached one (a request list's payload dedup)
void stage_ring_reset(ggrs_world* w) { w->stage_used = w->stage_tail = 0; ++w->stage_gen; for (auto& b : w->pending) b.stage_end = 0; }
// vx and vy of one particles spawn, side by side in the ring (2 x n floats must fit: the default ring of 8 MiB takes a spawn of 1 M particles;
// GGRS_STAGE_BYTES), copied to the device twin for the unfused spawn kernel
int stage_pair(ggrs_world* w, const float* vx, const float* vy, uint64_t n, float** dvx, float** dvy) {
    if (2 * n * 4 > w->stage_bytes) return w->fail(GGRS_E_CAPACITY, "spawn payload of 2 x %llu floats exceeds the staging buffer (%llu bytes: GGRS_STAGE_BYTES)", (unsigned long long)n, (unsigned long long)w->stage_bytes);
    uint64_t off = 0;
    if (!stage_ring_alloc(w, 2 * n * 4, &off)) {
        HIPCHK(w, hipStreamSynchronize(w->stream));
        stage_ring_reset(w);
        (void)stage_ring_alloc(w, 2 * n * 4, &off);
    }
    memcpy(w->h_stage + off, vx, n * 4); memcpy(w->h_stage + off + n * 4, vy, n * 4);
    HIPCHK(w, hipMemcpyAsync(w->d_stage + off, w->h_stage + off, 2 * n * 4, hipMemcpyHostToDevice, w->stream));
    *dvx = reinterpret_cast<float*>(w->d_stage + off); *dvy = reinterpret_cast<float*>(w->d_stage + off + n * 4);
    return GGRS_OK;
}

uint32_t dt_bits_for_frame(uint64_t fps, int32_t frame) {
    // GgrsTimePlugin::update (time.rs:63-87): runtime = frame * 1e9 / fps ns; the clock's previous
    // elapsed is runtime(frame-1) (restored by its own snapshot on load, time.rs:111), and
    // Time::delta_secs = Duration::as_secs_f32 = secs as f32 + nanos as f32 / 1e9 as f32.
    const uint64_t f = (uint64_t)(int64_t)frame;
    const uint64_t d = f * 1000000000ULL / fps - (f - 1) * 1000000000ULL / fps;
    const uint64_t secs = d / 1000000000ULL; const uint32_t nanos = (uint32_t)(d % 1000000000ULL);
    volatile float a = (float)secs;
    volatile float b = (float)nanos / (float)1000000000u;
    const float r = a + b;
    uint32_t bits; memcpy(&bits, &r, 4);
    return bits;
}

// Commands are deferred: spawns materialise after every system of the frame ran (set.rs:118-134).
// `pressed`: any player's input (its first byte: the whole input of a Config<Input = u8> session) carries the system's bit
inline bool spawn_pressed(const ggrs_world* w, const ggrs_system_desc& s, const uint8_t* inputs, uint32_t n_inputs) {
    for (uint32_t k = 0; k < n_inputs; ++k) if (inputs[(size_t)k * w->input_bytes] & (uint8_t)s.iparam[1]) return true;
    return false;
}
int run_spawn_systems(ggrs_world* w, const uint8_t* inputs, uint32_t n_inputs, uint64_t spawn_count,
                      const float* spawn_vx, const float* spawn_vy) {
    int rc = GGRS_OK;
    const uint32_t n_cks = w->cks_args.n_cks;
    uint64_t* part_cnt = w->cks_args.part_cnt;
    for (auto& s : w->systems) {
        if (s.kind != GGRS_SYS_PARTICLES_SPAWN) continue;
        if (!spawn_pressed(w, s, inputs, n_inputs) || spawn_count == 0) continue;      // spawn_pressed, particles.rs:254-256
        if (w->len + spawn_count > w->capacity) return w->fail(GGRS_E_CAPACITY, "spawn of %llu exceeds capacity %llu", (unsigned long long)spawn_count, (unsigned long long)w->capacity);
        const uint32_t cT = s.comp[0], cV = s.comp[1], cL = s.comp[2];
        const Comp& T = w->comps[cT]; const Comp& V = w->comps[cV]; const Comp& L = w->comps[cL];
        const uint64_t first = w->len;
        float *dvx = nullptr, *dvy = nullptr;
        rc = stage_pair(w, spawn_vx, spawn_vy, spawn_count, &dvx, &dvy); if (rc) return rc;
        rc = fill_defaults(w, cT, first, spawn_count); if (rc) return rc;
        SpawnArgs a; memset(&a, 0, sizeof a);
        a.state = w->live.ptr;
        for (int k = 0; k < 3; ++k) {
            a.off_t[k] = w->col_off[T.col_base + k]; a.off_v[k] = w->col_off[V.col_base + k];
            memcpy(&a.t_default[k], &T.defaults[(size_t)(w->fused_ok ? w->f_tw + k : k) * 4], 4);
        }
        a.off_ttl = w->col_off[L.col_base + 0];
        a.ts = w->ts;
        a.vx = dvx; a.vy = dvy; a.first = first; a.count = spawn_count; a.ttl = (uint64_t)s.iparam[0];
        const uint32_t gs = (uint32_t)((spawn_count + TPB - 1) / TPB);
        const bool keep = w->pending_valid && (w->pending_parts + gs <= w->part_stride);
        // partial slots appended after the step's (scratch at the tail when partials are not kept)
        const uint32_t pbase = keep ? w->pending_parts : (w->part_stride - std::min(gs, w->part_stride));
        uint64_t* scratch = w->d_parts;   // column 0 exists whenever n_cks > 0; else counts column
        a.part_T = a.part_V = (n_cks ? scratch : part_cnt) + pbase;
        a.cks_T = a.cks_V = 0;
        if (keep) {
            for (uint32_t k = 0; k < n_cks; ++k) {
                if ((int)w->cks_comp[k] == w->f_T && w->f_cksT) { a.part_T = w->d_parts + (uint64_t)k * w->part_stride + pbase; a.cks_T = 1; }
                if ((int)w->cks_comp[k] == w->f_V && w->f_cksV) { a.part_V = w->d_parts + (uint64_t)k * w->part_stride + pbase; a.cks_V = 1; }
            }
        }
        a.part_cnt = part_cnt + pbase;
        if (gs > w->part_stride) return w->fail(GGRS_E_CAPACITY, "spawn too large for partial buffer");
        hipLaunchKernelGGL(k_spawn_particles, dim3(gs), dim3(TPB), 0, w->stream, a);
        HIPCHK(w, hipGetLastError());
        rc = set_masks_for_range(w, first, spawn_count, (1ULL << cT) | (1ULL << cV) | (1ULL << cL)); if (rc) return rc;
        ver_touch_comp(w, cT); ver_touch_comp(w, cV); ver_touch_comp(w, cL); ver_sync_live(w);   // new rows in every column of the bundle
        live_tags_lost_comp(w, cT); live_tags_lost_comp(w, cV); live_tags_lost_comp(w, cL);
        w->len += spawn_count;
        w->live.dirty_len = std::max(w->live.dirty_len, w->len);
        if (keep) w->pending_parts += gs; else w->pending_valid = false;
    }
    return GGRS_OK;
}

template <bool CT, bool CV>
void launch_step_fused(ggrs_world* w, const StepArgs& a, uint32_t g) {
    hipLaunchKernelGGL((k_particles_step<true, true, CT, CV>), dim3(g), dim3(TPB), 0, w->stream, a);
}

// The generated translation unit: ABI text, the entity view, the user's source, and a one-slot-per-lane kernel whose
// binding count and word widths are compile-time constants (so e.w[] lives in registers, not scratch).
std::string custom_source(const ggrs_world* w, const ggrs_world::Custom& c, const char* user) {
    std::string s;
    s += GGRS_CUSTOM_ABI_TEXT;
    char buf[256];
    snprintf(buf, sizeof buf, "static_assert(sizeof(GgrsCustomArgs) == %zu, \"host/device argument block mismatch\");\n", sizeof(GgrsCustomArgs));
    s += buf;
    s += GGRS_FRAME_TEXT;
    s += GGRS_ENTITY_TEXT;
    s += "#line 1 \"ggrs_system\"\n";
    s += user;
    snprintf(buf, sizeof buf, "\n#line 1 \"ggrs_custom_kernel\"\n#define GGRS_N_BIND %u\n#define GGRS_N_PRES %u\n", c.n_bind, c.n_pres);
    s += buf;
    s += "__device__ constexpr int GGRS_WB[8] = {";
    for (uint32_t i = 0; i < 8; ++i) { snprintf(buf, sizeof buf, "%u,", i < c.n_bind ? w->comps[c.comp[i]].word_bytes : 4u); s += buf; }
    s += "};\n";
    snprintf(buf, sizeof buf, "#define GGRS_LT_SHIFT %d\n", LT_SHIFT);
    s += buf;
    s += "extern \"C\" __global__ __launch_bounds__(256) void ggrs_custom_kernel(GgrsCustomArgs a) {\n"
         "    __shared__ unsigned char s_in[272];                   // PlayerInputs: a handle read from a component may index them\n"
         "    for (int i = threadIdx.x; i < 272; i += 256) s_in[i] = a.fr.in[i];\n"
         "    __syncthreads();\n"
         "    GgrsFrame fr; fr.dt = a.fr.dt; fr.frame = a.fr.frame; fr.n_inputs = a.fr.n_inputs; fr.input_bytes = a.fr.input_bytes;\n"
         "    fr.input.p = s_in; fr.input.ib = a.fr.input_bytes; fr.status = s_in + a.fr.status_off;\n"
         "    for (int k = 0; k < 4; ++k) fr.fparam[k] = a.fr.fparam[k];\n"
         "    fr.iparam[0] = a.fr.iparam[0]; fr.iparam[1] = a.fr.iparam[1];\n"
         "    const ggrs_u64 e = (ggrs_u64)blockIdx.x * 256 + threadIdx.x;\n"
         "    if (e >= a.len_pad64) return;                         // whole waves only (len padded to 64)\n"
         "    const ggrs_u64 aw = *reinterpret_cast<const ggrs_u64*>(a.state + a.off_alive + (e >> 6) * 8);\n"
         "    ggrs_u64 on = aw;\n"
         "    #pragma unroll\n"
         "    for (int p = 0; p < GGRS_N_PRES; ++p) on &= *reinterpret_cast<const ggrs_u64*>(a.state + a.off_present[p] + (e >> 6) * 8);\n"
         "    bool alive = (aw >> (e & 63)) & 1ULL;\n"
         "    int kill = 0;\n"
         "    if ((on >> (e & 63)) & 1ULL) {\n"
         "        GgrsEntity ent; ent.slot = e; ent.kill = 0;\n"
         "        unsigned char* at[8];\n"
         "        #pragma unroll\n"
         "        for (int i = 0; i < GGRS_N_BIND; ++i) {\n"
         "            at[i] = a.state + a.col_off[i] + (e >> GGRS_LT_SHIFT) * a.ts[i] + (e & ((1ULL << GGRS_LT_SHIFT) - 1)) * GGRS_WB[i];\n"
         "            ent.w[i] = GGRS_WB[i] == 8 ? *reinterpret_cast<const ggrs_u64*>(at[i]) : (ggrs_u64)*reinterpret_cast<const ggrs_u32*>(at[i]);\n"
         "        }\n"
         "        ggrs_system(ent, fr);\n"
         "        #pragma unroll\n"
         "        for (int i = 0; i < GGRS_N_BIND; ++i) {\n"
         "            if (GGRS_WB[i] == 8) *reinterpret_cast<ggrs_u64*>(at[i]) = ent.w[i];\n"
         "            else *reinterpret_cast<ggrs_u32*>(at[i]) = (ggrs_u32)ent.w[i];\n"
         "        }\n"
         "        kill = ent.kill;\n"
         "    }\n"
         "    if (kill) alive = false;\n"
         "    const ggrs_u64 nw = __builtin_amdgcn_ballot_w64(alive);\n"
         "    if ((threadIdx.x & 63u) == 0 && nw != aw) *reinterpret_cast<ggrs_u64*>(a.state + a.off_alive + (e >> 6) * 8) = nw;\n"
         "    if (a.defer) {                                        // despawn_rollback on an unconfirmed frame: RollbackDespawned(frame)\n"
         "        const bool mark = kill == 2;\n"
         "        const ggrs_u64 kw = __builtin_amdgcn_ballot_w64(mark);\n"
         "        if (mark) *reinterpret_cast<int*>(a.state + a.off_dframe + e * 4) = a.fr.frame;\n"
         "        if ((threadIdx.x & 63u) == 0 && kw) *reinterpret_cast<ggrs_u64*>(a.state + a.off_disabled + (e >> 6) * 8) |= kw;\n"
         "    }\n"
         "}\n";
    return s;
}

// PlayerInputs of one AdvanceFrame in the layout the device sees: n x input_bytes bytes, then (at max_players x input_bytes) one InputStatus byte per player
inline void pack_inputs(const ggrs_world* w, const ggrs_request& r, unsigned char* dst /* >= max_players * (input_bytes + 1) */) {
    const uint32_t ib = w->input_bytes, mp = w->max_players;
    memset(dst, 0, (size_t)mp * (ib + 1));
    const uint32_t n = std::min<uint32_t>(r.n_inputs, mp);
    if (n && r.inputs) memcpy(dst, r.inputs, (size_t)n * ib);
    if (n && r.status) memcpy(dst + (size_t)mp * ib, r.status, n);       // NULL: every input Confirmed (0)
}

int launch_custom(ggrs_world* w, const ggrs_system_desc& s, uint32_t dt_bits, const ggrs_request& r) {
    const ggrs_world::Custom& c = w->customs[s.comp[0]];
    GgrsCustomArgs a; memset(&a, 0, sizeof a);
    a.state = w->live.ptr; a.off_alive = w->off_alive;
    a.off_disabled = w->marks.off_disabled; a.off_dframe = w->marks.off_dframe;
    a.len_pad64 = align_up(w->len, 64);
    for (uint32_t p = 0; p < c.n_pres; ++p) a.off_present[p] = w->off_present[c.pres_comp[p]];
    for (uint32_t i = 0; i < c.n_bind; ++i) {
        const uint32_t col = w->comps[c.comp[i]].col_base + c.word[i];
        a.col_off[i] = w->col_off[col]; a.ts[i] = w->col_ts[col];
    }
    // despawn_rollback (despawn.rs:129-142): only an unconfirmed frame defers the despawn.  Whether the source REACHES the call is
    // not known to the host, so the markers are assumed possible whenever deferral is on and the source names it (Custom::may_defer).
    a.defer = (w->confirmed < w->frame && c.may_defer) ? 1 : 0;
    if (a.defer) w->marks_possible = true;
    memcpy(&a.fr.dt, &dt_bits, 4);
    a.fr.frame = w->frame;
    a.fr.n_inputs = std::min<uint32_t>(r.n_inputs, w->max_players);
    a.fr.input_bytes = w->input_bytes; a.fr.status_off = w->max_players * w->input_bytes;
    pack_inputs(w, r, a.fr.in);
    for (int k = 0; k < 4; ++k) a.fr.fparam[k] = s.fparam[k];
    a.fr.iparam[0] = s.iparam[0]; a.fr.iparam[1] = s.iparam[1];
    const uint32_t gx = (uint32_t)((a.len_pad64 + 255) / 256);
    if (gx == 0) return GGRS_OK;
    void* params[] = {&a};
    HIPCHK(w, hipModuleLaunchKernel(c.fn, gx, 1, 1, 256, 1, 1, 0, w->stream, params, nullptr));
    return GGRS_OK;
}

// ---- AdvanceWorld
int do_advance(ggrs_world* w, const ggrs_request& r) {
    uint32_t dt_bits = r.dt_bits; const uint8_t* inputs = r.inputs; const uint32_t n_inputs = r.n_inputs;
    const uint64_t spawn_count = r.spawn_count; const float* spawn_vx = r.spawn_vx; const float* spawn_vy = r.spawn_vy;
    int rc = seal(w); if (rc) return rc;
    w->frame += 1;                                              // schedule_systems.rs:254-259
    if (dt_bits == 0) dt_bits = dt_bits_for_frame(w->fps, w->frame);
    rc = step_despawn_confirmed(w); if (rc) return rc;         // AdvanceWorldSystems::DespawnConfirmed, before Main
    ver_step(w); ver_sync_live(w);                              // the systems write the live block in place
    w->live.tag_ok = 0;                                         // (per-request kernels keep no value tags)
    const uint32_t g = tiles_for(w->len);
    const uint32_t n_cks = w->cks_args.n_cks;
    uint64_t* part_cnt = w->cks_args.part_cnt;
    w->pending_valid = false;

    auto step_args = [&](const ggrs_system_desc* upd, const ggrs_system_desc* ttl) {
        StepArgs a; memset(&a, 0, sizeof a);
        a.state = w->live.ptr; a.off_alive = w->off_alive; a.dt_bits = dt_bits; a.ts = w->ts;
        if (upd) {
            const Comp& T = w->comps[upd->comp[0]]; const Comp& V = w->comps[upd->comp[1]];
            a.off_pT = w->off_present[upd->comp[0]]; a.off_pV = w->off_present[upd->comp[1]];
            for (int k = 0; k < 3; ++k) {
                a.off_t[k] = w->col_off[T.col_base + upd->word[0] + k];
                a.off_v[k] = w->col_off[V.col_base + upd->word[1] + k];
                a.g[k] = upd->fparam[k];
            }
        }
        if (ttl) {
            const Comp& L = w->comps[ttl->comp[0]];
            a.off_pL = w->off_present[ttl->comp[0]];
            a.off_ttl = w->col_off[L.col_base + ttl->word[0]];
        }
        return a;
    };

    if (g > 0) {
        if (w->fused_ok) {
            const ggrs_system_desc *upd = nullptr, *ttl = nullptr;
            for (auto& s : w->systems) { if (s.kind == GGRS_SYS_PARTICLES_UPDATE) upd = &s; if (s.kind == GGRS_SYS_TTL_DESPAWN) ttl = &s; }
            StepArgs a = step_args(upd, ttl);
            // partial columns in checksum-spec order
            for (uint32_t k = 0; k < n_cks; ++k) {
                if ((int)w->cks_comp[k] == w->f_T) a.part_T = w->d_parts + (uint64_t)k * w->part_stride;
                if ((int)w->cks_comp[k] == w->f_V) a.part_V = w->d_parts + (uint64_t)k * w->part_stride;
            }
            a.part_cnt = part_cnt;
            ProfScope ps(w, GGRS_KERNEL_ADVANCE);
            const bool ck = w->fused_cks;
            if (ck && w->f_cksT && w->f_cksV) launch_step_fused<true, true>(w, a, g);
            else if (ck && w->f_cksT) launch_step_fused<true, false>(w, a, g);
            else if (ck && w->f_cksV) launch_step_fused<false, true>(w, a, g);
            else if (ck) {   // no component checksums at all: still produce the live count
                hipLaunchKernelGGL((k_particles_step<true, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
            } else hipLaunchKernelGGL((k_particles_step<true, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
            if (ck && (w->f_cksT || w->f_cksV)) { w->pending_valid = true; w->pending_parts = g; }
        } else {
            for (auto& s : w->systems) {
                ProfScope ps(w, GGRS_KERNEL_ADVANCE);
                switch (s.kind) {
                case GGRS_SYS_PARTICLES_UPDATE: {
                    StepArgs a = step_args(&s, nullptr);
                    hipLaunchKernelGGL((k_particles_step<true, false, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_TTL_DESPAWN: {
                    StepArgs a = step_args(nullptr, &s);
                    hipLaunchKernelGGL((k_particles_step<false, true, false, false>), dim3(g), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_ADD_U32: {
                    const Comp& C = w->comps[s.comp[0]];
                    hipLaunchKernelGGL(k_add_u32, dim3((uint32_t)((w->len + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, w->live.ptr,
                                       w->off_alive, w->off_present[s.comp[0]], w->col_off[C.col_base + s.word[0]], w->col_ts[C.col_base + s.word[0]], (uint32_t)s.iparam[0], w->len);
                } break;
                case GGRS_SYS_SAT_SUB_DESPAWN: {
                    const Comp& C = w->comps[s.comp[0]];
                    const uint64_t lp = align_up(w->len, 64);
                    // despawn_rollback (despawn.rs:129-142): only an unconfirmed frame defers the despawn
                    const int defer = (s.iparam[1] == GGRS_DESPAWN_ROLLBACK && w->confirmed < w->frame) ? 1 : 0;
                    if (defer) w->marks_possible = true;
                    hipLaunchKernelGGL(k_sat_sub_despawn, dim3((uint32_t)((lp + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, w->live.ptr,
                                       w->off_alive, w->off_present[s.comp[0]], w->col_off[C.col_base + s.word[0]], w->col_ts[C.col_base + s.word[0]], (uint32_t)s.iparam[0], lp,
                                       defer, w->frame, w->marks);
                } break;
                case GGRS_SYS_BOX_MOVE: {
                    const Comp& T = w->comps[s.comp[0]]; const Comp& V = w->comps[s.comp[1]]; const Comp& P = w->comps[s.comp[2]];
                    BoxMoveArgs a; memset(&a, 0, sizeof a);
                    a.state = w->live.ptr; a.off_alive = w->off_alive;
                    a.off_pT = w->off_present[s.comp[0]]; a.off_pV = w->off_present[s.comp[1]]; a.off_pP = w->off_present[s.comp[2]];
                    for (int k = 0; k < 3; ++k) { a.off_t[k] = w->col_off[T.col_base + s.word[0] + k]; a.off_v[k] = w->col_off[V.col_base + s.word[1] + k]; }
                    a.off_handle = w->col_off[P.col_base + s.word[2]];
                    a.ts_t = w->col_ts[T.col_base + s.word[0]]; a.ts_v = w->col_ts[V.col_base + s.word[1]]; a.ts_handle = w->col_ts[P.col_base + s.word[2]];
                    a.len = w->len; a.dt_bits = dt_bits;
                    // FRICTION.powf(dt) (box_game.rs:189-195): Rust lowers f32::powf to the platform libm's powf
                    float dtf; memcpy(&dtf, &dt_bits, 4);
                    const float fp = powf(s.fparam[2], dtf);
                    memcpy(&a.friction_pow_bits, &fp, 4);
                    a.accel = s.fparam[0]; a.max_speed = s.fparam[1]; a.half_width = s.fparam[3];
                    a.n_inputs = std::min<uint32_t>(n_inputs, 16);
                    for (uint32_t k = 0; k < a.n_inputs; ++k) a.inputs[k] = inputs[(size_t)k * w->input_bytes];      // box_game's input is one byte (box_game.rs:13-16)
                    hipLaunchKernelGGL(k_box_move, dim3((uint32_t)((w->len + TPB - 1) / TPB)), dim3(TPB), 0, w->stream, a);
                } break;
                case GGRS_SYS_CUSTOM: { rc = launch_custom(w, s, dt_bits, r); if (rc) return rc; } break;
                default: break;
                }
            }
        }
        HIPCHK(w, hipGetLastError());
    }

    return run_spawn_systems(w, inputs, n_inputs, spawn_count, spawn_vx, spawn_vy);
}

// Bounded poll of tags in pinned host memory: true once tags[0], tags[stride], .. (n of them) all equal seq.  k_gen_finalize writes each value and then, with a
// system-scope release, its tag (stride 1); ff_fold_row writes {value, tag} as ONE 16-byte store into a 16-byte cell (stride 2: tags at the odd u64s,
// device_prelude.hpp) -- either way seeing every tag means the values are in host memory.
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
bool spin_for_tags(const volatile uint64_t* tags, uint32_t n, uint64_t seq, int budget_us, uint32_t stride = 1) {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t k = 0;
    for (uint32_t it = 1; ; ++it) {
        while (k < n && tags[(size_t)k * stride] == seq) ++k;
        if (k == n) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
        if ((it & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(budget_us)) return false;
        cpu_relax();
    }
}

// WHEN a fold-forward group's values are on their way: an event recorded right behind the launch (or k_ff_fold) that carries the fold.  A collect whose
// tag poll times out (or is switched off: GGRS_SPIN_WAIT_US=0) waits for THAT event -- not for the whole stream, which may already hold the next ticks
// (with nothing but a stream-wide wait, collect(k) stalled behind batch k + 1).  A small ring: only the newest few groups can be uncollected.
int ff_mark_folded(ggrs_world* w, uint64_t id) {
    ggrs_world::FfEvent& e = w->ff_events[id % ggrs_world::FF_EVENTS];
    if (!e.ev) HIPCHK(w, hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
    HIPCHK(w, hipEventRecord(e.ev, w->stream));
    e.id = id;
    return GGRS_OK;
}
hipError_t ff_wait_event(ggrs_world* w, uint64_t id) {
    ggrs_world::FfEvent& e = w->ff_events[id % ggrs_world::FF_EVENTS];
    if (e.ev && e.id == id) return hipEventSynchronize(e.ev);
    return hipStreamSynchronize(w->stream);                            // (its event was reused by a newer group: that one is behind it on the stream)
}

// Fold-forward: the rows of the last launch are still unfolded and no later launch took them along -- k_ff_fold does it now (a collect with
// nothing enqueued behind the batch, a blocking call, a world being synchronised)
int ff_flush(ggrs_world* w) {
    ggrs_world::FfPending& p = w->ff_pending;
    if (!p.valid) return GGRS_OK;
    FfArgs f; memset(&f, 0, sizeof f);
    f.rows = w->d_ff_rows[p.buf]; f.out = w->d_rows + p.out_off; f.seq = p.seq; f.nvals = p.nvals / p.split; f.g = p.g; f.stride = p.stride; f.istride = p.istride; f.nc1 = w->cks_args.n_cks + 1; f.split = p.split;
    hipLaunchKernelGGL(k_ff_fold, dim3(p.nvals), dim3(TPB), 0, w->stream, f);
    HIPCHK(w, hipGetLastError());
    w->ff_done_id = p.id; p.valid = false;
    return ff_mark_folded(w, p.id);
}

// component_checksum.rs:92-95 (hash the XOR of the entity hashes once more), entity_checksum.rs:29-52, checksum.rs:88-99 (XOR of all
// parts; the upper 64 bits of the u128 are always 0) -- what k_gen_finalize does, over rows the device left in pinned memory: one row per
// workgroup (small groups) or one value per row (fold-forward: the device folded the rows, the tags say when the values are there)
int run_host_folds(ggrs_world* w, uint32_t n) {
    for (; n && !w->folds.empty(); --n) {
        const ggrs_world::HostFold f = w->folds.front(); w->folds.pop_front();
        const uint32_t nc = f.n_cks + 1;
        if (f.ff_id) {
            // the values arrive with the launch (or k_ff_fold) that follows this group's on the stream
            if (f.ff_id > w->ff_done_id) { const int rc = ff_flush(w); if (rc) return rc; }
            const double t0 = w->tl.on ? tl_now_us() : 0;
            const uint32_t nvals = f.n_saves * nc * f.g;              // g = chunks per row here: one {value, tag} cell per chunk
            if (!spin_for_tags(w->h_rows + f.rows_off + 1, nvals, f.ff_seq, std::max(w->knobs.spin_wait_us, 0), 2)) {
                HIPCHK(w, ff_wait_event(w, f.ff_id));
                if (!spin_for_tags(w->h_rows + f.rows_off + 1, nvals, f.ff_seq, 1000000, 2)) return w->fail(GGRS_E_HIP, "fold-forward: the tags of group %llu never arrived", (unsigned long long)f.ff_id);
            }
            if (w->tl.on) w->tl.tag_wait_us += tl_now_us() - t0;
        }
        const double t1 = w->tl.on ? tl_now_us() : 0;
        const uint32_t vs = f.ff_id ? 2u : 1u;                        // fold-forward cells are {value, tag}: the values sit at the even u64s
        for (uint32_t m = 0; m < f.members; ++m)
            for (uint32_t sv = 0; sv < f.n_saves; ++sv) {
                uint64_t total = 0;
                for (uint32_t c = 0; c < nc; ++c) {
                    const uint64_t* row = w->h_rows + f.rows_off + ((uint64_t)(m * f.n_saves + sv) * nc + c) * f.g * vs;
                    if (c == f.n_cks) { uint64_t sum = 0; for (uint32_t t = 0; t < f.g; ++t) sum += row[(size_t)t * vs]; total ^= sea_pair(sum, f.save_len[sv]); }
                    else { uint64_t x = 0; for (uint32_t t = 0; t < f.g; ++t) x ^= row[(size_t)t * vs]; total ^= sea_one(x); }
                }
                uint64_t* out = w->h_results + 2 * (uint64_t)(f.res_slot + m * f.n_saves + sv);
                out[0] = total; out[1] = 0;
            }
        if (w->tl.on) w->tl.fold_us += tl_now_us() - t1;
    }
    // the row buffer is a ring: everything before the oldest unfolded group is free again
    if (w->folds.empty()) { w->rows_used = 0; w->rows_tail = 0; } else w->rows_tail = w->folds.front().rows_off;
    return GGRS_OK;
}
// `need` u64 of the pinned row ring (rows of pending folds live in [tail, head), mod wrap); false: no room (the group is folded by k_gen_finalize)
bool rows_ring_alloc(ggrs_world* w, uint64_t need, uint64_t* off) {
    if (!w->h_rows) return false;
    if (w->folds.empty()) { w->rows_used = 0; w->rows_tail = 0; }
    uint64_t& head = w->rows_used;
    if (head >= w->rows_tail) {
        if (head + need <= w->rows_cap) { *off = head; head += need; return true; }
        if (need < w->rows_tail) { *off = 0; head = need; return true; }          // wrap: the front of the buffer has been folded
        return false;
    }
    if (head + need < w->rows_tail) { *off = head; head += need; return true; }
    return false;
}
// room for the partial rows of a small group in the pinned row buffer?  (no: the group is folded on the device)
// `blocking`: the caller waits for this group's checksums right away (the synchronous API) -- the host's fold is then serial with the
// kernel instead of hidden behind the next tick's, so only small groups take it.
bool host_fold_rows(ggrs_world* w, uint32_t g, uint32_t n_saves, uint32_t n_cks, uint32_t members, uint64_t* off, bool blocking = false) {
    if (!w->h_rows || w->device_results_only || w->dev_spawn || !n_saves) return false;      // (device-decided spawns: RollbackOrdered::len at each Save is only known on the device -- k_gen_finalize)
    if (blocking ? g > HOST_FOLD_MAX_WGS_BLOCKING : g > (uint32_t)w->knobs.fold_forward_min_wgs) return false;
    return rows_ring_alloc(w, (uint64_t)g * n_saves * (n_cks + 1) * members, off);
}

// Spawns decided on the device: RollbackOrdered::len of the live world as the last launch left it (pinned), and what went wrong inside a launch
int len_sync(ggrs_world* w) {
    if (!w->dev_spawn || !w->len_stale || !w->h_sp_len) return GGRS_OK;
    HIPCHK(w, hipStreamSynchronize(w->stream));
    w->len_stale = false;
    const uint64_t err = w->h_sp_len[1];
    const uint64_t l_ = w->h_sp_len[0]; w->len = l_ < w->capacity ? l_ : w->capacity;
    w->live.dirty_len = std::max(w->live.dirty_len, w->len);
    if (err) {
        w->h_sp_len[1] = 0;
        return err == 1 ? w->fail(GGRS_E_CAPACITY, "entities spawned by the schedule's systems (e.spawn) exceed the world's capacity of %llu: that frame's spawns were dropped", (unsigned long long)w->capacity)
                        : w->fail(GGRS_E_HIP, "a grid barrier of a device-spawn launch timed out (the launch was not resident as a whole?)");
    }
    return GGRS_OK;
}
int read_back(ggrs_world* w, uint32_t n_results, uint64_t* out) {
    // When the list's last GPU operation is a k_gen_finalize (arm_spin), its workgroups write a tag behind their results in the same pinned
    // allocation: seeing every tag means every kernel of the list has run (one in-order stream) and the results are in host memory -- the
    // runtime's own wait (a marker packet + its completion signal) costs several us more per call, which is all a blocking caller has to
    // hide behind.  Bounded: past GGRS_SPIN_WAIT_US, with host-folded rows pending, and every 256th call (so that the runtime retires its
    // command records) it is the stream wait.
    bool seen = false;
    if (w->spin_n && w->folds.empty() && (w->spin_seq & 255u) != 0) {
        seen = spin_for_tags(w->h_done, w->spin_n, w->spin_seq, w->knobs.spin_wait_us);
        ++(seen ? w->spin_hits : w->spin_misses);
    }
    w->spin_n = 0;
    // ... and a list whose every group folded ITSELF (self-fold: the launch's fold workgroups write {value, tag} cells once all of its tiles are done) needs no
    // stream wait either: run_host_folds polls those tags.  (Every 256th call it is the stream wait all the same.)
    if (!seen && !w->folds.empty() && !w->ff_pending.valid && w->knobs.spin_wait_us > 0 && (++w->self_fold_calls & 255u) != 0) {
        seen = true;
        for (const auto& f : w->folds) if (!f.ff_id || f.ff_id > w->ff_done_id) { seen = false; break; }
    }
    if (!seen) {
        int rc = ff_flush(w); if (rc) return rc;                     // (nothing is pending in the synchronous API: a no-op there)
        HIPCHK(w, hipStreamSynchronize(w->stream));
    }
    int rc = run_host_folds(w, ~0u); if (rc) return rc;
    rc = len_sync(w); if (rc) return rc;
    stage_ring_reset(w);
    if (n_results && out) memcpy(out, w->h_results, (size_t)n_results * 16);
    return GGRS_OK;
}

void apply_synctest_confirmed(ggrs_world* w) {
    // handle_requests, schedule_systems.rs:204-220: SyncTest => current_frame - check_distance, if >= 0
    if (w->synctest_cd < 0) return;
    const int32_t c = w->frame - w->synctest_cd;
    if (c >= 0) { w->has_confirmed = true; w->confirmed = c; }
}

// does a spawn system fire in this AdvanceFrame?  spawn_particles (particles.rs:254-270): a player holds the system's input bit and the host
// drew spawn_count velocities; a user-written spawner (ggrs_hip_add_spawn_system): the host decided -- spawn_count is what the system spawns
bool advance_spawns(const ggrs_world* w, const ggrs_request& r) {
    if (r.spawn_count == 0 || w->dev_spawn) return false;             // (spawns decided on the device: the request says nothing about them)
    for (auto& s : w->systems) {
        if (s.kind == GGRS_SYS_SPAWN_CUSTOM) return true;
        if (s.kind == GGRS_SYS_PARTICLES_SPAWN && r.inputs && spawn_pressed(w, s, r.inputs, r.n_inputs)) return true;
    }
    return false;
}


// ---- tracing (the reference: tracing spans "HandleRequests" / "SaveWorld" / "LoadWorld" / "AdvanceWorld" and a
// debug! line per request, schedule_systems.rs:171,224-267).  GGRS_HIP_TRACE=1 prints one line per request to stderr;
// GGRS_HIP_ROCTX=1 opens roctx ranges with the same names (the roctx library is dlopen'ed: no link-time dependency), so a
// `rocprofv3 --marker-trace` timeline shows which requests every fused launch carries.
struct Tracer {
    bool log = false;
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Tracer() {
        if (const char* v = getenv("GGRS_HIP_TRACE")) log = atoi(v) != 0;
        if (const char* v = getenv("GGRS_HIP_ROCTX")) if (atoi(v)) {
            void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);    // what rocprofv3 --marker-trace intercepts
            if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (!push || !pop) { push = nullptr; pop = nullptr; }
            }
        }
    }
    bool on() const { return log || push; }
    void begin(const char* name) const { if (push) push(name); }
    void end() const { if (pop) pop(); }
};
const Tracer& tracer() { static Tracer t; return t; }
struct TraceRange {
    bool active;
    explicit TraceRange(const char* name) : active(tracer().push != nullptr) { if (active) tracer().begin(name); }
    ~TraceRange() { if (active) tracer().end(); }
};
void trace_request(const ggrs_world* w, const ggrs_request& r) {
    const Tracer& t = tracer();
    if (!t.on()) return;
    char buf[96];
    switch (r.kind) {
    case GGRS_REQ_SAVE: snprintf(buf, sizeof buf, "SaveWorld: saving snapshot for frame %d", r.frame); break;
    case GGRS_REQ_LOAD: snprintf(buf, sizeof buf, "LoadWorld: restoring snapshot for frame %d", r.frame); break;
    case GGRS_REQ_ADVANCE: snprintf(buf, sizeof buf, "AdvanceWorld: advancing to frame: %d", w->frame + 1); break;
    default: snprintf(buf, sizeof buf, "unknown request %u", r.kind);
    }
    if (t.log) fprintf(stderr, "[ggrs_hip] %s\n", buf);
    if (t.push) { t.begin(buf); t.end(); }            // a zero-length marker inside the enclosing HandleRequests range
}

}  // namespace
