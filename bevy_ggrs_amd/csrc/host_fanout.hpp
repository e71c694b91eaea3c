// host_fanout.hpp -- speculative fan-out ACROSS GPUs over RCCL (include/ggrs_hip.h, "Speculative fan-out ACROSS GPUs").
// Part of the single translation unit ggrs_hip.hip (included last: it uses the C ABI entry points above).
#pragma once

// =============================================================================================
// Speculative fan-out over RCCL (include/ggrs_hip.h, "Speculative fan-out ACROSS GPUs")
// =============================================================================================
}  // extern "C"

namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    std::string why;
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && Broadcast && AllGather && GetErrorString && CommCount && CommUserRank; }
};
// ONE RCCL per process: a copy that is already mapped (a torch process ships its own librccl.so) wins over /opt/rocm's,
// or two collective runtimes would each initialise the device.
void rccl_load(Rccl& r) {
    // GGRS_RCCL_LIB=<path>: load THIS collective library instead (tests: a same-GPU transport double, tests/cpp/rccl_double.cpp,
    // so that the rank != 0 half of the fan-out runs on a one-GPU box where RCCL refuses two ranks per device)
    if (const char* forced = getenv("GGRS_RCCL_LIB")) { if (*forced) r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL); }
    else {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
        for (const char* n : names) if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) { const char* e = dlerror(); r.why = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    if (!r.ok()) r.why = "librccl.so lacks an expected entry point";
}
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_load(r); });                     // worlds may start their fan-out from several threads
    return r;
}
constexpr int FANOUT_MAX_INFLIGHT = 8;
}  // namespace

struct ggrs_fanout {
    ggrs_world* w = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
    hipStream_t comm_stream = nullptr;
    // one slot = one all-gather: the checksums of `interval` consecutive steps
    // What a rank sends per group: [n_steps x n_saves Checksum(u128)][n_steps tags].  A tag = {frame of the step's first request, number of its
    // SaveGameState requests}: collectives pair up by ORDER, so ranks that ran different step counts would silently gather different frames'
    // checksums into one table (bench.py's clock-based pre-heat did exactly that in round 4) -- the tags make collect refuse such a table.
    struct Slot { uint64_t* d_send = nullptr; uint64_t* d_recv = nullptr; uint64_t* h_recv = nullptr; uint64_t* h_tags = nullptr; hipEvent_t ready = nullptr, done = nullptr;
                  uint32_t n_saves = 0, n_steps = 0; bool closed = false; uint32_t first[64]; };   // first[k]: step k's slot in the pinned result ring
    Slot slot[FANOUT_MAX_INFLIGHT];
    uint32_t head = 0, tail = 0;         // tail: slot being filled, head: oldest uncollected
    uint32_t cap_u128 = 4096;            // checksums per rank a slot can hold (steps x saves)
    uint32_t interval = 1;               // steps per all-gather
    // The shape every all-gather of this fan-out has: `interval` steps of `agreed_saves` Checksum(u128)s (+ one tag per step) from EVERY rank -- agreed
    // by ONE small all-gather at the first step after init / set_interval (fanout_agree_shape) and used as the send count of every later one, a closing
    // partial group included (zero-padded): ranks can no longer hand RCCL mismatched counts (DESIGN 9.8 of round 4) -- a rank whose lists have
    // another shape is refused before anything is enqueued, a rank that ran fewer steps shows up in the tags (collect: "ranks are out of step")
    bool shape_agreed = false; uint32_t agreed_saves = 0;
    bool owns_results_flag = false;      // this object switched its world to device-side folds (ggrs_world::device_results_only)
    BranchKeep keep;                     // what the last ggrs_hip_fanout_step_branches retained (ggrs_hip_fanout_adopt's input); invalid after any other step
    uint32_t last_branches = 0;          // branches per rank of that step: global branch index = rank x last_branches + local index
    std::string err;
    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        err = buf;
        return code;
    }
};
#define FANCHK_HIP(f, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (f)->fail(GGRS_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define FANCHK_NCCL(f, call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return (f)->fail(GGRS_E_HIP, "%s failed: %s", #call, rccl().GetErrorString(e_)); } while (0)

namespace {
// {steps per all-gather, SaveGameState requests per step} of every rank, exchanged ONCE (blocking: a start-up cost) through the buffers of the empty
// slot being filled; every rank sees the whole table, so every rank fails the same way when they differ
int fanout_agree_shape(ggrs_fanout* f, uint32_t saves) {
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    // (every rank enters the collective whatever its own numbers are -- a rank that bailed out here would leave the others waiting in it for ever -- and every
    // rank judges the WHOLE table afterwards, so they all fail the same way)
    s.h_tags[0] = f->interval; s.h_tags[1] = saves;
    FANCHK_HIP(f, hipMemcpyAsync(s.d_send, s.h_tags, 16, hipMemcpyHostToDevice, f->comm_stream));
    FANCHK_NCCL(f, rccl().AllGather(s.d_send, s.d_recv, 2, ncclUint64, f->comm, f->comm_stream));
    FANCHK_HIP(f, hipMemcpyAsync(s.h_recv, s.d_recv, (size_t)16 * f->size, hipMemcpyDeviceToHost, f->comm_stream));
    FANCHK_HIP(f, hipStreamSynchronize(f->comm_stream));
    for (int r = 0; r < f->size; ++r)
        if (s.h_recv[2 * r] * s.h_recv[2 * r + 1] > f->cap_u128)
            return f->fail(GGRS_E_INVALID, "%llu steps x %llu checksums per rank in one all-gather on rank %d (at most %u)", (unsigned long long)s.h_recv[2 * r], (unsigned long long)s.h_recv[2 * r + 1], r, f->cap_u128);
    for (int r = 0; r < f->size; ++r)
        if (s.h_recv[2 * r] != f->interval || s.h_recv[2 * r + 1] != saves)
            return f->fail(GGRS_E_INVALID, "ranks disagree on the shape of an all-gather group: %u steps x %u saves on rank %d, %llu steps x %llu saves on rank %d "
                                           "(every rank must use the same interval and pass lists with the same number of SaveGameState requests)",
                           f->interval, saves, f->rank, (unsigned long long)s.h_recv[2 * r], (unsigned long long)s.h_recv[2 * r + 1], r);
    f->shape_agreed = true; f->agreed_saves = saves;
    return GGRS_OK;
}
// One status word per rank, seen by all (blocking; through the buffers of the empty slot): a collective that only SOME ranks can take part in -- the broadcast of
// an adopted block, whose owner alone knows whether it kept that frame -- is entered by all of them or by none.  *bad_rank: the first rank whose status is not 0.
int fanout_agree_status(ggrs_fanout* f, int status, int* bad_rank, int* bad_status) {
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    s.h_tags[0] = (uint64_t)(int64_t)status;
    FANCHK_HIP(f, hipMemcpyAsync(s.d_send, s.h_tags, 8, hipMemcpyHostToDevice, f->comm_stream));
    FANCHK_NCCL(f, rccl().AllGather(s.d_send, s.d_recv, 1, ncclUint64, f->comm, f->comm_stream));
    FANCHK_HIP(f, hipMemcpyAsync(s.h_recv, s.d_recv, (size_t)8 * f->size, hipMemcpyDeviceToHost, f->comm_stream));
    FANCHK_HIP(f, hipStreamSynchronize(f->comm_stream));
    *bad_rank = -1; *bad_status = 0;
    for (int r = 0; r < f->size; ++r) if (s.h_recv[r] != 0) { *bad_rank = r; *bad_status = (int)(int64_t)s.h_recv[r]; break; }
    return GGRS_OK;
}
// closes the slot being filled: ONE all-gather of the agreed size -- interval x agreed_saves checksums + interval tags per rank, zero beyond the steps the
// group really holds --, then device -> pinned, on the side stream
int fanout_close_slot(ggrs_fanout* f) {
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    if (s.n_steps == 0 || s.closed) return GGRS_OK;
    const size_t n = (size_t)s.n_steps * s.n_saves;
    const size_t n_full = (size_t)f->interval * f->agreed_saves;             // checksums per rank of a full group: the fixed layout of every all-gather
    ggrs_world* w = f->w;
    // everything of the group happens here, once per `interval` steps and on the side stream.  What does not depend on the kernels goes first: the steps' tags
    // ride behind the checksums (one small pinned -> device copy per group), a partial group (the closing one) is padded with zeros.  Then the wait for the
    // group's last tick.  The Checksum(u128)s themselves are ALREADY in the send buffer -- k_gen_finalize wrote its second copy there (ggrs_world::dev_results_dst)
    // -- so between the last kernel and the collective there is no copy at all; a world without the generated kernel (its folds write the pinned ring only) still
    // takes the pinned -> device hop (consecutive steps sit in consecutive ring slots unless the ring wrapped).  A step itself adds nothing to the world's stream.
    const size_t per_rank = n_full + f->interval;
    for (uint32_t k = s.n_steps; k < f->interval; ++k) { s.h_tags[2 * k] = 0; s.h_tags[2 * k + 1] = 0; }
    if (n < n_full) FANCHK_HIP(f, hipMemsetAsync(s.d_send + n * 2, 0, (n_full - n) * 16, f->comm_stream));
    FANCHK_HIP(f, hipMemcpyAsync(s.d_send + n_full * 2, s.h_tags, (size_t)f->interval * 16, hipMemcpyHostToDevice, f->comm_stream));
    FANCHK_HIP(f, hipEventRecord(s.ready, w->stream));
    FANCHK_HIP(f, hipStreamWaitEvent(f->comm_stream, s.ready, 0));
    for (uint32_t k = 0; k < s.n_steps && s.n_saves && !w->gen_ok; ) {
        uint32_t run = 1;
        while (k + run < s.n_steps && s.first[k + run] == s.first[k] + run * s.n_saves) ++run;
        FANCHK_HIP(f, hipMemcpyAsync(s.d_send + (size_t)k * s.n_saves * 2, w->h_results + 2 * (size_t)s.first[k], (size_t)run * s.n_saves * 16, hipMemcpyHostToDevice, f->comm_stream));
        k += run;
    }
    FANCHK_NCCL(f, rccl().AllGather(s.d_send, s.d_recv, per_rank * 2, ncclUint64, f->comm, f->comm_stream));
    FANCHK_HIP(f, hipMemcpyAsync(s.h_recv, s.d_recv, per_rank * 16 * f->size, hipMemcpyDeviceToHost, f->comm_stream));
    FANCHK_HIP(f, hipEventRecord(s.done, f->comm_stream));
    s.closed = true;
    ++f->tail;
    return GGRS_OK;
}
}  // namespace

extern "C" {

int ggrs_hip_fanout_unique_id(uint8_t id_out[GGRS_FANOUT_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == GGRS_FANOUT_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return GGRS_E_INVALID;
    if (!rccl().ok()) return GGRS_E_HIP;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return GGRS_E_HIP;
    memcpy(id_out, &id, sizeof id);
    return GGRS_OK;
}
int ggrs_hip_fanout_init(ggrs_world* w, const uint8_t id[GGRS_FANOUT_ID_BYTES], int rank, int world_size, ggrs_fanout** out) {
    if (!w || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return GGRS_E_INVALID;
    DeviceGuard dg(w);
    int rc = seal(w); if (rc) return rc;
    if (!rccl().ok()) return w->fail(GGRS_E_HIP, "%s", rccl().why.c_str());
    // the fan-out's all-gather reads the result ring in stream order: a list enqueued earlier whose rows the HOST is still to fold would
    // hand it slots nobody has written yet
    if (!w->pending.empty() || !w->folds.empty()) return w->fail(GGRS_E_INVALID, "fan-out init while %zu enqueued batches are uncollected", w->pending.size());
    if (w->device_results_only) return w->fail(GGRS_E_INVALID, "this world already drives a fan-out");
    ggrs_fanout* f = new ggrs_fanout();
    f->w = w; f->rank = rank; f->size = world_size;
    ncclUniqueId uid; memcpy(&uid, id, sizeof uid);
    ncclResult_t e = rccl().CommInitRank(&f->comm, world_size, uid, rank);
    if (e != ncclSuccess) { rc = w->fail(GGRS_E_HIP, "ncclCommInitRank failed: %s", rccl().GetErrorString(e)); delete f; return rc; }
    bool ok = hipStreamCreateWithFlags(&f->comm_stream, hipStreamNonBlocking) == hipSuccess;
    for (auto& s : f->slot) {
        ok = ok && hipMalloc((void**)&s.d_send, (size_t)(f->cap_u128 + 64) * 16) == hipSuccess;
        ok = ok && hipMalloc((void**)&s.d_recv, (size_t)(f->cap_u128 + 64) * 16 * world_size) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&s.h_recv, (size_t)(f->cap_u128 + 64) * 16 * world_size) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&s.h_tags, 64 * 16) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { rc = w->fail(GGRS_E_HIP, "fan-out staging buffers could not be allocated"); ggrs_hip_fanout_destroy(f); return rc; }
    // from here on the all-gather reads the Checksum(u128)s from the result ring on the GPU's side of the stream: no host-side folds.
    // Set only once everything above succeeded, and cleared by ggrs_hip_fanout_destroy: a failed or finished fan-out leaves the world
    // with its host-side fold and event-on-kernel path
    w->device_results_only = true; f->owns_results_flag = true; w->fanout_backref = &f->w;
    *out = f;
    return GGRS_OK;
}
void ggrs_hip_fanout_destroy(ggrs_fanout* f) {
    if (!f) return;
    if (f->w) (void)hipSetDevice(f->w->device);
    if (f->comm_stream) (void)hipStreamSynchronize(f->comm_stream);
    for (auto& s : f->slot) {
        if (s.d_send) (void)hipFree(s.d_send);
        if (s.d_recv) (void)hipFree(s.d_recv);
        if (s.h_recv) (void)hipHostFree(s.h_recv);
        if (s.h_tags) (void)hipHostFree(s.h_tags);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (f->comm && rccl().ok()) (void)rccl().CommDestroy(f->comm);
    if (f->comm_stream) (void)hipStreamDestroy(f->comm_stream);
    if (f->w) f->w->fanout_backref = nullptr;
    if (f->owns_results_flag && f->w) {
        // device-side folds may still be queued on the world's stream; they write the result ring whoever reads it.  What must not be
        // left behind is a batch whose checksums nobody collected through this object
        if (f->w->stream) (void)hipStreamSynchronize(f->w->stream);
        f->w->device_results_only = false;
    }
    delete f;
}
const char* ggrs_hip_fanout_last_error(ggrs_fanout* f) { return f ? f->err.c_str() : "null fan-out"; }
int ggrs_hip_fanout_comm_info(ggrs_fanout* f, int* rank_out, int* size_out, int* device_out) {
    if (!f || !f->comm || !f->w) return GGRS_E_INVALID;
    int n = 0, r = 0;
    FANCHK_NCCL(f, rccl().CommCount(f->comm, &n));
    FANCHK_NCCL(f, rccl().CommUserRank(f->comm, &r));
    if (rank_out) *rank_out = r;
    if (size_out) *size_out = n;
    if (device_out) *device_out = f->w->device;
    return GGRS_OK;
}

int ggrs_hip_fanout_set_interval(ggrs_fanout* f, uint32_t steps_per_all_gather) {
    // a group's steps are all outstanding batches of the world until the group is collected: at most 16 (ggrs_hip_enqueue_requests)
    if (!f || steps_per_all_gather == 0 || steps_per_all_gather > 16) return GGRS_E_INVALID;
    if (f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps) return f->fail(GGRS_E_INVALID, "interval changed inside a partly filled group");
    f->interval = steps_per_all_gather;
    f->shape_agreed = false;                                     // the next step exchanges {interval, saves per step} again
    return GGRS_OK;
}

int ggrs_hip_fanout_sync_confirmed(ggrs_fanout* f, int root) {
    if (!f || !f->w || root < 0 || root >= f->size) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->head != f->tail || f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps || !w->pending.empty())
        return f->fail(GGRS_E_INVALID, "sync_confirmed while steps are in flight: collect them first");
    void* live = nullptr;
    int rc = ggrs_hip_live_state_ptr(w, &live);          // refreshes the block's header (len, frame) on every rank
    if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    // xGMI is point-to-point: one flat broadcast of the packed block, in place in HBM, on the world's own stream
    FANCHK_NCCL(f, rccl().Broadcast(live, live, (size_t)w->state_bytes, ncclUint8, root, f->comm, w->stream));
    FANCHK_HIP(f, hipStreamSynchronize(w->stream));
    rc = ggrs_hip_adopt_live_state(w);
    if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    return GGRS_OK;
}

// A step this rank REFUSES after the fan-out's shape was agreed still takes part in the group's all-gather: its place carries the tag {~0, error code} and the group is
// closed at once (a partial group is padded to the agreed size), so that the other ranks -- whose steps went through -- find "rank r refused step k" in their collect
// instead of waiting in the collective for a rank that has stopped calling.  Returns the error the step is refused with.
static int fanout_refuse_step(ggrs_fanout* f, int code) {
    if (!f->shape_agreed || f->size == 1) return code;
    const std::string why = f->err;
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    s.n_saves = f->agreed_saves;
    s.first[s.n_steps] = 0;
    s.h_tags[2 * s.n_steps] = ~0ull; s.h_tags[2 * s.n_steps + 1] = (uint64_t)(int64_t)code;
    ++s.n_steps;
    (void)fanout_close_slot(f);
    f->err = why;
    return code;
}
static int fanout_step_impl(ggrs_fanout* f, const ggrs_request* reqs, uint32_t n, const ggrs_branch_step* bs, uint32_t* n_saves_out) {
    if (!f || !f->w || (!reqs && n)) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->tail - f->head >= (uint32_t)FANOUT_MAX_INFLIGHT) return f->fail(GGRS_E_INVALID, "%d all-gathers in flight: call ggrs_hip_fanout_collect", FANOUT_MAX_INFLIGHT);
    ggrs_fanout::Slot& s = f->slot[f->tail % FANOUT_MAX_INFLIGHT];
    if (s.n_steps == 0) { s.closed = false; s.n_saves = 0; }
    // everything that can refuse the step is checked BEFORE the world advances: a batch enqueued here and not tracked by a slot
    // would shift every later collect by one
    uint32_t want = 0;
    for (uint32_t i = 0; i < n; ++i) want += reqs[i].kind == GGRS_REQ_SAVE;
    if (bs) {
        if (bs->n_branches == 0 || bs->n_branches > BRANCH_MAX || bs->n_frames == 0 || bs->n_frames > (uint32_t)MAX_TICK_STEPS) return f->fail(GGRS_E_INVALID, "a branch step holds 1..%u branches of 1..%d frames", BRANCH_MAX, MAX_TICK_STEPS);
        want += bs->n_branches * ((bs->flags & GGRS_BRANCH_SAVE_LAST) ? bs->n_frames : bs->n_frames - 1);
    }
    f->keep.valid = false;                                              // whatever an earlier step retained is about to be overwritten (or is history)
    if (!f->shape_agreed) { const int arc = fanout_agree_shape(f, want); if (arc) return arc; }
    if (want != f->agreed_saves) return fanout_refuse_step(f, f->fail(GGRS_E_INVALID, "every step of this fan-out holds %u SaveGameState requests (agreed by all ranks at the first step); this list has %u -- "
                                                              "ggrs_hip_fanout_set_interval starts a new agreement", f->agreed_saves, want));
    uint32_t ns = 0;
    // the device copy of this step's Checksum(u128)s lands where the all-gather sends from
    if (w->gen_ok) w->dev_results_dst = s.d_send + 2 * (size_t)s.n_steps * f->agreed_saves;
    int rc = enqueue_impl(w, reqs, n, bs, bs ? &f->keep : nullptr, &ns);
    w->dev_results_dst = nullptr;
    if (rc) return fanout_refuse_step(f, f->fail(rc, "%s", ggrs_hip_last_error(w)));
    if (bs) f->last_branches = bs->n_branches;
    s.n_saves = ns;
    s.first[s.n_steps] = w->pending.back().first;        // where the kernels write this step's checksums (pinned result ring)
    s.h_tags[2 * s.n_steps] = n ? (uint64_t)(uint32_t)reqs[0].frame | ((uint64_t)reqs[0].kind << 32) : 0;   // (slot not closed: the side stream does not read h_tags yet)
    s.h_tags[2 * s.n_steps + 1] = ns;
    ++s.n_steps;
    if (n_saves_out) *n_saves_out = ns;
    if (s.n_steps >= f->interval) return fanout_close_slot(f);
    return GGRS_OK;
}
int ggrs_hip_fanout_step(ggrs_fanout* f, const ggrs_request* reqs, uint32_t n, uint32_t* n_saves_out) { return fanout_step_impl(f, reqs, n, nullptr, n_saves_out); }
int ggrs_hip_fanout_step_branches(ggrs_fanout* f, const ggrs_branch_step* step, uint32_t* n_saves_out) {
    if (!f || !step || (!step->prefix && step->n_prefix)) return GGRS_E_INVALID;
    return fanout_step_impl(f, step->prefix, step->n_prefix, step, n_saves_out);
}

// The true inputs matched what `branch` predicted up to `frame`: its retained state becomes the world (include/ggrs_hip.h).  Collective.
int ggrs_hip_fanout_adopt(ggrs_fanout* f, uint32_t branch, int32_t frame, uint32_t mode, const ggrs_request* replay, uint32_t n_replay, uint64_t* checksums_out, uint32_t* n_checksums_out) {
    if (!f || !f->w || (mode != GGRS_ADOPT_RECOMPUTE && mode != GGRS_ADOPT_BROADCAST) || (!replay && n_replay)) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (n_checksums_out) *n_checksums_out = 0;
    if (f->head != f->tail || f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps || !w->pending.empty())
        return f->fail(GGRS_E_INVALID, "adopt while steps are in flight: collect them first");
    if (!f->last_branches) return f->fail(GGRS_E_INVALID, "adopt without a preceding ggrs_hip_fanout_step_branches");
    if (branch >= f->last_branches * (uint32_t)f->size) return f->fail(GGRS_E_INVALID, "branch %u of %u x %d", branch, f->last_branches, f->size);
    const int owner = (int)(branch / f->last_branches);
    const uint32_t local = branch % f->last_branches;
    const bool mine = owner == f->rank;
    const int32_t F = w->frame;                                        // where the last step's prefix left every rank
    // ---- the owner: its retained block trades places with the ring slot a SaveGameState(frame) would have filled
    int spec_idx = -1, pre = GGRS_OK;
    if (mine) {
        const BranchKeep& k = f->keep;
        const int64_t o = (int64_t)frame - (int64_t)k.base_frame - 1;
        if (!k.valid || k.base_frame != F || o < 0 || o >= (int64_t)k.n_out || local >= k.n_branches || k.blk[(size_t)local * k.n_out + (size_t)o] < 0)
            pre = f->fail(GGRS_E_NO_SNAPSHOT, "branch %u holds no retained state of frame %d (the last branch step started at frame %d and %s)", branch, frame, k.base_frame,
                          k.valid ? "kept other frames: GGRS_BRANCH_RETAIN_ALL keeps every one" : "kept none: GGRS_BRANCH_RETAIN_*");
        else spec_idx = k.blk[(size_t)local * k.n_out + (size_t)o];
    }
    if (mode == GGRS_ADOPT_BROADCAST && f->size > 1) {
        // only the owner knows whether it kept that frame; the broadcast below is entered by every rank or by none (a rank that bailed out alone would leave the
        // others waiting in the collective for ever)
        const std::string mine_err = f->err;
        int bad_rank = -1, bad_status = 0;
        const int arc = fanout_agree_status(f, pre, &bad_rank, &bad_status); if (arc) return arc;
        if (pre) { f->err = mine_err; return pre; }
        if (bad_rank >= 0) return f->fail(bad_status, "rank %d cannot hand over branch %u at frame %d (its error %d): no rank adopted anything", bad_rank, branch, frame, bad_status);
    } else if (pre) return pre;
    if (mine || mode == GGRS_ADOPT_BROADCAST) {
        // RollbackFrameCount = frame; discard_old_snapshots + GgrsSnapshots::push(frame) (mod.rs:147-202) over slot indices -- the slot's block is then the branch's
        w->frame = frame; w->has_confirmed = true; w->confirmed = frame;
        ring_confirm(w, w->confirmed);
        int sl = -1;
        int rc = ring_push(w, frame, &sl); if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
        if (sl < 0) return f->fail(GGRS_E_INVALID, "adopt needs a ring depth of at least 1");
        Block& slot = w->slots[sl];
        if (mine) std::swap(slot, w->spec_blocks[spec_idx]);
        if (mode == GGRS_ADOPT_BROADCAST && f->size > 1) {
            // the block describes itself: {len, frame, .., extent of its mask bits} in its header
            Header h; memset(&h, 0, sizeof h);
            if (mine) { h.len = slot.len; h.frame = frame; h.active = slot.dirty_len; FANCHK_HIP(f, hipMemcpyAsync(slot.ptr, &h, sizeof h, hipMemcpyHostToDevice, w->stream)); }
            FANCHK_NCCL(f, rccl().Broadcast(slot.ptr, slot.ptr, (size_t)w->state_bytes, ncclUint8, owner, f->comm, w->stream));
            if (!mine) {
                FANCHK_HIP(f, hipMemcpyAsync(&h, slot.ptr, sizeof h, hipMemcpyDeviceToHost, w->stream));
                FANCHK_HIP(f, hipStreamSynchronize(w->stream));
                if (h.len > w->capacity || h.frame != frame) return f->fail(GGRS_E_INVALID, "the broadcast block says frame %d, len %llu (expected frame %d, capacity %llu)", h.frame, (unsigned long long)h.len, frame, (unsigned long long)w->capacity);
                slot.len = h.len; slot.dirty_len = std::min<uint64_t>(std::max<uint64_t>(h.active, h.len), w->cap_pad);
                for (uint32_t c = 0; c < slot.ver.size(); ++c) slot.ver[c] = ++w->ver_counter;      // bytes from another rank: every column is new
            }
        }
        // LoadWorld from that slot (schedule_systems.rs:238-250): the live world IS the adopted state
        ggrs_request ld; memset(&ld, 0, sizeof ld); ld.kind = GGRS_REQ_LOAD; ld.frame = frame;
        rc = ggrs_hip_enqueue_requests(w, &ld, 1, nullptr);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
        uint32_t got = 0;
        rc = ggrs_hip_collect_checksums(w, nullptr, 0, &got);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    } else {
        // the other ranks re-simulate F -> frame with the confirmed inputs: no bytes cross xGMI, and the checksums are a desync check for free
        uint32_t ns = 0;
        for (uint32_t i = 0; i < n_replay; ++i) ns += replay[i].kind == GGRS_REQ_SAVE;
        if (ns && !checksums_out) return f->fail(GGRS_E_INVALID, "replay holds %u SaveGameState requests but checksums_out is NULL", ns);
        int rc = ggrs_hip_enqueue_requests(w, replay, n_replay, nullptr);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
        uint32_t got = 0;
        rc = ggrs_hip_collect_checksums(w, checksums_out, ns, &got);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
        if (n_checksums_out) *n_checksums_out = got;
        if (w->frame != frame || !ggrs_hip_has_snapshot(w, frame))
            return f->fail(GGRS_E_INVALID, "replay left the world at frame %d %s a snapshot of frame %d: it must re-simulate %d -> %d and end with SaveGameState(%d)", w->frame,
                           ggrs_hip_has_snapshot(w, frame) ? "with" : "without", frame, F, frame, frame);
        w->has_confirmed = true; w->confirmed = frame;
    }
    f->keep.valid = false;                                              // the speculation is history now
    return GGRS_OK;
}

int ggrs_hip_fanout_collect(ggrs_fanout* f, uint64_t* checksums_out, uint32_t max_u128_per_rank, uint32_t* n_steps_out, uint32_t* n_saves_out) {
    if (!f || !f->w) return GGRS_E_INVALID;
    ggrs_world* w = f->w;
    DeviceGuard dg(w);
    if (f->head == f->tail) {                                   // only a partly filled group is left: close it now
        if (f->slot[f->tail % FANOUT_MAX_INFLIGHT].n_steps == 0) return f->fail(GGRS_E_INVALID, "no step in flight");
        int rc = fanout_close_slot(f); if (rc) return rc;
    }
    ggrs_fanout::Slot& s = f->slot[f->head % FANOUT_MAX_INFLIGHT];
    const uint32_t per_rank = s.n_steps * s.n_saves;
    if (per_rank > max_u128_per_rank || (per_rank && !checksums_out)) return f->fail(GGRS_E_INVALID, "oldest group holds %u checksums per rank, room for %u", per_rank, max_u128_per_rank);
    FANCHK_HIP(f, hipEventSynchronize(s.done));
    // keep the world's own batch queue in step (its checksums are this rank's rows of the gathered table)
    std::vector<uint64_t> own(2 * (size_t)s.n_saves + 2);
    for (uint32_t k = 0; k < s.n_steps; ++k) {
        uint32_t got = 0;
        int rc = ggrs_hip_collect_checksums(w, own.data(), s.n_saves, &got);
        if (rc) return f->fail(rc, "%s", ggrs_hip_last_error(w));
    }
    // every rank must have gathered the SAME steps: compare the tags of all `interval` places (a rank that ran fewer steps left zeros), then hand out the
    // checksums without them
    const size_t n_full = (size_t)f->interval * f->agreed_saves;
    const size_t stride = (n_full + f->interval) * 2;                            // u64 per rank in h_recv: the agreed layout
    const uint64_t* mine = s.h_recv + (size_t)f->rank * stride + n_full * 2;
    int out_of_step = -1; uint32_t bad_step = 0;
    for (int r = 0; r < f->size && out_of_step < 0; ++r) {
        const uint64_t* theirs = s.h_recv + (size_t)r * stride + n_full * 2;
        for (uint32_t k = 0; k < f->interval; ++k) if (theirs[2 * k] != mine[2 * k] || theirs[2 * k + 1] != mine[2 * k + 1]) { out_of_step = r; bad_step = k; break; }
    }
    for (int r = 0; r < f->size; ++r) {                                            // a rank that refused a step of this group says so (fanout_refuse_step)
        const uint64_t* theirs = s.h_recv + (size_t)r * stride + n_full * 2;
        for (uint32_t k = 0; k < f->interval; ++k) if (theirs[2 * k] == ~0ull) {
            const int rc = f->fail(GGRS_E_INVALID, "rank %d refused step %u of this all-gather (its error %d): the group's table is incomplete on every rank", r, k, (int)(int64_t)theirs[2 * k + 1]);
            s.n_steps = 0; s.closed = false; ++f->head;
            return rc;
        }
    }
    if (out_of_step >= 0) {
        const uint64_t* theirs = s.h_recv + (size_t)out_of_step * stride + n_full * 2;
        const int rc = f->fail(GGRS_E_INVALID, "ranks are out of step: step %u of this all-gather starts at frame %d with %llu saves on rank %d, at frame %d with %llu saves on rank %d "
                               "(every rank must call ggrs_hip_fanout_step the same number of times with lists of the same shape)", bad_step,
                               (int32_t)(uint32_t)mine[2 * bad_step], (unsigned long long)mine[2 * bad_step + 1], f->rank,
                               (int32_t)(uint32_t)theirs[2 * bad_step], (unsigned long long)theirs[2 * bad_step + 1], out_of_step);
        s.n_steps = 0; s.closed = false; ++f->head;
        return rc;
    }
    for (int r = 0; r < f->size && per_rank; ++r) memcpy(checksums_out + (size_t)r * per_rank * 2, s.h_recv + (size_t)r * stride, (size_t)per_rank * 16);
    if (n_steps_out) *n_steps_out = s.n_steps;
    if (n_saves_out) *n_saves_out = s.n_saves;
    s.n_steps = 0; s.closed = false;
    ++f->head;
    return GGRS_OK;
}

}  // extern "C"

