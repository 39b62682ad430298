// gsx_comm.cpp — the collectives of the multi-GPU path inside libgsx, over RCCL (xGMI): one communicator per viewer, every
// call enqueued on the viewer's stream.  No reference counterpart: the reference renders on one wgpu device
// (src/main.rs:85-98).  RCCL is loaded at run time (dlopen), so that a single-GPU host needs nothing but the HIP runtime and
// `ldd libgsx.so` stays what it was; a missing or failing RCCL is GSX_ERR_RCCL, never an abort.
//
// The path has three collectives:
//   all-to-all of fixed-size record slots   grouped ncclSend / ncclRecv: point-to-point, all 7 xGMI links of a GPU busy at once
//   all-gather of the per-tile saturation keys (32 KB per frame at 1080p)
//   all-gather of the framebuffer bands, in place (the bands are disjoint: "reduce of per-GPU tile fragments" with no arithmetic)
// and gsx_shard_render_frame strings the stage calls of gsx_api_shard.cpp and these together: one call per frame, no host
// round trip inside it.
#include <dlfcn.h>

#include <chrono>

#include <mutex>

#include "gsx_state.h"

namespace {

typedef int ncclResult_t;            // ncclSuccess = 0
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ { char internal[128]; };
enum { kNcclChar = 0 };              // ncclInt8 / ncclChar

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // a copy the process already holds (PyTorch ships one) is reused; otherwise the ROCm installation's
    // GSX_RCCL_LIBRARY (read once, at the first collective call of the process): this file instead of the search list
    const char* forced = getenv("GSX_RCCL_LIBRARY");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::vector<const char*> names;
    if (forced && *forced) names.push_back(forced);
    else names.assign(std::begin(defaults), std::end(defaults));
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    std::string why;
    for (const char* n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* e = dlerror();  // (one call: dlerror clears what it returns)
            why = e ? e : "?";
        }
    }
    if (!h) {
        g_rccl.error = std::string("librccl not found: ") + (why.empty() ? "?" : why);
        return;
    }
    g_rccl.handle = h;
#define GSX_SYM(field, name)                                                             \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));             \
    if (!g_rccl.field && g_rccl.error.empty()) g_rccl.error = std::string("librccl lacks ") + name
    GSX_SYM(GetUniqueId, "ncclGetUniqueId");
    GSX_SYM(CommInitRank, "ncclCommInitRank");
    GSX_SYM(CommDestroy, "ncclCommDestroy");
    GSX_SYM(GetErrorString, "ncclGetErrorString");
    GSX_SYM(AllGather, "ncclAllGather");
    GSX_SYM(Send, "ncclSend");
    GSX_SYM(Recv, "ncclRecv");
    GSX_SYM(GroupStart, "ncclGroupStart");
    GSX_SYM(GroupEnd, "ncclGroupEnd");
#undef GSX_SYM
}

gsx_status rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) return fail(GSX_ERR_RCCL, "%s", g_rccl.error.c_str());
    return GSX_OK;
}

#define RCCLCHK(expr)                                                                                            \
    do {                                                                                                         \
        ncclResult_t _r = (expr);                                                                                \
        if (_r != 0) return fail(GSX_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

inline gsx_viewer* owner_of(gsx_viewer* v) { return v->parent ? v->parent : v; }
inline ncclComm_t comm_of(gsx_viewer* v) { return static_cast<ncclComm_t>(owner_of(v)->comm); }

// Which stream does a collective of viewer v go to?  Its own — unless the owner runs sharded frames in flight: then every
// collective of every lane is enqueued on the owner's comm stream, in program order (one communicator, one order, the same
// on every rank: nothing two streams could interleave differently on two ranks), and the lane's stream hands over to it
// and takes over from it through a pair of events.
gsx_status route_begin(gsx_viewer* v, hipStream_t* out) {
    gsx_viewer* o = owner_of(v);
    *out = v->stream;
    if (!o->comm_stream) return GSX_OK;
    if (!v->comm_ev_in) {
        HIPCHK(hipEventCreateWithFlags(&v->comm_ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&v->comm_ev_out, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(v->comm_ev_in, v->stream));
    HIPCHK(hipStreamWaitEvent(o->comm_stream, v->comm_ev_in, 0));
    *out = o->comm_stream;
    return GSX_OK;
}
gsx_status route_end(gsx_viewer* v) {
    gsx_viewer* o = owner_of(v);
    if (!o->comm_stream) return GSX_OK;
    HIPCHK(hipEventRecord(v->comm_ev_out, o->comm_stream));
    HIPCHK(hipStreamWaitEvent(v->stream, v->comm_ev_out, 0));
    return GSX_OK;
}

}  // namespace

extern "C" {

gsx_status gsx_comm_unique_id(uint8_t out_id[128]) {
    if (!out_id) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_unique_id: null argument");
    gsx_status st = rccl_ready();
    if (st) return st;
    ncclUniqueId_ id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out_id, id.internal, 128);
    return GSX_OK;
}

gsx_status gsx_viewer_comm_init(gsx_viewer* v, uint32_t world, uint32_t rank, const uint8_t id[128]) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!id || world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init: bad world/rank %u/%u", world, rank);
    if (has_comm(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init: this viewer already has a communicator");
    if ((st = rccl_ready())) return st;
    ncclUniqueId_ uid;
    memcpy(uid.internal, id, 128);
    ncclComm_t c = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&c, (int)world, uid, (int)rank));
    v->comm = c;
    v->comm_world = world;
    v->comm_rank = rank;
    v->comm_self_via_rccl = getenv("GSX_COMM_SELF_VIA_RCCL") != nullptr;
    return GSX_OK;
}

gsx_status gsx_viewer_comm_init_custom(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_comm_all_to_all_fn all_to_all,
                                       gsx_comm_all_gather_fn all_gather, void* ctx) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!all_to_all || !all_gather || world == 0 || world > 64 || rank >= world)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_custom: null function or bad world/rank %u/%u", world, rank);
    if (has_comm(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_custom: this viewer already has a communicator");
    v->comm_a2a_fn = all_to_all;
    v->comm_ag_fn = all_gather;
    v->comm_ctx = ctx;
    v->comm_world = world;
    v->comm_rank = rank;
    return GSX_OK;
}

gsx_status gsx_viewer_comm_destroy(gsx_viewer* v) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_destroy: viewer is null");
    if (!has_comm(v)) return GSX_OK;
    (void)hipSetDevice(v->device);
    v->shard_pending.clear();  // frames in flight die with the communicator
    for (gsx_viewer* l : v->lanes) (void)hipStreamSynchronize(l->stream);
    if (v->comm_stream) (void)hipStreamSynchronize(v->comm_stream);
    (void)hipStreamSynchronize(v->stream);
    v->comm_world = 0;
    if (v->comm_a2a_fn) {
        if (v->comm_group) group_leave(v);
        v->comm_a2a_fn = nullptr;
        v->comm_ag_fn = nullptr;
        v->comm_ctx = nullptr;
        return GSX_OK;
    }
    ncclComm_t c = comm_of(v);
    v->comm = nullptr;
    RCCLCHK(g_rccl.CommDestroy(c));
    return GSX_OK;
}

gsx_status gsx_comm_all_to_all(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_peer) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* o = owner_of(v);
    if (!has_comm(o)) return fail(GSX_ERR_RCCL, "gsx_comm_all_to_all: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_all_to_all: null buffer");
    if (bytes_per_peer == 0) return GSX_OK;
    o->shard_stats.wire_bytes += (uint64_t)(o->comm_world - 1u) * bytes_per_peer;
    if (o->comm_a2a_fn) {  // the caller's transport (or the in-process group): the whole exchange, own slot included
        hipStream_t ts;
        if ((st = route_begin(v, &ts))) return st;
        g_err.clear();
        if ((st = o->comm_a2a_fn(o->comm_ctx, d_send, d_recv, bytes_per_peer, ts)))
            return g_err.empty() ? fail(st, "gsx_comm_all_to_all: the custom transport failed with status %d", (int)st) : st;
        return route_end(v);
    }
    // what this rank keeps for itself does not travel: a device copy on the viewer's own stream, beside the exchange
    // (GSX_COMM_SELF_VIA_RCCL, read by gsx_viewer_comm_init: send it to oneself through RCCL like everything else — the
    // one-rank tests on a one-GPU box exercise ncclSend / ncclRecv that way)
    const bool bypass = !o->comm_self_via_rccl;
    if (bypass) {
        const size_t self_off = (size_t)o->comm_rank * bytes_per_peer;
        HIPCHK(hipMemcpyAsync(static_cast<char*>(d_recv) + self_off, static_cast<const char*>(d_send) + self_off, bytes_per_peer,
                              hipMemcpyDeviceToDevice, v->stream));
        if (o->comm_world == 1) return GSX_OK;
    }
    hipStream_t cs;
    if ((st = route_begin(v, &cs))) return st;
    RCCLCHK(g_rccl.GroupStart());
    for (uint32_t p = 0; p < o->comm_world; ++p) {
        if (bypass && p == o->comm_rank) continue;
        RCCLCHK(g_rccl.Send(static_cast<const char*>(d_send) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm_of(v), cs));
        RCCLCHK(g_rccl.Recv(static_cast<char*>(d_recv) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm_of(v), cs));
    }
    RCCLCHK(g_rccl.GroupEnd());
    return route_end(v);
}

gsx_status gsx_comm_all_gather(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_rank) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* o = owner_of(v);
    if (!has_comm(o)) return fail(GSX_ERR_RCCL, "gsx_comm_all_gather: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_all_gather: null buffer");
    if (bytes_per_rank == 0) return GSX_OK;
    o->shard_stats.wire_bytes += (uint64_t)(o->comm_world - 1u) * bytes_per_rank;
    hipStream_t cs;
    if ((st = route_begin(v, &cs))) return st;
    if (o->comm_ag_fn) {
        g_err.clear();
        if ((st = o->comm_ag_fn(o->comm_ctx, d_send, d_recv, bytes_per_rank, cs)))
            return g_err.empty() ? fail(st, "gsx_comm_all_gather: the custom transport failed with status %d", (int)st) : st;
        return route_end(v);
    }
    RCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, kNcclChar, comm_of(v), cs));
    return route_end(v);
}

}  // extern "C"

namespace {

struct ShardFrame {
    gsx_viewer* l;       // the lane (a viewer of its own, or the owner itself)
    const char* key;
    uint32_t world, rank;
    gsx_shard_layout_t lay;
    uint32_t sat_words;
};

// the padded framebuffer the bands are gathered into (owned by the library, per lane) and the saturation-map buffers
gsx_status frame_buffers(ShardFrame& f) {
    gsx_viewer* v = f.l;
    gsx_status st = gsx_shard_layout(v, f.world, f.rank, &f.lay);
    if (st) return st;
    if (v->ext_fb != v->shard_fb.p || v->shard_fb.bytes < f.lay.padded_framebuffer_bytes) {
        HIPCHK(hipStreamSynchronize(v->stream));
        if (v->shard_fb.bytes < f.lay.padded_framebuffer_bytes) {
            HIPCHK(v->shard_fb.ensure(f.lay.padded_framebuffer_bytes));
            HIPCHK(hipMemsetAsync(v->shard_fb.p, 0, v->shard_fb.bytes, v->stream));
        }
        v->ext_fb = v->shard_fb.p;
        v->ext_fb_bytes = v->shard_fb.bytes;
    }
    if ((st = gsx_shard_feedback_words(v, f.world, &f.sat_words))) return st;
    HIPCHK(v->shard_sat_band.ensure(4 * (size_t)f.sat_words + 16));
    HIPCHK(v->shard_sat_all.ensure((4 * (size_t)f.sat_words + 16) * f.world));
    return GSX_OK;
}

gsx_status exchange_round(ShardFrame& f, uint32_t round, uint32_t T) {
    gsx_viewer* v = f.l;
    const uint64_t per_peer = (uint64_t)(T + 1u) * GSX_RECORD_BYTES;
    {
        gsx_shard_stats& ss = (v->parent ? v->parent : v)->shard_stats;
        ss.exchange_rounds += 1;
        (round == 0 ? ss.last_slot_records : ss.last_repair_slot_records) = T;
    }
    HIPCHK(v->shard_send.ensure(per_peer * f.world));
    HIPCHK(v->shard_recv.ensure(per_peer * f.world));
    gsx_status s2;
    if ((s2 = gsx_shard_pack_slots(v, f.key, f.world, round, v->shard_send.p, T))) return s2;
    if ((s2 = gsx_comm_all_to_all(v, v->shard_send.p, v->shard_recv.p, per_peer))) return s2;
    if ((s2 = gsx_shard_import_slots(v, f.key, v->shard_recv.p, f.world, f.rank, round, T))) return s2;
    if ((s2 = gsx_shard_feedback(v, f.key, f.world, f.rank, v->shard_sat_band.p))) return s2;
    return gsx_comm_all_gather(v, v->shard_sat_band.p, v->shard_sat_all.p, 4 * (uint64_t)f.sat_words);
}

// next frame's limits + the bands, in place: every rank's band lands where it belongs
gsx_status finish_round(ShardFrame& f, float margin, uint32_t radius) {
    gsx_viewer* v = f.l;
    gsx_status s2 = gsx_shard_next_windows(v, f.key, f.world, v->shard_sat_all.p, margin, radius);
    if (s2) return s2;
    char* fb = static_cast<char*>(v->ext_fb);
    return gsx_comm_all_gather(v, fb + f.lay.band_offset_bytes, fb, f.lay.band_bytes);
}

// everything of a frame up to (not including) the look at its verdict: projection, round 0, verification, and — because
// that is what follows in the usual frame — the next limits and the band gather
gsx_status frame_enqueue(ShardFrame& f, ShardPending& p) {
    gsx_status st = frame_buffers(f);
    if (st) return st;
    if ((st = gsx_shard_frame_begin(f.l, f.key, f.world, f.rank, p.speculate, nullptr))) return st;
    if ((st = gsx_shard_slot_records(f.l, f.key, f.world, p.shard_records_max, &p.slot_records))) return st;
    if ((st = exchange_round(f, 0, p.slot_records))) return st;
    if ((st = gsx_shard_verify(f.l, f.key, f.world, f.l->shard_sat_all.p, &p.seq))) return st;
    return finish_round(f, p.margin, p.radius);
}

// the verdict, and what it asks for: round 0 again with whole-shard slots (a slot overflowed), the repair round
gsx_status frame_complete(gsx_viewer* owner, ShardPending& p) {
    ShardFrame f{p.lane, p.key.c_str(), owner->comm_world, owner->comm_rank, {}, 0};
    gsx_status st = frame_buffers(f);
    if (st) return st;
    gsx_shard_verdict verdict{};
    gsx_shard_stats& ss = owner->shard_stats;
    auto timed_wait = [&](gsx_viewer* wv, const char* wkey, uint32_t seq, gsx_shard_verdict* out) {
        const auto t0 = std::chrono::steady_clock::now();
        const gsx_status ws = gsx_shard_wait_verdict(wv, wkey, seq, out);
        ss.verdict_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        return ws;
    };
    for (int attempt = 0;; ++attempt) {
        if ((st = timed_wait(f.l, f.key, p.seq, &verdict))) return st;
        if (!verdict.overflow) break;
        if (attempt == 0) ss.redo_frames += 1;
        if (attempt == 1)
            return fail(GSX_ERR_OOM, "gsx_shard_render_frame: an exchange slot of %u records (a whole shard) overflowed: shard_records_max is wrong",
                        p.slot_records);
        p.slot_records = std::max<uint32_t>(p.shard_records_max, 1u);  // a destination can be sent at most a whole shard: this always fits
        if ((st = exchange_round(f, 0, p.slot_records))) return st;
        if ((st = gsx_shard_verify(f.l, f.key, f.world, f.l->shard_sat_all.p, &p.seq))) return st;
        if ((st = finish_round(f, p.margin, p.radius))) return st;
    }
    ss.frames += 1;
    if (verdict.need_tiles) {
        ss.repair_frames += 1;
        gsx_viewer* v = f.l;
        HIPCHK(v->shard_counts.ensure(16 * (size_t)(f.world + 1)));
        char* cnt = static_cast<char*>(v->shard_counts.p);
        if ((st = gsx_shard_repair_count(v, f.key, f.world, cnt + 16 * (size_t)f.world))) return st;
        if ((st = gsx_comm_all_gather(v, cnt + 16 * (size_t)f.world, cnt, 16))) return st;
        uint32_t seq = 0;
        if ((st = gsx_shard_post_counts(v, f.world, cnt, &seq))) return st;
        gsx_shard_verdict sized{};
        if ((st = timed_wait(v, nullptr, seq, &sized))) return st;
        if ((st = exchange_round(f, 1, std::max<uint32_t>(sized.max_records, 1u)))) return st;
        if ((st = finish_round(f, p.margin, p.radius))) return st;
    }
    if ((st = gsx_shard_frame_end(f.l, f.key))) return st;
    if (p.lane != owner) {  // model-changing calls on the owner's stream come after this lane's frame (viewer_bind)
        HIPCHK(hipEventRecord(p.lane->lane_event, p.lane->stream));
        p.lane->lane_busy = true;
    }
    owner->latest = p.lane == owner ? nullptr : p.lane;
    return GSX_OK;
}

struct BusyGuard {
    gsx_viewer* v;
    explicit BusyGuard(gsx_viewer* v_) : v(v_) { v->shard_busy = true; }
    ~BusyGuard() { v->shard_busy = false; }
};

}  // namespace

// viewer_bind (gsx_state.h): any other entry point first finishes the sharded frames in flight (on every rank alike: an
// SPMD host makes the same calls in the same order)
gsx_status gsx::shard_complete_pending(gsx_viewer* v) {
    BusyGuard guard(v);
    while (!v->shard_pending.empty()) {
        gsx_status st = frame_complete(v, v->shard_pending.front());
        v->shard_pending.pop_front();
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    return GSX_OK;
}

extern "C" {

// One index-sharded frame: what a host without Python calls once per frame after gsx_update_camera /
// gsx_update_model_transform.  Afterwards (gsx_sync) gsx_download_framebuffer returns the whole frame on every rank.  The
// sequence is the one documented in include/gsx.h; parallel.ShardedViewer runs the same stage calls with an injectable
// transport for the tests.  One host wait per frame (the verdict), overlapped with the band all-gather.
// gsx_render_options.frames_in_flight = L > 1: frame k is enqueued on lane k mod L BEFORE the verdict of frame k - L + 1 is
// looked at, so the host wait of one frame hides under the device work of the next; the call returns with frame k - L + 1
// complete, gsx_sync (or any readback call) completes the rest.  Collectives keep ONE order on every rank (route_begin).
gsx_status gsx_shard_render_frame(gsx_viewer* v, const char* key, uint32_t shard_records_max, uint32_t speculate, float margin, uint32_t radius) {
    if (!v || !key) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: null argument");
    if (v->parent) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: called on a lane");
    HIPCHK(hipSetDevice(v->device));
    if (!has_comm(v)) return fail(GSX_ERR_RCCL, "gsx_shard_render_frame: no communicator (gsx_viewer_comm_init)");
    if (!find_model(v, key)) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_render_frame: no model '%s'", key);
    BusyGuard guard(v);
    gsx_status st = GSX_OK;
    const uint32_t lanes = std::max(1u, std::min(v->options.frames_in_flight, 4u));
    if (lanes > 1 && !v->comm_stream) HIPCHK(hipStreamCreateWithFlags(&v->comm_stream, hipStreamNonBlocking));
    gsx_viewer* lane = v;
    if (lanes > 1) {
        // (order the owner's stream after lanes whose frames are complete: uploads since then are in the epoch)
        const char* keys[1] = {key};
        if ((st = lane_acquire(v, v->shard_turn++ % lanes, keys, 1, &lane))) return st;
    }
    ShardPending p;
    p.lane = lane;
    p.key = key;
    p.shard_records_max = shard_records_max;
    p.speculate = speculate;
    p.margin = margin;
    p.radius = radius;
    ShardFrame f{lane, key, v->comm_world, v->comm_rank, {}, 0};
    if ((st = frame_enqueue(f, p))) return st;
    v->shard_pending.push_back(p);
    while (v->shard_pending.size() > lanes - 1u) {
        st = frame_complete(v, v->shard_pending.front());
        v->shard_pending.pop_front();
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    return GSX_OK;
}

}  // extern "C"
