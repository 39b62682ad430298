// gsx_comm.cpp — the collectives of the multi-GPU path inside libgsx, over RCCL (xGMI): one communicator per viewer — one
// per LANE when sharded frames are in flight — every call enqueued on the stream of the viewer / lane whose frame it serves.  No reference counterpart: the reference renders on one wgpu device
// (src/main.rs:85-98).  RCCL is loaded at run time (dlopen), so that a single-GPU host needs nothing but the HIP runtime and
// `ldd libgsx.so` stays what it was; a missing or failing RCCL is GSX_ERR_RCCL, never an abort.
//
// The path has three collectives:
//   all-to-all of fixed-size record slots   grouped ncclSend / ncclRecv: point-to-point, all 7 xGMI links of a GPU busy at once
//   all-gather of the per-tile saturation keys (32 KB per frame at 1080p)
//   all-gather of the framebuffer bands, in place (the bands are disjoint: "reduce of per-GPU tile fragments" with no arithmetic)
// and gsx_shard_render_frame (gsx_shard_frame.cpp) strings the stage calls of gsx_api_shard.cpp and these together: one call
// per frame.
//
// Frames in flight (gsx_render_options.frames_in_flight = L): collectives on ONE communicator execute in issue order, so a
// single communicator would make frame k+1's exchange wait for frame k's last gather — which sits behind frame k's
// compositor (round 2 measured exactly that: 1230 -> 1015 -> 928 fps with 1 / 2 / 3 lanes).  Every lane therefore gets a
// communicator of its own (created on first use: rank 0 draws a unique id, ncclBroadcast over the viewer's communicator
// carries it, every rank joins — a collective step, once) and its collectives run on the lane's own stream: frames in
// flight are independent pipelines on the device, and every rank issues every communicator's calls in the same order because
// every branch of the frame loop is decided by globally gathered data.
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
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    // introspection only (gsx_viewer_comm_info): a library without them still runs the frame
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
    GSX_SYM(Broadcast, "ncclBroadcast");
    GSX_SYM(Send, "ncclSend");
    GSX_SYM(Recv, "ncclRecv");
    GSX_SYM(GroupStart, "ncclGroupStart");
    GSX_SYM(GroupEnd, "ncclGroupEnd");
#undef GSX_SYM
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    g_rccl.CommCuDevice = reinterpret_cast<decltype(g_rccl.CommCuDevice)>(dlsym(h, "ncclCommCuDevice"));
    g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(dlsym(h, "ncclGetVersion"));
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

// Inside an open ncclGroupStart nothing may return: the group has to be closed whatever a Send / Recv answered, or every later
// RCCL call of the thread lands in a group nobody ends.  GROUPED remembers the first failure and skips the rest; group_end
// closes the group and reports it.
struct RcclGroup {
    ncclResult_t first = 0;
    const char* what = nullptr;
    int line = 0;
};
#define GROUPED(grp, expr)                                   \
    do {                                                     \
        if ((grp).first == 0) {                              \
            (grp).first = (expr);                            \
            if ((grp).first != 0) {                          \
                (grp).what = #expr;                          \
                (grp).line = __LINE__;                       \
            }                                                \
        }                                                    \
    } while (0)
inline gsx_status group_end(RcclGroup& grp) {
    const ncclResult_t e = g_rccl.GroupEnd();
    if (grp.first != 0) return fail(GSX_ERR_RCCL, "%s failed: %s (%s:%d)", grp.what, g_rccl.GetErrorString(grp.first), __FILE__, grp.line);
    if (e != 0) return fail(GSX_ERR_RCCL, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(e));
    return GSX_OK;
}

inline gsx_viewer* owner_of(gsx_viewer* v) { return v->parent ? v->parent : v; }
// the communicator of a viewer (lane 0) or of one of its lanes
inline ncclComm_t comm_of(gsx_viewer* v) {
    gsx_viewer* o = owner_of(v);
    if (!v->parent || v->lane_index == 0) return static_cast<ncclComm_t>(o->comm);
    return v->lane_index - 1 < o->lane_comms.size() ? static_cast<ncclComm_t>(o->lane_comms[v->lane_index - 1]) : nullptr;
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

gsx_status gsx_viewer_comm_info(gsx_viewer* v, gsx_comm_info* out) {
    if (!v || !out) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_info: null argument");
    *out = gsx_comm_info{0u, 0u, 0u, 0u, -1, 0};
    if (!has_comm(v)) return GSX_OK;
    out->nranks = v->comm_world;
    out->rank = v->comm_rank;
    out->device = v->device;
    out->lane_comms = (uint32_t)v->lane_comms.size();
    if (v->comm) {  // RCCL: the communicator's own answers
        out->transport = 1;
        ncclComm_t c = static_cast<ncclComm_t>(v->comm);
        int x = 0;
        if (g_rccl.CommCount && g_rccl.CommCount(c, &x) == 0) out->nranks = (uint32_t)x;
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(c, &x) == 0) out->rank = (uint32_t)x;
        if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(c, &x) == 0) out->device = x;
        if (g_rccl.GetVersion && g_rccl.GetVersion(&x) == 0) out->version = x;
    } else {
        out->transport = v->comm_group ? 2u : 3u;
    }
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

gsx_status gsx_viewer_comm_init_custom_v(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_comm_all_to_all_v_fn all_to_all_v,
                                         gsx_comm_gather_v_fn gather_v, void* ctx) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!all_to_all_v || !gather_v || world == 0 || world > 64 || rank >= world)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_custom_v: null function or bad world/rank %u/%u", world, rank);
    if (has_comm(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_custom_v: this viewer already has a communicator");
    v->comm_a2a_v_fn = all_to_all_v;
    v->comm_gather_v_fn = gather_v;
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
    for (gsx_viewer* l : v->lanes) (void)gsx::op::StreamSynchronize(l->stream);
    (void)gsx::op::StreamSynchronize(v->stream);
    v->comm_world = 0;
    if (v->comm_a2a_fn || v->comm_a2a_v_fn) {
        if (v->comm_group) group_leave(v);
        v->comm_a2a_fn = nullptr;
        v->comm_ag_fn = nullptr;
        v->comm_a2a_v_fn = nullptr;
        v->comm_gather_v_fn = nullptr;
        v->comm_ctx = nullptr;
        return GSX_OK;
    }
    for (void* lc : v->lane_comms)
        if (lc) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(lc));
    v->lane_comms.clear();
    ncclComm_t c = static_cast<ncclComm_t>(v->comm);
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
    if (!o->comm && !o->comm_a2a_fn) {  // a transport of unequal pieces only: equal ones are a case of those
        PeerSpans sp{};
        for (uint32_t p = 0; p < o->comm_world; ++p) {
            sp.off[p] = (uint64_t)p * bytes_per_peer;
            sp.bytes[p] = bytes_per_peer;
        }
        return comm_all_to_all_v(v, d_send, sp, d_recv, sp);
    }
    o->shard_stats.wire_bytes += (uint64_t)(o->comm_world - 1u) * bytes_per_peer;
    if (o->comm_a2a_fn) {  // the caller's transport (or the in-process group): the whole exchange, own slot included
        g_err.clear();
        trace_flush();  // (the caller's function enqueues on the stream itself: what is still recorded must be in front of it)
        if ((st = o->comm_a2a_fn(o->comm_ctx, d_send, d_recv, bytes_per_peer, v->stream)))
            return g_err.empty() ? fail(st, "gsx_comm_all_to_all: the custom transport failed with status %d", (int)st) : st;
        return GSX_OK;
    }
    // what this rank keeps for itself does not travel: a device copy on the viewer's own stream, beside the exchange
    // (GSX_COMM_SELF_VIA_RCCL, read by gsx_viewer_comm_init: send it to oneself through RCCL like everything else — the
    // one-rank tests on a one-GPU box exercise ncclSend / ncclRecv that way)
    const bool bypass = !o->comm_self_via_rccl;
    if (bypass) {
        const size_t self_off = (size_t)o->comm_rank * bytes_per_peer;
        HIPCHK(gsx::op::MemcpyAsync(static_cast<char*>(d_recv) + self_off, static_cast<const char*>(d_send) + self_off, bytes_per_peer,
                              hipMemcpyDeviceToDevice, v->stream));
        if (o->comm_world == 1) return GSX_OK;
    }
    ncclComm_t comm = comm_of(v);
    if (!comm) return fail(GSX_ERR_RCCL, "gsx_comm_all_to_all: lane %u has no communicator", v->lane_index);
    trace_flush();  // (RCCL enqueues on the stream itself)
    RCCLCHK(g_rccl.GroupStart());
    RcclGroup grp;
    for (uint32_t p = 0; p < o->comm_world; ++p) {
        if (bypass && p == o->comm_rank) continue;
        GROUPED(grp, g_rccl.Send(static_cast<const char*>(d_send) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm, v->stream));
        GROUPED(grp, g_rccl.Recv(static_cast<char*>(d_recv) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm, v->stream));
    }
    return group_end(grp);
}

gsx_status gsx_comm_all_gather(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_rank) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* o = owner_of(v);
    if (!has_comm(o)) return fail(GSX_ERR_RCCL, "gsx_comm_all_gather: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_all_gather: null buffer");
    if (bytes_per_rank == 0) return GSX_OK;
    if (!o->comm && !o->comm_ag_fn) {
        PeerSpans sp{};
        for (uint32_t p = 0; p < o->comm_world; ++p) {
            sp.off[p] = (uint64_t)p * bytes_per_rank;
            sp.bytes[p] = bytes_per_rank;
        }
        return comm_gather_v(v, d_send, bytes_per_rank, d_recv, sp, -1);
    }
    o->shard_stats.wire_bytes += (uint64_t)(o->comm_world - 1u) * bytes_per_rank;
    if (o->comm_ag_fn) {
        g_err.clear();
        trace_flush();  // (the caller's function enqueues on the stream itself: what is still recorded must be in front of it)
        if ((st = o->comm_ag_fn(o->comm_ctx, d_send, d_recv, bytes_per_rank, v->stream)))
            return g_err.empty() ? fail(st, "gsx_comm_all_gather: the custom transport failed with status %d", (int)st) : st;
        return GSX_OK;
    }
    if (o->comm_world == 1 && !o->comm_self_via_rccl) {  // one rank: its own piece is all there is, and it does not travel
        if (d_send != d_recv) HIPCHK(gsx::op::MemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, v->stream));
        return GSX_OK;
    }
    ncclComm_t comm = comm_of(v);
    if (!comm) return fail(GSX_ERR_RCCL, "gsx_comm_all_gather: lane %u has no communicator", v->lane_index);
    trace_flush();
    RCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, kNcclChar, comm, v->stream));
    return GSX_OK;
}

}  // extern "C"

// ---- pieces of unequal size: balanced bands, slots sized pair by pair (gsx_shard_frame.cpp) ----
// Over RCCL both are grouped ncclSend / ncclRecv — point-to-point, every xGMI link of the GPU busy at once, each message as long
// as its piece; what a rank keeps for itself is a device copy beside the exchange.
gsx_status gsx::comm_all_to_all_v(gsx_viewer* v, const void* d_send, const PeerSpans& snd, void* d_recv, const PeerSpans& rcv) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* o = owner_of(v);
    if (!has_comm(o)) return fail(GSX_ERR_RCCL, "all-to-all: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "all-to-all: null buffer");
    const uint32_t world = o->comm_world, me = o->comm_rank;
    for (uint32_t p = 0; p < world; ++p)
        if (p != me) o->shard_stats.wire_bytes += snd.bytes[p];
    if (o->comm_a2a_v_fn) {
        g_err.clear();
        trace_flush();  // (the caller's function enqueues on the stream itself: what is still recorded must be in front of it)
        if ((st = o->comm_a2a_v_fn(o->comm_ctx, d_send, snd.off, snd.bytes, d_recv, rcv.off, rcv.bytes, v->stream)))
            return g_err.empty() ? fail(st, "all-to-all: the custom transport failed with status %d", (int)st) : st;
        return GSX_OK;
    }
    if (!o->comm) return fail(GSX_ERR_UNSUPPORTED, "all-to-all of unequal slots: this transport moves equal pieces only (gsx_viewer_comm_init_custom_v)");
    const bool bypass = !o->comm_self_via_rccl;
    if (bypass) {
        if (snd.bytes[me] != rcv.bytes[me]) return fail(GSX_ERR_INVALID_ARG, "all-to-all: own slot of %llu bytes sent, %llu expected", (unsigned long long)snd.bytes[me], (unsigned long long)rcv.bytes[me]);
        if (snd.bytes[me])
            HIPCHK(gsx::op::MemcpyAsync(static_cast<char*>(d_recv) + rcv.off[me], static_cast<const char*>(d_send) + snd.off[me], snd.bytes[me], hipMemcpyDeviceToDevice, v->stream));
        if (world == 1) return GSX_OK;
    }
    ncclComm_t comm = comm_of(v);
    if (!comm) return fail(GSX_ERR_RCCL, "all-to-all: lane %u has no communicator", v->lane_index);
    trace_flush();
    RCCLCHK(g_rccl.GroupStart());
    RcclGroup grp;
    for (uint32_t p = 0; p < world; ++p) {
        if (bypass && p == me) continue;
        if (snd.bytes[p]) GROUPED(grp, g_rccl.Send(static_cast<const char*>(d_send) + snd.off[p], snd.bytes[p], kNcclChar, (int)p, comm, v->stream));
        if (rcv.bytes[p]) GROUPED(grp, g_rccl.Recv(static_cast<char*>(d_recv) + rcv.off[p], rcv.bytes[p], kNcclChar, (int)p, comm, v->stream));
    }
    return group_end(grp);
}

gsx_status gsx::comm_gather_v(gsx_viewer* v, const void* d_send, uint64_t send_bytes, void* d_recv, const PeerSpans& rcv, int32_t root) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_viewer* o = owner_of(v);
    if (!has_comm(o)) return fail(GSX_ERR_RCCL, "gather: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gather: null buffer");
    const uint32_t world = o->comm_world, me = o->comm_rank;
    if (root >= (int32_t)world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_gather_root: root %d of %u ranks", (int)root, world);
    if (rcv.bytes[me] != send_bytes) return fail(GSX_ERR_INVALID_ARG, "gather: own piece of %llu bytes sent, %llu expected", (unsigned long long)send_bytes, (unsigned long long)rcv.bytes[me]);
    o->shard_stats.wire_bytes += root < 0 ? (uint64_t)(world - 1u) * send_bytes : ((uint32_t)root == me ? 0u : send_bytes);
    if (o->comm_gather_v_fn) {
        g_err.clear();
        trace_flush();  // (the caller's function enqueues on the stream itself: what is still recorded must be in front of it)
        if ((st = o->comm_gather_v_fn(o->comm_ctx, d_send, send_bytes, d_recv, rcv.off, rcv.bytes, root, v->stream)))
            return g_err.empty() ? fail(st, "gather: the custom transport failed with status %d", (int)st) : st;
        return GSX_OK;
    }
    if (!o->comm) {  // equal pieces only: the transport's all-gather, when the pieces happen to be equal and in rank order
        for (uint32_t p = 0; p < world; ++p)
            if (rcv.bytes[p] != send_bytes || rcv.off[p] != (uint64_t)p * send_bytes)
                return fail(GSX_ERR_UNSUPPORTED, "gather of unequal pieces: this transport moves equal pieces only (gsx_viewer_comm_init_custom_v)");
        o->shard_stats.wire_bytes -= root < 0 ? (uint64_t)(world - 1u) * send_bytes : ((uint32_t)root == me ? 0u : send_bytes);  // (counted by the call below)
        return gsx_comm_all_gather(v, d_send, d_recv, send_bytes);
    }
    char* own = static_cast<char*>(d_recv) + rcv.off[me];
    const bool receives = root < 0 || (uint32_t)root == me;
    const bool self_copy = !o->comm_self_via_rccl || root >= 0;
    if (receives && self_copy && own != d_send && send_bytes) HIPCHK(gsx::op::MemcpyAsync(own, d_send, send_bytes, hipMemcpyDeviceToDevice, v->stream));
    if (world == 1 && self_copy) return GSX_OK;
    ncclComm_t comm = comm_of(v);
    if (!comm) return fail(GSX_ERR_RCCL, "gather: lane %u has no communicator", v->lane_index);
    trace_flush();
    RCCLCHK(g_rccl.GroupStart());
    RcclGroup grp;
    for (uint32_t p = 0; p < world; ++p) {
        if (p == me && self_copy) continue;
        const bool p_receives = root < 0 || (uint32_t)root == p;
        if (p_receives && send_bytes) GROUPED(grp, g_rccl.Send(d_send, send_bytes, kNcclChar, (int)p, comm, v->stream));
        if (receives && rcv.bytes[p]) GROUPED(grp, g_rccl.Recv(static_cast<char*>(d_recv) + rcv.off[p], rcv.bytes[p], kNcclChar, (int)p, comm, v->stream));
    }
    return group_end(grp);
}

// One communicator per lane (see the head of this file).  Collective: every rank calls it with the same `lanes` — they do,
// it follows from gsx_render_options.frames_in_flight, which an SPMD host sets alike everywhere.
gsx_status gsx::comm_ensure_lanes(gsx_viewer* v, uint32_t lanes) {
    if (!v->comm || lanes <= 1) return GSX_OK;  // a custom transport serves every lane with the same two functions
    // one rank: nothing of a frame goes through RCCL (own slot and own pieces are device copies), so no lane needs a communicator.
    // (Measured with ROCm 7.2's RCCL 2.27.7: a second communicator in the process costs 1413 -> 1214 fps at world 1 although no
    // RCCL call is made in the loop; the 2.26.6 that ships with PyTorch does not show it.)
    if (v->comm_world == 1 && !v->comm_self_via_rccl) return GSX_OK;
    gsx_status st = rccl_ready();
    if (st) return st;
    while (v->lane_comms.size() + 1 < lanes) {
        ncclUniqueId_ uid{};
        HIPCHK(v->scratch.ensure(128));
        if (v->comm_rank == 0) {
            RCCLCHK(g_rccl.GetUniqueId(&uid));
            HIPCHK(gsx::op::MemcpyAsync(v->scratch.p, uid.internal, 128, hipMemcpyHostToDevice, v->stream));
        }
        trace_flush();
        RCCLCHK(g_rccl.Broadcast(v->scratch.p, v->scratch.p, 128, kNcclChar, 0, static_cast<ncclComm_t>(v->comm), v->stream));
        HIPCHK(gsx::op::MemcpyAsync(uid.internal, v->scratch.p, 128, hipMemcpyDeviceToHost, v->stream));
        HIPCHK(gsx::op::StreamSynchronize(v->stream));
        ncclComm_t c = nullptr;
        RCCLCHK(g_rccl.CommInitRank(&c, (int)v->comm_world, uid, (int)v->comm_rank));
        v->lane_comms.push_back(c);
    }
    return GSX_OK;
}
