// gsx_comm_group.cpp — the in-process transport of the multi-GPU path: one host process, one host thread + one viewer per
// GPU (or, in the tests, several viewers on ONE GPU), the two collectives of the sharded frame as device-to-device copies
// ordered by HIP events.  No reference counterpart (src/main.rs:85-98: one wgpu device); a single-process host like the egui
// app would drive its GPUs this way, and it is how tests/test_gpu_shard_lib.py runs gsx_shard_render_frame's own control flow
// with 2 ... 8 ranks on a one-GPU box.  Implemented on the custom-transport interface of include/gsx.h for pieces of unequal size
// (gsx_viewer_comm_init_custom_v: two functions that enqueue on a stream), like any transport a caller could bring.
//
// Frames in flight: a viewer and its lanes share one seat; the host issues their collectives one after the other (one thread),
// each on the stream of the lane whose frame it serves, so frames on different lanes overlap on the device.
//
// A collective, seen from rank r (every rank runs the same sequence, gsx_comm.cpp guarantees one order of collectives):
//   record `ready` on r's stream, publish {send, per-peer spans}       -- what r contributes exists once `ready` fires
//   host rendezvous A                                                  -- everybody has published; sizes are compared
//   for every source p, in rank order: wait for p's `ready`, copy p's piece for r into r's receive buffer   (on r's stream)
//   record `done` on r's stream; host rendezvous B; wait for every peer's `done` (the event handles are read BEFORE B)
//                                                                      -- r's send buffer may be reused: every reader is behind
// The host threads meet twice per collective; the DEVICE never waits for a host (copies and waits are enqueued).  A rendezvous
// that is not complete after timeout_ms aborts the group: every rank's call — the waiting one and all later ones — returns
// GSX_ERR_RCCL with the reason.  Nothing hangs.
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>

#include "gsx_state.h"

namespace {

enum Op : int { OP_NONE = 0, OP_ALL_TO_ALL = 1, OP_GATHER = 2 /* + (root + 1) << 8: 0 = every rank receives */ };

struct Seat {
    gsx_comm_group* group = nullptr;
    uint32_t rank = 0;
    bool taken = false;
    int device = 0;
    // one event pair per stream this seat has issued collectives on (a viewer and its lanes: frames in flight)
    std::map<hipStream_t, std::pair<hipEvent_t, hipEvent_t>> events;
    // published for the collective in flight
    hipEvent_t ready = nullptr, done = nullptr;
    const char* send = nullptr;
    uint64_t send_off[64]{}, send_bytes[64]{};  // what this rank has for every peer (a gather: the same piece for everybody)
    int op = OP_NONE;
};

}  // namespace

struct gsx_comm_group {
    uint32_t world = 0;
    uint32_t timeout_ms = 60000;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    std::string why;
    Seat seats[64];
};

namespace {

gsx_status group_abort(gsx_comm_group* g, const std::string& why) {  // (mu held)
    if (!g->aborted) {
        g->aborted = true;
        g->why = why;
    }
    g->cv.notify_all();
    return fail(GSX_ERR_RCCL, "in-process group: %s", g->why.c_str());
}

// all `world` ranks, or GSX_ERR_RCCL for everybody
gsx_status rendezvous(gsx_comm_group* g, uint32_t rank, const char* what) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->aborted) return fail(GSX_ERR_RCCL, "in-process group: %s", g->why.c_str());
    const uint64_t gen = g->generation;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        g->generation += 1;
        g->cv.notify_all();
        return GSX_OK;
    }
    const bool ok = g->cv.wait_for(lk, std::chrono::milliseconds(g->timeout_ms), [&] { return g->generation != gen || g->aborted; });
    if (g->generation != gen) return GSX_OK;  // complete (whatever happened to the group a moment later)
    if (g->aborted) return fail(GSX_ERR_RCCL, "in-process group: %s", g->why.c_str());
    if (!ok) {
        char buf[256];
        snprintf(buf, sizeof buf, "rank %u waited %u ms for its peers in %s (a rank left the frame loop, or the ranks disagree about the "
                 "sequence of collectives)", rank, g->timeout_ms, what);
        return group_abort(g, buf);
    }
    return GSX_OK;
}

// snd: what this rank has for peer p (bytes at an offset of d_send); rcv: where peer p's piece for this rank goes and how long the
// ranks agreed it is (a peer that publishes another length aborts the group: sizes are derived from gathered data, they cannot
// differ unless the ranks ran different frames)
gsx_status collective(Seat* me, int op, const void* d_send, const uint64_t* snd_off, const uint64_t* snd_bytes, void* d_recv, const uint64_t* rcv_off,
                      const uint64_t* rcv_bytes, hipStream_t stream) {
    gsx_comm_group* g = me->group;
    const int kind = op & 0xFF;
    const int32_t root = (int32_t)((uint32_t)op >> 8) - 1;  // OP_GATHER: the only rank that receives (-1: every rank)
    const char* what = kind == OP_ALL_TO_ALL ? "an all-to-all" : "a gather";
    auto ev = me->events.find(stream);
    if (ev == me->events.end()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        ev = me->events.emplace(stream, std::make_pair(a, b)).first;
    }
    me->ready = ev->second.first;
    me->done = ev->second.second;
    HIPCHK(gsx::op::EventRecord(me->ready, stream));
    me->send = static_cast<const char*>(d_send);
    for (uint32_t p = 0; p < g->world; ++p) {
        me->send_off[p] = snd_off[p];
        me->send_bytes[p] = snd_bytes[p];
    }
    me->op = op;
    gsx_status st = rendezvous(g, me->rank, what);
    if (st) return st;
    // every rank compares what its peers published with what it expects of them
    const bool receives = kind == OP_ALL_TO_ALL || root < 0 || (uint32_t)root == me->rank;
    for (uint32_t p = 0; p < g->world; ++p)
        if (g->seats[p].op != op || (receives && g->seats[p].send_bytes[me->rank] != rcv_bytes[p])) {
            std::unique_lock<std::mutex> lk(g->mu);
            char buf[256];
            snprintf(buf, sizeof buf, "ranks disagree: rank %u is in %s and expects %llu bytes of rank %u, which is in op %d with %llu for it", me->rank, what,
                     (unsigned long long)rcv_bytes[p], p, g->seats[p].op, (unsigned long long)g->seats[p].send_bytes[me->rank]);
            return group_abort(g, buf);
        }
    // the peers' `done` events of THIS collective, taken while the published fields are stable (between the two rendezvous): after
    // rendezvous B a faster peer may already be publishing its next collective, on another lane's stream with that lane's event pair
    hipEvent_t peer_done[64];
    for (uint32_t p = 0; p < g->world; ++p) peer_done[p] = g->seats[p].done;
    for (uint32_t p = 0; receives && p < g->world; ++p) {  // delivery by source rank, like ncclRecv from p
        const Seat& src = g->seats[p];
        const char* from = src.send + src.send_off[me->rank];
        char* to = static_cast<char*>(d_recv) + rcv_off[p];
        if (from == to || rcv_bytes[p] == 0) continue;  // an in-place gather's own piece / nothing to move
        if (p != me->rank) HIPCHK(gsx::op::StreamWaitEvent(stream, src.ready, 0));
        HIPCHK(gsx::op::MemcpyAsync(to, from, rcv_bytes[p], hipMemcpyDefault, stream));
    }
    HIPCHK(gsx::op::EventRecord(me->done, stream));
    if ((st = rendezvous(g, me->rank, what))) return st;
    for (uint32_t p = 0; p < g->world; ++p)
        if (p != me->rank) HIPCHK(gsx::op::StreamWaitEvent(stream, peer_done[p], 0));
    return GSX_OK;
}

gsx_status group_all_to_all_v(void* ctx, const void* d_send, const uint64_t* send_offsets, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_offsets,
                              const uint64_t* recv_bytes, void* hip_stream) {
    return collective(static_cast<Seat*>(ctx), OP_ALL_TO_ALL, d_send, send_offsets, send_bytes, d_recv, recv_offsets, recv_bytes, static_cast<hipStream_t>(hip_stream));
}

gsx_status group_gather_v(void* ctx, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* recv_offsets, const uint64_t* recv_bytes, int32_t root,
                          void* hip_stream) {
    Seat* me = static_cast<Seat*>(ctx);
    uint64_t off[64], bytes[64];
    for (uint32_t p = 0; p < me->group->world; ++p) {  // the same piece for every rank that receives
        off[p] = 0;
        bytes[p] = (root < 0 || (uint32_t)root == p) ? send_bytes : 0;
    }
    return collective(me, OP_GATHER | (int)((uint32_t)(root + 1) << 8), d_send, off, bytes, d_recv, recv_offsets, recv_bytes, static_cast<hipStream_t>(hip_stream));
}

}  // namespace

void gsx::group_leave(gsx_viewer* v) {
    gsx_comm_group* g = v->comm_group;
    Seat* me = static_cast<Seat*>(v->comm_ctx);
    if (!g || !me) return;
    std::unique_lock<std::mutex> lk(g->mu);
    for (auto& kv : me->events) {
        (void)hipEventDestroy(kv.second.first);
        (void)hipEventDestroy(kv.second.second);
    }
    me->events.clear();
    me->ready = me->done = nullptr;
    me->taken = false;
    // a rank that leaves while the others still render: they find out at their next rendezvous, at once
    if (!g->aborted) {
        g->aborted = true;
        g->why = "rank " + std::to_string(me->rank) + " destroyed its communicator";
    }
    g->cv.notify_all();
    v->comm_group = nullptr;
}

extern "C" {

gsx_status gsx_comm_group_create(uint32_t world, uint32_t timeout_ms, gsx_comm_group** out) {
    if (!out || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_group_create: world must be 1..64");
    gsx_comm_group* g = new gsx_comm_group();
    g->world = world;
    g->timeout_ms = timeout_ms ? timeout_ms : 60000u;
    for (uint32_t r = 0; r < world; ++r) {
        g->seats[r].group = g;
        g->seats[r].rank = r;
    }
    *out = g;
    return GSX_OK;
}

void gsx_comm_group_destroy(gsx_comm_group* g) { delete g; }

gsx_status gsx_viewer_comm_init_group(gsx_viewer* v, gsx_comm_group* g, uint32_t rank) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!g || rank >= g->world) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_group: null group or rank %u out of range", rank);
    if (has_comm(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_group: this viewer already has a communicator");
    Seat* me = &g->seats[rank];
    {
        std::unique_lock<std::mutex> lk(g->mu);
        if (me->taken) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init_group: rank %u of this group is taken", rank);
        if (g->aborted) return fail(GSX_ERR_RCCL, "in-process group: %s", g->why.c_str());
        me->taken = true;
    }
    me->device = v->device;
    // several devices in one process: direct peer copies over xGMI where the platform allows (otherwise the runtime stages them)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) == hipSuccess)
        for (int d = 0; d < n_dev; ++d) {
            int can = 0;
            if (d != v->device && hipDeviceCanAccessPeer(&can, v->device, d) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(d, 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled: fine
            }
        }
    if ((st = gsx_viewer_comm_init_custom_v(v, g->world, rank, group_all_to_all_v, group_gather_v, me))) {
        std::unique_lock<std::mutex> lk(g->mu);
        me->taken = false;
        return st;
    }
    v->comm_group = g;
    return GSX_OK;
}

}  // extern "C"
