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
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (const char* n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) {
        g_rccl.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?");
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

inline ncclComm_t comm_of(gsx_viewer* v) { return static_cast<ncclComm_t>(v->comm); }

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
    if (v->comm) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_init: this viewer already has a communicator");
    if ((st = rccl_ready())) return st;
    ncclUniqueId_ uid;
    memcpy(uid.internal, id, 128);
    ncclComm_t c = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&c, (int)world, uid, (int)rank));
    v->comm = c;
    v->comm_world = world;
    v->comm_rank = rank;
    return GSX_OK;
}

gsx_status gsx_viewer_comm_destroy(gsx_viewer* v) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_comm_destroy: viewer is null");
    if (!v->comm) return GSX_OK;
    (void)hipSetDevice(v->device);
    (void)hipStreamSynchronize(v->stream);
    ncclComm_t c = comm_of(v);
    v->comm = nullptr;
    v->comm_world = 0;
    RCCLCHK(g_rccl.CommDestroy(c));
    return GSX_OK;
}

gsx_status gsx_comm_all_to_all(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_peer) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!v->comm) return fail(GSX_ERR_RCCL, "gsx_comm_all_to_all: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_all_to_all: null buffer");
    if (bytes_per_peer == 0) return GSX_OK;
    RCCLCHK(g_rccl.GroupStart());
    for (uint32_t p = 0; p < v->comm_world; ++p) {
        RCCLCHK(g_rccl.Send(static_cast<const char*>(d_send) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm_of(v), v->stream));
        RCCLCHK(g_rccl.Recv(static_cast<char*>(d_recv) + (size_t)p * bytes_per_peer, bytes_per_peer, kNcclChar, (int)p, comm_of(v), v->stream));
    }
    RCCLCHK(g_rccl.GroupEnd());
    return GSX_OK;
}

gsx_status gsx_comm_all_gather(gsx_viewer* v, const void* d_send, void* d_recv, uint64_t bytes_per_rank) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!v->comm) return fail(GSX_ERR_RCCL, "gsx_comm_all_gather: no communicator (gsx_viewer_comm_init)");
    if (!d_send || !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_comm_all_gather: null buffer");
    if (bytes_per_rank == 0) return GSX_OK;
    RCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, kNcclChar, comm_of(v), v->stream));
    return GSX_OK;
}

// One index-sharded frame, start to finish: what a host without Python calls once per frame after gsx_update_camera /
// gsx_update_model_transform.  Afterwards (gsx_sync) gsx_download_framebuffer returns the whole frame on every rank.  The
// sequence is the one documented in include/gsx.h; parallel.ShardedViewer runs the same stage calls with an injectable
// transport for the tests.  One host wait (the verdict), overlapped with the band all-gather.
gsx_status gsx_shard_render_frame(gsx_viewer* v, const char* key, uint32_t shard_records_max, uint32_t speculate, float margin, uint32_t radius) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!v->comm) return fail(GSX_ERR_RCCL, "gsx_shard_render_frame: no communicator (gsx_viewer_comm_init)");
    const uint32_t world = v->comm_world, rank = v->comm_rank;
    gsx_shard_layout_t lay;
    if ((st = gsx_shard_layout(v, world, rank, &lay))) return st;
    // the padded framebuffer the bands are gathered into, owned by the library
    if (v->ext_fb != v->shard_fb.p || v->shard_fb.bytes < lay.padded_framebuffer_bytes) {
        HIPCHK(hipStreamSynchronize(v->stream));
        if (v->shard_fb.bytes < lay.padded_framebuffer_bytes) {
            HIPCHK(v->shard_fb.ensure(lay.padded_framebuffer_bytes));
            HIPCHK(hipMemsetAsync(v->shard_fb.p, 0, v->shard_fb.bytes, v->stream));
        }
        v->ext_fb = v->shard_fb.p;
        v->ext_fb_bytes = v->shard_fb.bytes;
    }
    uint32_t sat_words = 0;
    if ((st = gsx_shard_feedback_words(v, world, &sat_words))) return st;
    HIPCHK(v->shard_sat_band.ensure(4 * (size_t)sat_words + 16));
    HIPCHK(v->shard_sat_all.ensure((4 * (size_t)sat_words + 16) * world));
    char* fb = static_cast<char*>(v->ext_fb);
    auto exchange_round = [&](uint32_t round, uint32_t T) -> gsx_status {
        const uint64_t per_peer = (uint64_t)(T + 1u) * GSX_RECORD_BYTES;
        HIPCHK(v->shard_send.ensure(per_peer * world));
        HIPCHK(v->shard_recv.ensure(per_peer * world));
        gsx_status s2;
        if ((s2 = gsx_shard_pack_slots(v, key, world, round, v->shard_send.p, T))) return s2;
        if ((s2 = gsx_comm_all_to_all(v, v->shard_send.p, v->shard_recv.p, per_peer))) return s2;
        if ((s2 = gsx_shard_import_slots(v, key, v->shard_recv.p, world, rank, round, T))) return s2;
        if ((s2 = gsx_shard_feedback(v, key, world, rank, v->shard_sat_band.p))) return s2;
        return gsx_comm_all_gather(v, v->shard_sat_band.p, v->shard_sat_all.p, 4 * (uint64_t)sat_words);
    };
    auto finish = [&]() -> gsx_status {  // next frame's limits + the bands, in place: every rank's band lands where it belongs
        gsx_status s2 = gsx_shard_next_windows(v, key, world, v->shard_sat_all.p, margin, radius);
        if (s2) return s2;
        return gsx_comm_all_gather(v, fb + lay.band_offset_bytes, fb, lay.band_bytes);
    };
    if ((st = gsx_shard_frame_begin(v, key, world, rank, speculate, nullptr))) return st;
    uint32_t T = 0, seq = 0;
    if ((st = gsx_shard_slot_records(v, key, world, shard_records_max, &T))) return st;
    gsx_shard_verdict verdict{};
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((st = exchange_round(0, T))) return st;
        if ((st = gsx_shard_verify(v, key, world, v->shard_sat_all.p, &seq))) return st;
        if ((st = finish())) return st;  // before the wait: the usual frame needs nothing more
        if ((st = gsx_shard_wait_verdict(v, key, seq, &verdict))) return st;
        if (!verdict.overflow) break;
        if (attempt == 1) return fail(GSX_ERR_OOM, "gsx_shard_render_frame: an exchange slot of %u records (a whole shard) overflowed: shard_records_max is wrong", T);
        T = std::max<uint32_t>(shard_records_max, 1u);  // a destination can be sent at most a whole shard: this always fits
    }
    if (verdict.need_tiles) {
        HIPCHK(v->shard_counts.ensure(16 * (size_t)(world + 1)));
        char* cnt = static_cast<char*>(v->shard_counts.p);
        if ((st = gsx_shard_repair_count(v, key, world, cnt + 16 * (size_t)world))) return st;
        if ((st = gsx_comm_all_gather(v, cnt + 16 * (size_t)world, cnt, 16))) return st;
        if ((st = gsx_shard_post_counts(v, world, cnt, &seq))) return st;
        gsx_shard_verdict sized{};
        if ((st = gsx_shard_wait_verdict(v, nullptr, seq, &sized))) return st;
        if ((st = exchange_round(1, std::max<uint32_t>(sized.max_records, 1u)))) return st;
        if ((st = finish())) return st;
    }
    return gsx_shard_frame_end(v, key);
}

}  // extern "C"
