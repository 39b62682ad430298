// gsx_api_buffer.cpp — ref-counted handles on a model's per-Gaussian buffers, for readback off the owner's thread.
//
// The reference's buffers are cheaply `Clone`: the export path clones the edit and the mask buffer of every model, moves the
// clones into two spawned threads and downloads there while the UI thread keeps rendering (src/app.rs:769-816,
// src/tab/scene.rs:635-648).  In wgpu a download sees the buffer as of its place in the queue's submission order.  Here:
// gsx_model_buffer_retain (owner thread) enqueues a device-side snapshot of the buffer on the viewer's stream — the contents as
// of this call in stream order, exactly what the clone + download would read — and returns a handle that owns the snapshot;
// gsx_buffer_download may then run on ANY thread, at any time, also after the model or the viewer is gone: it touches
// nothing but the handle (its own stream, an event that says the snapshot is complete).  The snapshot costs its size in HBM
// for as long as a handle lives (edits: 32 B + 1 bit per Gaussian; mask / selection: 1 bit) — 288 GB are there to be used.
#include <atomic>
#include <mutex>

#include "gsx_state.h"

struct gsx_buffer {
    std::atomic<int> refs{1};
    int device = 0;
    gsx_buffer_kind kind = GSX_BUFFER_MASK;
    uint64_t n = 0;        // Gaussians of the model when the handle was taken
    bool present = false;  // false: the model had no mask / selection / edits — the defaults are returned, nothing was copied
    DevBuf words, a, b;    // the snapshot: bitset words | edit record planes
    hipEvent_t ready = nullptr;
    std::mutex mu;         // one download at a time per handle (handles are cheap: clone one per thread instead)
};

extern "C" {

gsx_status gsx_model_buffer_retain(gsx_viewer* v, const char* key, gsx_buffer_kind kind, gsx_buffer** out) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !out) return fail(GSX_ERR_NOT_FOUND, "gsx_model_buffer_retain: no model '%s' / null output", key ? key : "(null)");
    if ((int)kind < 0 || (int)kind > GSX_BUFFER_SELECTION) return fail(GSX_ERR_INVALID_ARG, "gsx_model_buffer_retain: unknown buffer kind %d", (int)kind);
    std::unique_ptr<gsx_buffer> b(new gsx_buffer());
    b->device = v->device;
    b->kind = kind;
    b->n = m->n;
    const size_t words = std::max<size_t>(((size_t)m->n + 31) / 32, 1), n = std::max<size_t>(m->n, 1);
    const DevBuf* bits = kind == GSX_BUFFER_MASK ? &m->mask : kind == GSX_BUFFER_SELECTION ? &m->selection : &m->edited;
    b->present = kind == GSX_BUFFER_MASK ? m->has_mask : kind == GSX_BUFFER_SELECTION ? m->has_selection : m->has_edits;
    if (b->present && bits->bytes < 4 * words) b->present = false;
    if (b->present) {
        HIPCHK(b->words.ensure(4 * words));
        HIPCHK(gsx::op::MemcpyAsync(b->words.p, bits->p, 4 * words, hipMemcpyDeviceToDevice, v->stream));
        if (kind == GSX_BUFFER_EDITS) {
            HIPCHK(b->a.ensure(16 * n));
            HIPCHK(b->b.ensure(16 * n));
            HIPCHK(gsx::op::MemcpyAsync(b->a.p, m->edit_a.p, 16 * n, hipMemcpyDeviceToDevice, v->stream));
            HIPCHK(gsx::op::MemcpyAsync(b->b.p, m->edit_b.p, 16 * n, hipMemcpyDeviceToDevice, v->stream));
        }
    }
    HIPCHK(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    HIPCHK(gsx::op::EventRecord(b->ready, v->stream));
    *out = b.release();
    return GSX_OK;
}

gsx_status gsx_buffer_retain(gsx_buffer* b) {
    if (!b) return fail(GSX_ERR_INVALID_ARG, "gsx_buffer_retain: null handle");
    b->refs.fetch_add(1, std::memory_order_relaxed);
    return GSX_OK;
}

void gsx_buffer_release(gsx_buffer* b) {
    if (!b || b->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    (void)hipSetDevice(b->device);
    if (b->ready) {
        (void)gsx::op::EventSynchronize(b->ready);  // the snapshot copies read and write memory this handle is about to free
        (void)hipEventDestroy(b->ready);
    }
    delete b;
}

gsx_status gsx_buffer_len(gsx_buffer* b, uint64_t* out_elements) {
    if (!b || !out_elements) return fail(GSX_ERR_INVALID_ARG, "gsx_buffer_len: null argument");
    *out_elements = b->kind == GSX_BUFFER_EDITS ? b->n : (b->n + 31) / 32;
    return GSX_OK;
}

gsx_status gsx_buffer_download(gsx_buffer* b, void* out, uint64_t n_elements) {
    if (!b || !out) return fail(GSX_ERR_INVALID_ARG, "gsx_buffer_download: null argument");
    const uint64_t words = (b->n + 31) / 32, expect = b->kind == GSX_BUFFER_EDITS ? b->n : words;
    if (n_elements != expect) return fail(GSX_ERR_INVALID_ARG, "gsx_buffer_download: expected %llu elements", (unsigned long long)expect);
    if (!b->present) {  // gs:: semantics of a buffer nobody wrote: mask all ones (MaskOpTree::Reset), selection empty, default edit pods
        if (b->kind == GSX_BUFFER_MASK) memset(out, 0xFF, 4 * words);
        else if (b->kind == GSX_BUFFER_SELECTION) memset(out, 0, 4 * words);
        else {
            gsx_gaussian_edit def;
            gsx_gaussian_edit_default(&def);
            for (uint64_t i = 0; i < b->n; ++i) static_cast<gsx_gaussian_edit*>(out)[i] = def;
        }
        return GSX_OK;
    }
    std::lock_guard<std::mutex> lock(b->mu);
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = nullptr;  // the caller's thread gets a stream of its own: nothing of the viewer is touched
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    gsx_status st = GSX_OK;
    auto run = [&]() -> gsx_status {
        HIPCHK(gsx::op::StreamWaitEvent(s, b->ready, 0));
        if (b->kind != GSX_BUFFER_EDITS) {
            HIPCHK(gsx::op::MemcpyAsync(out, b->words.p, 4 * words, hipMemcpyDeviceToHost, s));
            HIPCHK(gsx::op::StreamSynchronize(s));
            return GSX_OK;
        }
        std::vector<uint32_t> bits(std::max<uint64_t>(words, 1));
        std::vector<float4> pa(std::max<uint64_t>(b->n, 1)), pb(std::max<uint64_t>(b->n, 1));
        HIPCHK(gsx::op::MemcpyAsync(bits.data(), b->words.p, 4 * words, hipMemcpyDeviceToHost, s));
        HIPCHK(gsx::op::MemcpyAsync(pa.data(), b->a.p, 16 * b->n, hipMemcpyDeviceToHost, s));
        HIPCHK(gsx::op::MemcpyAsync(pb.data(), b->b.p, 16 * b->n, hipMemcpyDeviceToHost, s));
        HIPCHK(gsx::op::StreamSynchronize(s));
        gsx_gaussian_edit def;
        gsx_gaussian_edit_default(&def);
        gsx_gaussian_edit* o = static_cast<gsx_gaussian_edit*>(out);
        for (uint64_t i = 0; i < b->n; ++i) {  // Gaussians never edited read as the default pod (gsx_model_download_edits)
            o[i] = def;
            if (!((bits[i >> 5] >> (i & 31)) & 1u)) continue;
            memcpy(&o[i].flag, &pa[i].x, 4);
            o[i].color[0] = pa[i].y; o[i].color[1] = pa[i].z; o[i].color[2] = pa[i].w;
            o[i].contrast = pb[i].x; o[i].exposure = pb[i].y; o[i].gamma = pb[i].z; o[i].alpha = pb[i].w;
        }
        return GSX_OK;
    };
    st = run();
    (void)hipStreamDestroy(s);
    return st;
}

}  // extern "C"
