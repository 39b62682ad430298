// gsx_api_edit.cpp — C ABI for selection, per-Gaussian edits and queries (spec/RENDER_SPEC.md section 7; kernels_edit.hip).
#include "gsx_state.h"

using namespace gsx;

extern "C" {

// ---- selection / edits / queries ----
void gsx_gaussian_edit_default(gsx_gaussian_edit* e) {
    if (!e) return;
    *e = gsx_gaussian_edit{0u, {0.0f, 1.0f, 1.0f}, 0.0f, 0.0f, 1.0f, 1.0f};
}

gsx_status gsx_update_query(gsx_viewer* v, const gsx_query* q) {
    if (!v || !q) return fail(GSX_ERR_INVALID_ARG, "gsx_update_query: null argument");
    if (q->kind > GSX_QUERY_TEXTURE || q->selection_op > GSX_SELECTION_REMOVE)
        return fail(GSX_ERR_INVALID_ARG, "gsx_update_query: unknown kind %u / selection op %u", q->kind, q->selection_op);
    v->query = *q;
    return GSX_OK;
}

gsx_status gsx_update_query_texture(gsx_viewer* v, const uint8_t* texels, uint32_t width, uint32_t height) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!texels || width != v->width || height != v->height)
        return fail(GSX_ERR_INVALID_ARG, "gsx_update_query_texture: need %ux%u texels (the viewport)", v->width, v->height);
    HIPCHK(v->query_texture.ensure((size_t)width * height));
    HIPCHK(gsx::op::MemcpyAsync(v->query_texture.p, texels, (size_t)width * height, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    v->query_tex_w = width;
    v->query_tex_h = height;
    return GSX_OK;
}

gsx_status gsx_update_selection_highlight(gsx_viewer* v, const float rgba[4]) {
    if (!v || !rgba) return fail(GSX_ERR_INVALID_ARG, "gsx_update_selection_highlight: null argument");
    memcpy(v->highlight, rgba, sizeof v->highlight);
    return GSX_OK;
}

gsx_status gsx_update_selection_edit(gsx_viewer* v, const gsx_gaussian_edit* e) {
    if (!v || !e) return fail(GSX_ERR_INVALID_ARG, "gsx_update_selection_edit: null argument");
    v->sel_edit = *e;
    return GSX_OK;
}

gsx_status gsx_model_show_unedited(gsx_viewer* v, const char* key, uint32_t on) {
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_show_unedited: no model '%s'", key ? key : "(null)");
    m->show_unedited = on != 0;
    return GSX_OK;
}

gsx_status gsx_postprocess(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_postprocess: no model '%s'", key ? key : "(null)");
    if (m->flags_kind == GSX_QUERY_RECT || m->flags_kind == GSX_QUERY_BRUSH || m->flags_kind == GSX_QUERY_TEXTURE) {
        if ((st = ensure_selection(v, m))) return st;
        HIPCHK(launch_selection_op(v->stream, (uint32_t)(((size_t)m->n + 31) / 32), m->flags_op, m->query_flags.as<uint32_t>(),
                                   m->selection.as<uint32_t>()));
        m->has_selection = true;
        m->edit_epoch += 1;
        m->flags_kind = GSX_QUERY_NONE;  // consumed: one selection op per evaluated query
    }
    return GSX_OK;
}

gsx_status gsx_model_upload_selection(gsx_viewer* v, const char* key, const uint32_t* words, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_selection: no model '%s'", key ? key : "(null)");
    m->edit_epoch += 1;
    if (!words) {  // clear
        m->has_selection = false;
        if (m->selection.p) HIPCHK(gsx::op::MemsetAsync(m->selection.p, 0, m->selection.bytes, v->stream));
        return GSX_OK;
    }
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_selection: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    if ((st = ensure_selection(v, m))) return st;
    HIPCHK(gsx::op::MemcpyAsync(m->selection.p, words, 4 * n_words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    m->has_selection = true;
    return GSX_OK;
}

gsx_status gsx_model_download_selection(gsx_viewer* v, const char* key, uint32_t* words, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !words) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_selection: no model '%s'", key ? key : "(null)");
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_selection: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    if (!m->has_selection) {
        memset(words, 0, 4 * n_words);
        return GSX_OK;
    }
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(words, m->selection.p, 4 * n_words, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_model_download_edits(gsx_viewer* v, const char* key, gsx_gaussian_edit* out, uint64_t n) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !out) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_edits: no model '%s'", key ? key : "(null)");
    if (n != m->n) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_edits: expected %llu records", (unsigned long long)m->n);
    gsx_gaussian_edit def;
    gsx_gaussian_edit_default(&def);
    for (uint64_t i = 0; i < n; ++i) out[i] = def;
    if (!m->has_edits) return GSX_OK;
    const size_t words = ((size_t)n + 31) / 32;
    std::vector<uint32_t> bits(words);
    std::vector<float4> a(n), b(n);
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(bits.data(), m->edited.p, 4 * words, hipMemcpyDeviceToHost));
    HIPCHK(gsx::op::Memcpy(a.data(), m->edit_a.p, 16 * n, hipMemcpyDeviceToHost));
    HIPCHK(gsx::op::Memcpy(b.data(), m->edit_b.p, 16 * n, hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < n; ++i) {
        if (!((bits[i >> 5] >> (i & 31)) & 1u)) continue;
        memcpy(&out[i].flag, &a[i].x, 4);
        out[i].color[0] = a[i].y; out[i].color[1] = a[i].z; out[i].color[2] = a[i].w;
        out[i].contrast = b[i].x; out[i].exposure = b[i].y; out[i].gamma = b[i].z; out[i].alpha = b[i].w;
    }
    return GSX_OK;
}

gsx_status gsx_model_upload_edits(gsx_viewer* v, const char* key, const gsx_gaussian_edit* edits, uint64_t n) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_edits: no model '%s'", key ? key : "(null)");
    m->edit_epoch += 1;
    if (!edits) {  // drop every stored edit
        m->has_edits = false;
        if (m->edited.p) HIPCHK(gsx::op::MemsetAsync(m->edited.p, 0, m->edited.bytes, v->stream));
        return GSX_OK;
    }
    if (n != m->n) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_edits: expected %llu records", (unsigned long long)m->n);
    if ((st = ensure_edit_buffers(v, m))) return st;
    const size_t words = ((size_t)n + 31) / 32;
    std::vector<uint32_t> bits(std::max<size_t>(words, 1), 0u);
    std::vector<float4> a(std::max<uint64_t>(n, 1)), b(std::max<uint64_t>(n, 1));
    for (uint64_t i = 0; i < n; ++i) {
        memcpy(&a[i].x, &edits[i].flag, 4);
        a[i].y = edits[i].color[0]; a[i].z = edits[i].color[1]; a[i].w = edits[i].color[2];
        b[i] = make_float4(edits[i].contrast, edits[i].exposure, edits[i].gamma, edits[i].alpha);
        if (edits[i].flag & GSX_EDIT_ENABLED) bits[i >> 5] |= 1u << (i & 31);
    }
    HIPCHK(gsx::op::MemcpyAsync(m->edited.p, bits.data(), 4 * words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::MemcpyAsync(m->edit_a.p, a.data(), 16 * n, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::MemcpyAsync(m->edit_b.p, b.data(), 16 * n, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    m->has_edits = true;
    return GSX_OK;
}

gsx_status gsx_query_download_hits(gsx_viewer* v, const char* key, gsx_query_hit* out, uint64_t capacity, uint64_t* out_n) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !out_n) return fail(GSX_ERR_NOT_FOUND, "gsx_query_download_hits: no model '%s'", key ? key : "(null)");
    *out_n = 0;
    if (m->flags_kind != GSX_QUERY_HIT) return GSX_OK;
    uint32_t cnt = 0;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(&cnt, m->hit_count.p, 4, hipMemcpyDeviceToHost));
    cnt = std::min<uint32_t>(cnt, GSX_QUERY_MAX_HITS);
    std::vector<gsx_query_hit> h(cnt);
    if (cnt) HIPCHK(gsx::op::Memcpy(h.data(), m->hits.p, sizeof(gsx_query_hit) * cnt, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end(), [](const gsx_query_hit& x, const gsx_query_hit& y) {
        return x.depth != y.depth ? x.depth < y.depth : x.index < y.index;
    });
    *out_n = cnt;
    if (cnt > capacity) return fail(GSX_ERR_INVALID_ARG, "gsx_query_download_hits: %u hits exceed the capacity %llu", cnt, (unsigned long long)capacity);
    if (cnt && !out) return fail(GSX_ERR_INVALID_ARG, "gsx_query_download_hits: out is null");
    for (uint32_t i = 0; i < cnt; ++i) out[i] = h[i];
    return GSX_OK;
}

// world position on the pixel ray at view depth d: p_v = (ndc.x d / P00, ndc.y d / P11, -d), p_w = R^T (p_v - t)
static void unproject(const float view[16], const float proj[16], uint32_t w, uint32_t h, const float c[2], float d, float out[3]) {
    const float ndcx = 2.0f * c[0] / (float)w - 1.0f, ndcy = 1.0f - 2.0f * c[1] / (float)h;
    const float pv[3] = {ndcx * d / proj[0], ndcy * d / proj[5], -d};
    const float q[3] = {pv[0] - view[12], pv[1] - view[13], pv[2] - view[14]};
    for (int r = 0; r < 3; ++r) out[r] = (view[4 * r + 0] * q[0] + view[4 * r + 1] * q[1]) + view[4 * r + 2] * q[2];
}

gsx_status gsx_query_hit_pos_by_closest(const gsx_query_hit* hits, uint64_t n, const float view[16], const float proj[16],
                                        uint32_t width, uint32_t height, const float coords[2], uint32_t* out_index,
                                        float out_pos[3]) {
    if (!view || !proj || !coords || !out_pos || (n && !hits)) return fail(GSX_ERR_INVALID_ARG, "gsx_query_hit_pos_by_closest: null argument");
    if (n == 0) return fail(GSX_ERR_NOT_FOUND, "gsx_query_hit_pos_by_closest: no hit");
    uint64_t best = 0;
    for (uint64_t i = 1; i < n; ++i)
        if (hits[i].depth < hits[best].depth || (hits[i].depth == hits[best].depth && hits[i].index < hits[best].index)) best = i;
    if (out_index) *out_index = hits[best].index;
    unproject(view, proj, width, height, coords, hits[best].depth, out_pos);
    return GSX_OK;
}

gsx_status gsx_query_hit_pos_by_alpha_range(const gsx_query_hit* hits, uint64_t n, const float view[16], const float proj[16],
                                            uint32_t width, uint32_t height, const float coords[2], float range,
                                            uint32_t* out_index, float* out_alpha, float out_pos[3]) {
    if (!view || !proj || !coords || !out_pos || (n && !hits)) return fail(GSX_ERR_INVALID_ARG, "gsx_query_hit_pos_by_alpha_range: null argument");
    if (n == 0) return fail(GSX_ERR_NOT_FOUND, "gsx_query_hit_pos_by_alpha_range: no hit");
    float amax = hits[0].alpha;
    for (uint64_t i = 1; i < n; ++i) amax = std::max(amax, hits[i].alpha);
    uint64_t best = n;
    for (uint64_t i = 0; i < n; ++i) {
        if (hits[i].alpha < amax - range) continue;
        if (best == n || hits[i].depth < hits[best].depth || (hits[i].depth == hits[best].depth && hits[i].index < hits[best].index)) best = i;
    }
    if (out_index) *out_index = hits[best].index;
    if (out_alpha) *out_alpha = hits[best].alpha;
    unproject(view, proj, width, height, coords, hits[best].depth, out_pos);
    return GSX_OK;
}

}  // extern "C"
