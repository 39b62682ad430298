// gsx_api.cpp — host side of libgsx.so: the C ABI of include/gsx.h over the gfx950 kernels.
//
// Mirrors the call protocol the app drives every frame (src/tab/scene.rs:699-874 and 2263-2326):
//   update_* uniforms -> per model preprocess + radix sort -> submit/poll -> per model render far->near.
// One viewer = one HIP device + one stream; every buffer of a model lives in HBM for the model's
// lifetime (the reference's MultiModelViewerGaussianBuffers, scene.rs:2111-2112).
// There is NO CPU fallback: without a HIP device every entry point fails with GSX_ERR_NO_DEVICE.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gsx_internal.h"

namespace gsx {

static thread_local std::string g_err;

static gsx_status fail(gsx_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

gsx_status ply_fail(gsx_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(_e == hipErrorOutOfMemory ? GSX_ERR_OOM : GSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                   \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only; contents are NOT preserved
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        release();
        size_t want = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

using Counters = SlabStats;  // device copy + pinned host mirror

struct Model {
    std::string key;
    uint64_t n = 0;
    gsx_sh_kind sh_kind = GSX_SH_SINGLE;
    gsx_cov3d_kind cov_kind = GSX_COV3D_SINGLE;
    bool has_sh = true;
    bool has_mask = false;
    ModelTransform mt;
    FrameConsts fc{};

    DevBuf pc, cov_a, cov_b, sh4, sh1, sh_h, sh_q, sh_aos, cov_h, cov_h2, mask;
    DevBuf key_buf, rec_a, rec_b, rec_c;        // projection records of the model's own Gaussians
    DevBuf imp_key, imp_a, imp_b, imp_c;        // records imported from other ranks (kept apart: a frame may pack twice)
    bool use_imported = false;
    uint64_t sortbin_cap = 0, imp_cap = 0;
    DevBuf dp_a, dp_b, sk_out, sv_out, sort_ws; // depth sort: pair scratch, sorted keys / indices, workspace
    DevBuf cnt, block_sums, srect;              // per slab: tile counts in depth order, scan partials, tile rects
    DevBuf block_vis;                           // per-workgroup visible counts of the projection pass
    DevBuf tp_src, tp_a, tp_b, tk_out, tv_out, tsort_ws;  // tile pairs: emitted, scratch, sorted (split), workspace
    DevBuf ranges;
    DevBuf counters;
    Counters* h_counters = nullptr;             // pinned
    uint32_t* sorted_idx = nullptr;             // -> sv_a or sv_b after the depth sort
    uint32_t* tile_list = nullptr;              // -> tv_* after the tile sort
    uint32_t* tile_keys = nullptr;
    bool preprocessed = false, sorted = false, counters_valid = false, binned = false;
    bool stats_pending = false;                 // device statistics newer than the host mirror
    bool ranges_clean = false;                  // the tile range table is known to be all-zero
    bool lists_complete = false;                // the tile lists of the last render cover the whole model (one slab)
    uint32_t n_visible = 0, n_entries = 0, n_sorted = 0, n_sorted2 = 0;
    uint64_t tile_cap = 0;                      // capacity (entries) of the tile-pair buffers
    uint32_t slabs_hint = 0;                    // slabs the last observed frame needed (0 = unknown)
    hipEvent_t stats_event = nullptr;           // completion of the asynchronous statistics copy
    bool stats_copy_inflight = false;
    // the per-frame record set: the model's own projection (rec_n == n) or records imported from the
    // other ranks (gsx_shard_import); binning is restricted to the band of tile rows [row_lo, row_hi)
    uint64_t rec_n = 0, rec_cap = 0;
    uint32_t row_lo = 0, row_hi = 0xFFFFFFFFu;  // band of tile rows this viewer bins (clamped to tiles_y)
    DevBuf pack_table;
    // selection / edits / query (kernels_edit.hip); all allocated on first use
    DevBuf selection, edited, edit_a, edit_b, keep, query_flags, hits, hit_count;
    bool has_selection = false, has_edits = false, show_unedited = false;
    uint32_t flags_kind = GSX_QUERY_NONE, flags_op = GSX_SELECTION_SET;  // what the last preprocess evaluated
    // temporal occlusion speculation (kernels_spec.hip): this model's per-tile windows for its next frame, the repair
    // windows of the current one, the saturated-tile bitmap as it was before this model was composited
    DevBuf spec_win, spec_win2, spec_done_before, spec_need, spec_coarse, spec_coarse2;
    bool spec_valid = false, spec_round1 = false;
    bool order_consumed = false;   // a speculated render overwrote the depth order with its repair round's
    uint32_t spec_tiles_x = 0, spec_tiles_y = 0, shard_tiles_x = 0, shard_tiles_y = 0;
    // lazily projected shard (gsx_shard_set_windows): the windows of the coming exchange, their max-pyramid, and whether the
    // last preprocess left a candidate list in adm_pairs
    DevBuf shard_win, shard_pyr, trav_ballots, trav_counts;
    bool shard_win_set = false, cand_valid = false;
    DevBuf adm_ballots2;           // the repair round's ballots (the first round's stay: they say which records are shaded)
    bool lazy = false;             // this frame's projection shaded only the admitted Gaussians
    const uint32_t* last_pyramid = nullptr;  // the admission pyramid the projection pass used
    uint32_t* last_pod_mask = nullptr;  // the keep-bitset the projection pass used (mask, or mask & ~hidden)
    DevBuf adm_offsets, adm_counts2;  // scan output of adm_counts; counts of the admission passes that run outside the projection
    DevBuf adm_pairs, adm_ballots, adm_counts;  // admission pass: compacted (key, index) pairs, per-wave ballots, per-workgroup counts
    DevBuf pack_masks;             // destination bit mask per record (gsx_shard_pack)
    DevBuf window, pack_window;    // per-tile depth-key windows [lo, hi): of the imported set / of the pack in flight
    bool has_window = false;

    ~Model() {
        if (h_counters) (void)hipHostFree(h_counters);
        if (stats_event) (void)hipEventDestroy(stats_event);
    }
    PodPlanes pod() const {
        PodPlanes p;
        p.pc = pc.as<float4>();
        p.cov_a = cov_a.as<float4>();
        p.cov_b = cov_b.as<float2>();
        p.sh4 = sh4.as<float4>();
        p.sh1 = sh1.as<float>();
        p.sh_h = sh_h.as<uint4>();
        p.sh_q = sh_q.as<uint4>();
        p.sh_aos = sh_aos.as<uint4>();
        p.cov_h = cov_h.as<uint2>();
        p.cov_h2 = cov_h2.as<uint32_t>();
        p.sh_kind = (int)sh_kind;
        p.cov_kind = (int)cov_kind;
        p.mask = has_mask ? mask.as<uint32_t>() : nullptr;
        return p;
    }
    Records proj_rec() const {
        Records r;
        r.key = key_buf.as<uint32_t>();
        r.a = rec_a.as<float4>();
        r.b = rec_b.as<float4>();
        r.c = rec_c.as<float4>();
        return r;
    }
    Records imp_rec() const {
        Records r;
        r.key = imp_key.as<uint32_t>();
        r.a = imp_a.as<float4>();
        r.b = imp_b.as<float4>();
        r.c = imp_c.as<float4>();
        return r;
    }
    Records rec() const { return use_imported ? imp_rec() : proj_rec(); }  // the frame's active record set
};

struct PassTimer {
    hipEvent_t start, stop;
    int pass;
};

}  // namespace gsx

using namespace gsx;

struct gsx_viewer {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    gsx_spec_params params{};
    float view[16]{}, proj[16]{};
    uint32_t width = 1, height = 1;
    float size = 1.0f;
    uint32_t display_mode = GSX_DISPLAY_SPLAT, sh_deg = 3, no_sh0 = 0;
    std::map<std::string, std::unique_ptr<Model>> models;
    DevBuf fb, staging, scratch, done_bits;
    DevBuf frame_done;        // u32: tiles saturated so far in the current frame (all models)
    std::vector<std::string> last_keys;  // keys of the last gsx_render, for the overflow redo
    bool last_render_cont = false;
    uint32_t band_lo = 0, band_hi = 0xFFFFFFFFu;  // tile rows this viewer renders (gsx_viewer_set_band)
    gsx_query query{};                   // GSX_QUERY_NONE
    DevBuf query_texture;
    uint32_t query_tex_w = 0, query_tex_h = 0;
    float highlight[4]{0, 0, 0, 0};
    gsx_gaussian_edit sel_edit{0u, {0.0f, 1.0f, 1.0f}, 0.0f, 0.0f, 1.0f, 1.0f};
    void* ext_fb = nullptr;              // caller-owned framebuffer (multi-GPU: the RCCL gather target)
    uint64_t ext_fb_bytes = 0;
    gsx_render_options options{1u, 16u, 131072u, 2u, 1u, 0.25f, 3u};
    uint32_t timing = 0;  // bit p: bracket pass p with events
    std::vector<PassTimer> timers;     // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
    float pass_ms[GSX_PASS_COUNT]{};
    uint32_t pass_launches[GSX_PASS_COUNT]{};
};

namespace gsx {

static Model* find_model(gsx_viewer* v, const char* key) {
    if (!v || !key) return nullptr;
    auto it = v->models.find(key);
    return it == v->models.end() ? nullptr : it->second.get();
}

struct ScopedPass {
    gsx_viewer* v;
    int pass;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedPass(gsx_viewer* v_, int pass_) : v(v_), pass(pass_) {
        if (!((v->timing >> pass) & 1u)) return;
        if (!v->event_pool.empty()) {
            a = v->event_pool.back().first;
            b = v->event_pool.back().second;
            v->event_pool.pop_back();
        } else {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
        }
        (void)hipEventRecord(a, v->stream);
    }
    ~ScopedPass() {
        if (!a) return;
        (void)hipEventRecord(b, v->stream);
        v->timers.push_back({a, b, pass});
    }
};

static uint32_t ceil_log2(uint32_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

static gsx_status viewer_bind(gsx_viewer* v) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "viewer is null");
    HIPCHK(hipSetDevice(v->device));
    return GSX_OK;
}

static gsx_status ensure_fb(gsx_viewer* v) {
    if (v->ext_fb) {
        if (v->ext_fb_bytes < sizeof(float4) * (size_t)v->width * v->height)
            return fail(GSX_ERR_INVALID_ARG, "external framebuffer of %llu bytes is too small for %ux%u", (unsigned long long)v->ext_fb_bytes, v->width, v->height);
        return GSX_OK;
    }
    HIPCHK(v->fb.ensure(sizeof(float4) * (size_t)v->width * v->height));
    return GSX_OK;
}
static float4* fb_ptr(gsx_viewer* v) { return v->ext_fb ? static_cast<float4*>(v->ext_fb) : reinterpret_cast<float4*>(v->fb.p); }

static gsx_status do_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys, bool cont = false);
static gsx_status do_sort(gsx_viewer* v, Model* m, bool force_full = false);
static gsx_status complete_records(gsx_viewer* v, Model* m);

// Frames are enqueued without any host round trip; this is where the host catches up: wait for the
// stream, mirror the per-model statistics, and if a depth slab needed more tile-pair capacity than was
// allocated, grow the buffers and redo the last gsx_render (rare: capacity starts at 16 entries/record).
static gsx_status finish_frame(gsx_viewer* v) {
    for (int attempt = 0; attempt < 8; ++attempt) {
        bool pending = false;
        for (auto& kv : v->models) pending |= kv.second->stats_pending;
        if (!pending) return GSX_OK;
        for (auto& kv : v->models) {
            Model* m = kv.second.get();
            if (m->stats_pending)
                HIPCHK(hipMemcpyAsync(m->h_counters, m->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, v->stream));
        }
        HIPCHK(hipStreamSynchronize(v->stream));
        bool redo = false;
        for (auto& kv : v->models) {
            Model* m = kv.second.get();
            if (!m->stats_pending) continue;
            m->stats_pending = false;
            m->n_visible = m->h_counters->n_visible;
            m->n_sorted = m->h_counters->n_sorted;
            m->n_sorted2 = m->h_counters->n_sorted2;
            m->n_entries = m->h_counters->n_entries_total;
            m->counters_valid = true;
            if (m->binned) m->slabs_hint = m->h_counters->slabs_used;
            m->stats_copy_inflight = false;
            if (m->h_counters->overflow && m->binned) {
                m->tile_cap = std::max<uint64_t>(2 * m->tile_cap, (uint64_t)m->h_counters->max_needed + 1024);
                if (v->last_render_cont || m->rec_n != m->n)
                    return fail(GSX_ERR_OOM, "tile-pair capacity overflow in a sharded frame (model '%s'); capacity grown for the "
                                "next frame, this frame is incomplete", m->key.c_str());
                redo = true;
            }
        }
        if (!redo) return GSX_OK;
        std::vector<const char*> keys;
        for (auto& k : v->last_keys) keys.push_back(k.c_str());
        for (auto& k : v->last_keys) {  // a speculated frame is redone unspeculated: its depth order was consumed
            Model* m = find_model(v, k.c_str());
            if (m && m->spec_round1) {
                m->spec_valid = false;
                gsx_status st2 = do_sort(v, m, true);
                if (st2) return st2;
            }
        }
        gsx_status st = do_render(v, keys.data(), (uint32_t)keys.size());
        if (st) return st;
    }
    return fail(GSX_ERR_OOM, "tile-pair buffers kept overflowing");
}

static gsx_status sync_counters(gsx_viewer* v) { return finish_frame(v); }

// buffers sized by the model (projection outputs)
static gsx_status ensure_record_capacity(Model* m, uint64_t count) {
    if (count <= m->rec_cap) return GSX_OK;
    const size_t n = std::max<uint64_t>(count, 1);
    HIPCHK(m->key_buf.ensure(4 * n));
    HIPCHK(m->rec_a.ensure(16 * n));
    HIPCHK(m->rec_b.ensure(16 * n));
    HIPCHK(m->rec_c.ensure(16 * n));
    HIPCHK(m->block_vis.ensure(4 * (project_blocks(n) + 1)));
    m->rec_cap = n;
    return GSX_OK;
}

// buffers sized by the frame's active record set (depth sort + per-slab binning)
static gsx_status ensure_sortbin_capacity(Model* m, uint64_t count) {
    if (count <= m->sortbin_cap) return GSX_OK;
    const size_t n = std::max<uint64_t>(count + count / 8, 1);
    HIPCHK(m->dp_a.ensure(8 * n));
    HIPCHK(m->dp_b.ensure(8 * n));
    HIPCHK(m->sk_out.ensure(4 * n));
    HIPCHK(m->sv_out.ensure(4 * n));
    {
        const size_t ws = 4 * radix_workspace_words(n);
        if (ws > m->sort_ws.bytes) {
            HIPCHK(m->sort_ws.ensure(ws));
            HIPCHK(hipMemset(m->sort_ws.p, 0, m->sort_ws.bytes));  // status words must not alias a live epoch
        }
    }
    HIPCHK(m->cnt.ensure(4 * n));
    HIPCHK(m->srect.ensure(8 * n));
    HIPCHK(m->block_sums.ensure(4 * (scan_blocks(n) + 1)));
    m->sortbin_cap = n;
    return GSX_OK;
}

static gsx_status ensure_import_capacity(Model* m, uint64_t count) {
    if (count > m->imp_cap) {
        const size_t n = std::max<uint64_t>(count + count / 8, 1);
        HIPCHK(m->imp_key.ensure(4 * n));
        HIPCHK(m->imp_a.ensure(16 * n));
        HIPCHK(m->imp_b.ensure(16 * n));
        HIPCHK(m->imp_c.ensure(16 * n));
        m->imp_cap = n;
    }
    return ensure_sortbin_capacity(m, count);
}

static gsx_status ensure_selection(gsx_viewer* v, Model* m) {
    const size_t bytes = 4 * std::max<size_t>(((size_t)m->n + 31) / 32, 1);
    if (m->selection.bytes < bytes) {
        HIPCHK(m->selection.ensure(bytes));
        HIPCHK(hipMemsetAsync(m->selection.p, 0, bytes, v->stream));
    }
    return GSX_OK;
}

static gsx_status ensure_edit_buffers(gsx_viewer* v, Model* m) {
    const size_t words = std::max<size_t>(((size_t)m->n + 31) / 32, 1), n = std::max<size_t>(m->n, 1);
    if (m->edited.bytes < 4 * words) {
        HIPCHK(m->edited.ensure(4 * words));
        HIPCHK(hipMemsetAsync(m->edited.p, 0, 4 * words, v->stream));
        HIPCHK(m->keep.ensure(4 * words));
        HIPCHK(m->edit_a.ensure(16 * n));
        HIPCHK(m->edit_b.ensure(16 * n));
    }
    return GSX_OK;
}

static gsx_status do_preprocess(gsx_viewer* v, Model* m) {
    frame_consts_setup(v->view, v->proj, v->width, v->height, m->mt, v->size, v->display_mode, v->sh_deg, v->no_sh0,
                       v->params, &m->fc);
    m->fc.band_lo = std::min(v->band_lo, m->fc.tiles_y);
    m->fc.band_hi = std::min(v->band_hi, m->fc.tiles_y);
    m->preprocessed = m->sorted = m->counters_valid = m->binned = false;
    m->order_consumed = false;
    gsx_status st = ensure_record_capacity(m, m->n);
    if (st) return st;
    if ((st = ensure_sortbin_capacity(m, m->n))) return st;
    m->use_imported = false;
    m->rec_n = m->n;
    m->row_lo = m->fc.band_lo;
    m->row_hi = m->fc.band_hi;
    // selection edit / stored edits / highlight: only when something of the kind exists (spec §7)
    const uint32_t n32 = (uint32_t)m->n;
    const size_t words = ((size_t)m->n + 31) / 32;
    const bool sel_edit_on = m->has_selection && (v->sel_edit.flag & GSX_EDIT_ENABLED);
    const bool edits_on = !m->show_unedited && (m->has_edits || sel_edit_on);
    const bool highlight_on = m->has_selection && v->highlight[3] > 0.0f;
    PodPlanes pod = m->pod();
    if (edits_on) {
        if ((st = ensure_edit_buffers(v, m))) return st;
        HIPCHK(launch_edit_prepare(v->stream, n32, m->has_selection ? m->selection.as<uint32_t>() : nullptr,
                                   m->edited.as<uint32_t>(), m->edit_a.as<float4>(), m->edit_b.as<float4>(), v->sel_edit, pod.mask,
                                   m->keep.as<uint32_t>()));
        m->has_edits = true;
        pod.mask = m->keep.as<uint32_t>();
    }
    // admission is decided inside the projection kernel: every visible Gaussian, or — when this model has windows from
    // its previous frame — the conservative max-pyramid test of the temporal occlusion speculation
    m->spec_round1 = v->options.progressive && v->options.speculative && m->spec_valid && m->spec_tiles_x == m->fc.tiles_x &&
                     m->spec_tiles_y == m->fc.tiles_y;
    ProjectAdmission adm{};
    HIPCHK(m->adm_ballots.ensure(8 * ((std::max<size_t>(m->n, 1) + 63) / 64 + 4)));
    HIPCHK(m->adm_counts.ensure(4 * (std::max<size_t>(std::max(admit_blocks(m->n), (size_t)(m->n + 255) / 256), 1) + 4)));
    const bool shard_lazy = m->shard_win_set && m->shard_tiles_x == m->fc.tiles_x && m->shard_tiles_y == m->fc.tiles_y;
    if (shard_lazy) m->spec_round1 = false;  // a sharded frame: the windows come from the caller, not from this viewer's last frame
    if (m->spec_round1) adm.pyramid = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse.as<uint32_t>());
    if (shard_lazy) adm.pyramid = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->shard_pyr.as<uint32_t>());
    adm.ballots = m->adm_ballots.as<unsigned long long>();
    adm.block_counts = m->adm_counts.as<uint32_t>();
    // lazy shading: nothing else reads the conic / colour records of this frame (no edit, highlight or query pass)
    m->lazy = (m->spec_round1 || shard_lazy) && !edits_on && !highlight_on && v->query.kind == GSX_QUERY_NONE;
    m->cand_valid = false;
    adm.lazy = m->lazy ? 1u : 0u;
    m->last_pod_mask = pod.mask;
    m->last_pyramid = adm.pyramid.data;
    {
        ScopedPass t(v, GSX_PASS_PROJECT);  // brackets the projection kernel alone (bench.py's roofline kernel)
        HIPCHK(launch_project(v->stream, m->fc, n32, pod, m->proj_rec(), m->block_vis.as<uint32_t>(), adm));
        v->pass_launches[GSX_PASS_PROJECT] += m->n ? 1 : 0;
    }
    HIPCHK(launch_sum_counts(v->stream, m->block_vis.as<uint32_t>(), n32, &m->counters.as<Counters>()->n_visible));
    if (shard_lazy) {
        // the candidates of the coming exchange (a conservative superset of the travellers): compact them and give
        // exactly those their conic / colour records; gsx_shard_pack then looks at nothing else
        Counters* dcx = m->counters.as<Counters>();
        HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(m->n, 1)));
        HIPCHK(m->adm_offsets.ensure(m->adm_counts.bytes));
        HIPCHK(launch_admit_from_project(v->stream, m->proj_rec().key, n32, m->adm_ballots.as<unsigned long long>(),
                                         m->adm_counts.as<uint32_t>(), m->adm_offsets.as<uint32_t>(), &dcx->n_candidates,
                                         m->adm_pairs.as<uint2>()));
        if (m->lazy) HIPCHK(launch_shade(v->stream, m->fc, n32, pod, m->proj_rec(), LateProjection{m->adm_pairs.as<uint2>(), &dcx->n_candidates, nullptr}));
        m->cand_valid = true;
    }
    if (edits_on || highlight_on)
        HIPCHK(launch_edit_apply(v->stream, n32, m->proj_rec(), highlight_on ? m->selection.as<uint32_t>() : nullptr,
                                 edits_on ? m->edited.as<uint32_t>() : nullptr, m->edit_a.as<float4>(), m->edit_b.as<float4>(),
                                 v->highlight));
    m->flags_kind = GSX_QUERY_NONE;
    if (v->query.kind != GSX_QUERY_NONE) {
        if (v->query.kind == GSX_QUERY_HIT) {
            HIPCHK(m->hits.ensure(sizeof(gsx_query_hit) * (size_t)GSX_QUERY_MAX_HITS));
            HIPCHK(m->hit_count.ensure(4));
            HIPCHK(hipMemsetAsync(m->hit_count.p, 0, 4, v->stream));
        } else {
            HIPCHK(m->query_flags.ensure(4 * std::max<size_t>(words, 1)));
            if (v->query.kind == GSX_QUERY_TEXTURE && (v->query_tex_w != v->width || v->query_tex_h != v->height))
                return fail(GSX_ERR_INVALID_ARG, "gsx_preprocess: texture query without a viewport-sized query texture (gsx_update_query_texture)");
        }
        HIPCHK(launch_query(v->stream, n32, m->proj_rec(), v->query, v->query_texture.as<uint8_t>(), v->query_tex_w, v->query_tex_h,
                            m->fc, m->query_flags.as<uint32_t>(), m->hits.as<gsx_query_hit>(), m->hit_count.as<uint32_t>(),
                            GSX_QUERY_MAX_HITS));
        m->flags_kind = v->query.kind;
        m->flags_op = v->query.selection_op;
    }
    m->stats_pending = true;
    m->preprocessed = true;
    return GSX_OK;
}

// A lazily shaded frame left the conic / colour records of the refused Gaussians unwritten; whoever needs all of them
// (parity download, multi-GPU pack, a redone frame) gets them by running the projection again, unlazily: same values.
static gsx_status complete_records(gsx_viewer* v, Model* m) {
    if (!m->lazy || !m->preprocessed) return GSX_OK;
    ProjectAdmission adm{};
    adm.pyramid = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->last_pyramid);
    adm.ballots = m->adm_ballots.as<unsigned long long>();
    HIPCHK(m->block_sums.ensure(4 * std::max<size_t>((m->n + 255) / 256, 1)));
    adm.block_counts = m->block_sums.as<uint32_t>();  // scratch: the admission counts were consumed by the compaction
    PodPlanes pod = m->pod();
    pod.mask = m->last_pod_mask;
    HIPCHK(launch_project(v->stream, m->fc, (uint32_t)m->n, pod, m->proj_rec(), m->block_vis.as<uint32_t>(), adm));
    m->lazy = false;
    return GSX_OK;
}

// force_full: ignore the admission the projection pass made (a speculated frame being redone) and sort every visible record
static gsx_status do_sort(gsx_viewer* v, Model* m, bool force_full) {
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_sort('%s') before gsx_preprocess", m->key.c_str());
    const uint32_t n = (uint32_t)m->rec_n;
    Counters* dc = m->counters.as<Counters>();
    {
        ScopedPass t(v, GSX_PASS_DEPTH_SORT);
        if (m->use_imported) {  // every imported record is visible: sort the keys as they lie
            m->spec_round1 = false;
            RadixBuffers rb{m->rec().key, nullptr, nullptr, m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                            m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
            HIPCHK(launch_radix_sort(v->stream, rb, n, nullptr, 32, true));
        } else {
            // compact the (key, index) pairs the projection pass admitted, then sort only those
            HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(n, 1)));
            if (force_full) {
                gsx_status stc = complete_records(v, m);
                if (stc) return stc;
                m->spec_round1 = false;
                HIPCHK(m->adm_ballots2.ensure(8 * ((std::max<size_t>(n, 1) + 63) / 64)));
                HIPCHK(m->adm_counts2.ensure(4 * (std::max<size_t>(admit_blocks(n), 1) + 4)));
                HIPCHK(launch_admit(v->stream, m->proj_rec(), n, nullptr, m->fc.tiles_x, nullptr, 0, WindowPyramid{}, nullptr,
                                    m->adm_ballots2.as<unsigned long long>(), m->adm_counts2.as<uint32_t>(), &dc->n_sorted,
                                    m->adm_pairs.as<uint2>()));
            } else {
                HIPCHK(m->adm_offsets.ensure(m->adm_counts.bytes));
                HIPCHK(launch_admit_from_project(v->stream, m->proj_rec().key, n, m->adm_ballots.as<unsigned long long>(),
                                                 m->adm_counts.as<uint32_t>(), m->adm_offsets.as<uint32_t>(), &dc->n_sorted,
                                                 m->adm_pairs.as<uint2>()));
                if (m->lazy) {  // the projection pass was geometry only: shade what it admitted
                    PodPlanes pod = m->pod();
                    pod.mask = m->last_pod_mask;
                    HIPCHK(launch_shade(v->stream, m->fc, n, pod, m->proj_rec(), LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted, nullptr}));
                }
            }
            RadixBuffers rb{nullptr, nullptr, m->adm_pairs.as<uint2>(), m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                            m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
            HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted, 32, false));
        }
        m->sorted_idx = m->sv_out.as<uint32_t>();
        v->pass_launches[GSX_PASS_DEPTH_SORT] += n ? 4 : 0;
    }
    m->stats_pending = true;
    m->sorted = true;
    if (force_full) m->order_consumed = false;  // an unspeculated order over every visible record: renderable again
    m->binned = false;
    m->n_entries = 0;
    return GSX_OK;
}

// Depth slabs of the progressive mode: [0, n/div), then each slab `growth` times the previous one.
static void plan_slabs(const gsx_render_options& o, uint32_t n_vis, std::vector<uint32_t>* bounds) {
    bounds->clear();
    bounds->push_back(0);
    if (!o.progressive || n_vis <= o.min_slab) {
        bounds->push_back(n_vis);
        return;
    }
    uint64_t size = std::max<uint64_t>(o.min_slab, n_vis / std::max(1u, o.first_slab_divisor));
    uint64_t at = 0;
    while (at + size < n_vis) {
        at += size;
        bounds->push_back((uint32_t)at);
        size *= std::max(2u, o.growth);
    }
    bounds->push_back(n_vis);
}

// Frames are coherent: if the last observed frame saturated every tile after `used` slabs, the slabs
// after used + 1 are merged into ONE remainder slab.  When the prediction holds that slab falls through
// on the device (its count pass sees every tile done); when it does not, the remainder slab simply does
// the work — the image is the same either way, only the number of empty launches changes.
static void merge_tail_slabs(std::vector<uint32_t>* bounds, uint32_t used) {
    if (used == 0) return;
    const size_t keep = (size_t)used + 1;  // slabs kept as planned
    if (bounds->size() > keep + 2) {
        const uint32_t last = bounds->back();
        bounds->resize(keep + 1);
        bounds->push_back(last);
    }
}

// One model: bin + tile-sort + composite, front to back in depth slabs, enqueued without host syncs.
// Slab bounds are planned on the record count (an upper bound of N_vis; kernels clamp to the device-side
// N_vis), slab entry counts stay on the device, and once every tile this rank owns is saturated the
// remaining slabs' kernels fall through.  carry: the framebuffer already holds nearer models.
static gsx_status do_bin_and_composite(gsx_viewer* v, Model* m, bool carry) {
    if (!m->sorted) return fail(GSX_ERR_INVALID_ARG, "gsx_render: model '%s' was not preprocessed+sorted", m->key.c_str());
    if (m->order_consumed)
        return fail(GSX_ERR_INVALID_ARG, "gsx_render: the depth order of '%s' was consumed by a speculated frame's repair round; "
                    "call gsx_preprocess + gsx_sort('%s') again before rendering it once more (its admission belongs to the windows "
                    "that frame replaced)", m->key.c_str(), m->key.c_str());
    if (m->fc.w_px != v->width || m->fc.h_px != v->height)
        return fail(GSX_ERR_INVALID_ARG, "gsx_render: viewport changed since gsx_preprocess('%s')", m->key.c_str());
    const uint32_t n_tiles = m->fc.tiles_x * m->fc.tiles_y;
    const uint32_t row_words = (m->fc.tiles_x + 31) / 32;
    const bool progressive = v->options.progressive != 0;
    uint32_t* done = progressive ? v->done_bits.as<uint32_t>() + 1 : nullptr;  // word 0 is the saturated-tile counter
    uint32_t* done_count = v->done_bits.as<uint32_t>();
    const bool speculate = progressive && v->options.speculative && !m->use_imported;
    if (progressive && m->stats_copy_inflight && hipEventQuery(m->stats_event) == hipSuccess) {
        m->stats_copy_inflight = false;
        m->slabs_hint = m->h_counters->slabs_used;
        m->n_sorted = m->h_counters->n_sorted;
    }
    std::vector<uint32_t> bounds;
    if (m->spec_round1) {
        // a speculated round is ONE slab: the windows already bound what every tile takes to little more than it needs,
        // and the compositor stops a saturated tile by itself; more slabs only add launches (measured on cfg4: 551 fps
        // with one slab, 487 with three).  The kernels stride over what exists on the device, so the bound is free.
        bounds = {0u, (uint32_t)m->rec_n};
    } else {
        plan_slabs(v->options, (uint32_t)m->rec_n, &bounds);
        if (progressive) merge_tail_slabs(&bounds, m->slabs_hint);
    }
    Counters* dc = m->counters.as<Counters>();
    const uint32_t row_lo = std::min(m->row_lo, m->fc.tiles_y), row_hi = std::min(m->row_hi, m->fc.tiles_y);
    const uint32_t owned_tiles = (row_hi > row_lo ? row_hi - row_lo : 0) * m->fc.tiles_x;
    const uint2* window = (m->use_imported && m->has_window) ? m->window.as<uint2>() : nullptr;
    if (m->spec_round1) window = m->spec_win.as<uint2>();
    uint32_t* tile_sat = progressive ? done + row_words * m->fc.tiles_y : nullptr;  // [count | bitmap | saturation keys]

    if (m->tile_cap == 0) m->tile_cap = std::max<uint64_t>(1u << 20, 16 * m->rec_n);
    m->tile_cap = std::min<uint64_t>(m->tile_cap, 0xFFFFF000ull);
    const uint32_t cap = (uint32_t)m->tile_cap;
    {
        const size_t bytes = sizeof(uint32_t) * (size_t)cap;
        HIPCHK(m->tp_src.ensure(2 * bytes));
        HIPCHK(m->tk_out.ensure(bytes));
        HIPCHK(m->tv_out.ensure(bytes));
        HIPCHK(m->tp_a.ensure(2 * bytes));
        HIPCHK(m->tp_b.ensure(2 * bytes));
        const size_t ws = 4 * radix_workspace_words(cap);
        if (ws > m->tsort_ws.bytes) {
            HIPCHK(m->tsort_ws.ensure(ws));
            HIPCHK(hipMemsetAsync(m->tsort_ws.p, 0, m->tsort_ws.bytes, v->stream));
        }
        if (sizeof(uint2) * (size_t)n_tiles > m->ranges.bytes) m->ranges_clean = false;
        HIPCHK(m->ranges.ensure(sizeof(uint2) * (size_t)n_tiles));
    }
    // reset this model's per-frame totals (n_visible and n_sorted stay)
    HIPCHK(launch_zero_words(v->stream, &dc->n_entries, (uint32_t)((sizeof(Counters) - offsetof(Counters, n_entries)) / 4), nullptr, 0));
    const uint32_t* done_before = nullptr;
    if (speculate) {
        const size_t bm = 4 * (size_t)row_words * m->fc.tiles_y;
        HIPCHK(m->spec_win.ensure(sizeof(uint2) * (size_t)n_tiles));
        HIPCHK(m->spec_win2.ensure(sizeof(uint2) * (size_t)n_tiles));
        if (carry) {  // nearer models already saturated some tiles: remember which, they say nothing about this model
            HIPCHK(m->spec_done_before.ensure(bm));
            HIPCHK(hipMemcpyAsync(m->spec_done_before.p, done, bm, hipMemcpyDeviceToDevice, v->stream));
            done_before = m->spec_done_before.as<uint32_t>();
        }
    }
    const int bits = std::max<int>(1, (int)ceil_log2(n_tiles));
    // a single-slab front model keeps its complete tile lists for gsx_model_download_tile_lists
    const bool clear_ranges = progressive && !(bounds.size() == 2 && !carry && !m->spec_round1);
    // one depth slab [j0, j1) of the current depth order: bin -> tile sort -> ranges -> composite
    auto run_slab = [&](uint32_t j0, uint32_t j1, bool later, const uint2* win, const uint32_t* d_n, uint32_t slab_index) -> gsx_status {
        // the very first slab of the frame sees no saturated tile: plain rectangle areas
        const uint32_t* done_in = later ? done : nullptr;
        // a slab of S splats can produce at most S * n_tiles entries; size the sort launch by the smaller bound
        const uint32_t slab_cap = (uint32_t)std::min<uint64_t>(cap, (uint64_t)(j1 - j0) * std::min<uint64_t>(owned_tiles, 1u << 16));
        {
            ScopedPass t(v, GSX_PASS_BIN);
            HIPCHK(launch_tile_counts(v->stream, j0, j1, d_n, m->sorted_idx, m->rec(), m->srect.as<uint2>(),
                                      m->cnt.as<uint32_t>(), m->block_sums.as<uint32_t>(), dc, cap, row_lo, row_hi, done_in,
                                      row_words, (progressive && later) ? done_count : nullptr, owned_tiles, slab_index,
                                      win, m->sk_out.as<uint32_t>(), m->fc.tiles_x));
            HIPCHK(launch_tile_emit(v->stream, j0, j1, m->sorted_idx, m->srect.as<uint2>(), m->cnt.as<uint32_t>(),
                                    m->block_sums.as<uint32_t>(), m->fc.tiles_x, m->tp_src.as<uint2>(), row_lo, row_hi,
                                    done_in, row_words, d_n, &dc->n_entries, cap, win, m->sk_out.as<uint32_t>()));
            v->pass_launches[GSX_PASS_BIN] += 1;
        }
        {
            ScopedPass t(v, GSX_PASS_TILE_SORT);
            RadixBuffers rb{nullptr, nullptr, m->tp_src.as<uint2>(), m->tk_out.as<uint32_t>(), m->tv_out.as<uint32_t>(),
                            m->tp_a.as<uint2>(), m->tp_b.as<uint2>(), m->tsort_ws.as<uint32_t>()};
            HIPCHK(launch_radix_sort(v->stream, rb, slab_cap, &dc->n_entries, bits, false));
            m->tile_keys = m->tk_out.as<uint32_t>();
            m->tile_list = m->tv_out.as<uint32_t>();
            v->pass_launches[GSX_PASS_TILE_SORT] += (bits + 7) / 8;
        }
        {
            ScopedPass t(v, GSX_PASS_BIN);
            HIPCHK(launch_tile_ranges(v->stream, slab_cap, &dc->n_entries, m->tile_keys, n_tiles, m->ranges.as<uint2>(),
                                      m->ranges_clean));
        }
        {
            ScopedPass t(v, GSX_PASS_COMPOSITE);
            HIPCHK(launch_composite(v->stream, m->fc, m->ranges.as<uint2>(), m->tile_list, m->rec(), fb_ptr(v),
                                    later, done, row_words, done_count, clear_ranges, tile_sat));
            m->ranges_clean = clear_ranges;  // the compositor zeroed every range it consumed
            v->pass_launches[GSX_PASS_COMPOSITE] += 1;
        }
        return GSX_OK;
    };
    gsx_status st = GSX_OK;
    for (size_t sl = 0; sl + 1 < bounds.size(); ++sl)
        if ((st = run_slab(bounds[sl], bounds[sl + 1], carry || sl > 0, window, &dc->n_sorted, (uint32_t)sl))) return st;
    if (m->spec_round1) {
        // verification on the device: tiles with a bounded window that are still open get, in one more round, exactly
        // the records they were refused, composited behind what they hold.  Nothing to repair: the kernels fall through.
        const uint32_t n = (uint32_t)m->rec_n;
        {
            ScopedPass t(v, GSX_PASS_DEPTH_SORT);
            HIPCHK(m->spec_need.ensure(4 * (size_t)row_words * m->fc.tiles_y));
            HIPCHK(launch_spec_verify(v->stream, m->spec_win.as<uint2>(), done, row_words, m->fc.tiles_x, m->fc.tiles_y,
                                      m->spec_win2.as<uint2>(), m->spec_need.as<uint32_t>(), &dc->spec_need, row_lo, row_hi));
            HIPCHK(m->adm_ballots2.ensure(8 * ((std::max<size_t>(n, 1) + 63) / 64)));
            HIPCHK(m->adm_counts2.ensure(4 * (std::max<size_t>(admit_blocks(n), 1) + 4)));
            // conservative admission against the min-pyramid of the repair windows' starts (four loads per record; the
            // binning applies the exact windows): an exact per-tile scan of every visible record cost 260-350 us here
            HIPCHK(m->spec_coarse2.ensure(4 * window_pyramid_words(m->fc.tiles_x, m->fc.tiles_y)));
            HIPCHK(launch_window_pyramid(v->stream, m->spec_win2.as<uint2>(), m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse2.as<uint32_t>(), true, &dc->spec_need));
            WindowPyramid pyr2 = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse2.as<uint32_t>());
            pyr2.min_of_starts = 1;
            HIPCHK(launch_admit(v->stream, m->proj_rec(), n, nullptr, m->fc.tiles_x, nullptr,
                                row_words, pyr2, &dc->spec_need,
                                m->adm_ballots2.as<unsigned long long>(), m->adm_counts2.as<uint32_t>(), &dc->n_sorted2,
                                m->adm_pairs.as<uint2>()));
            if (m->lazy) {  // the repair round needs records the lazy projection did not shade
                PodPlanes pod = m->pod();
                pod.mask = m->last_pod_mask;
                HIPCHK(launch_shade(v->stream, m->fc, n, pod, m->proj_rec(),
                                    LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted2, m->adm_ballots.as<unsigned long long>()}));
            }
            RadixBuffers rb{nullptr, nullptr, m->adm_pairs.as<uint2>(), m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                            m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
            HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted2, 32, false));
        }
        if ((st = run_slab(0, n, true, m->spec_win2.as<uint2>(), &dc->n_sorted2, (uint32_t)bounds.size()))) return st;
        m->order_consumed = true;
    }
    if (speculate) {  // this model's windows for its next frame
        ScopedPass t(v, GSX_PASS_COMPOSITE);
        HIPCHK(launch_spec_next(v->stream, tile_sat, done, done_before, row_words, m->fc.tiles_x, m->fc.tiles_y,
                                v->options.spec_margin, v->options.spec_radius, m->spec_win.as<uint2>(), row_lo, row_hi));
        HIPCHK(m->spec_coarse.ensure(4 * window_pyramid_words(m->fc.tiles_x, m->fc.tiles_y)));
        HIPCHK(launch_window_pyramid(v->stream, m->spec_win.as<uint2>(), m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse.as<uint32_t>()));
        m->spec_valid = true;
        m->spec_tiles_x = m->fc.tiles_x;
        m->spec_tiles_y = m->fc.tiles_y;
    }
    if (progressive && !m->stats_copy_inflight) {  // feed the next frames' slab plan without waiting
        if (!m->stats_event) HIPCHK(hipEventCreateWithFlags(&m->stats_event, hipEventDisableTiming));
        HIPCHK(hipMemcpyAsync(m->h_counters, m->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, v->stream));
        HIPCHK(hipEventRecord(m->stats_event, v->stream));
        m->stats_copy_inflight = true;
    }
    m->binned = true;
    m->stats_pending = true;
    m->lists_complete = bounds.size() == 2 && !carry && !m->spec_round1;
    return GSX_OK;
}

// cont: a second round of the same frame (multi-GPU back set): keep the framebuffer, the saturated-tile state
// and carry (C, T) into the first model.
static gsx_status do_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys, bool cont) {
    gsx_status st = ensure_fb(v);
    if (st) return st;
    std::vector<Model*> order;
    std::vector<std::string> key_copy;
    for (uint32_t i = 0; i < n_keys; ++i) {
        Model* m = find_model(v, keys ? keys[i] : nullptr);
        if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_render: no model '%s'", keys && keys[i] ? keys[i] : "(null)");
        order.push_back(m);
        key_copy.push_back(keys[i]);
    }
    v->last_keys = key_copy;
    v->last_render_cont = cont;
    if (order.empty()) {
        if (cont) return GSX_OK;
        HIPCHK(launch_clear_fb(v->stream, fb_ptr(v), v->width * v->height));
        return GSX_OK;
    }
    if (!cont) {   // one memset: [saturated-tile counter | saturated-tile bitmap | per-tile saturation depth keys]
        const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
        const uint32_t row_words = (tiles_x + 31) / 32;
        const size_t bytes = 4 * (1 + (size_t)tiles_y * row_words + (size_t)tiles_y * tiles_x);
        HIPCHK(v->done_bits.ensure(bytes));
        HIPCHK(launch_zero_words(v->stream, v->done_bits.as<uint32_t>(), (uint32_t)(bytes / 4), nullptr, 0));
    }
    // the reference paints far -> near with "over"; front-to-back accumulation walks the same list backwards
    bool carry = cont;
    for (auto it = order.rbegin(); it != order.rend(); ++it) {
        if ((st = do_bin_and_composite(v, *it, carry))) return st;
        carry = true;
    }
    return GSX_OK;
}

}  // namespace gsx

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* gsx_last_error_string(void) { return g_err.c_str(); }
uint32_t gsx_abi_version(void) { return GSX_ABI_VERSION; }

void gsx_spec_params_default(gsx_spec_params* p) {
    if (!p) return;
    p->max_std_dev = 3.0f;
    p->cull_margin = 1.3f;
    p->jacobian_clamp = 1.3f;
    p->low_pass = 0.3f;
    p->alpha_max = 1.0f;
    p->alpha_min = 0.0f;
    p->t_epsilon = 1e-4f;
    p->point_radius = 2.0f;
}

gsx_status gsx_viewer_create(const gsx_viewer_desc* desc, gsx_viewer** out) {
    if (!desc || !out) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_create: null argument");
    if (desc->abi_version != GSX_ABI_VERSION)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_create: ABI version %u, library is %u", desc->abi_version, GSX_ABI_VERSION);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(GSX_ERR_NO_DEVICE, "gsx_viewer_create: no HIP device (%s); libgsx has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (desc->device < 0 || desc->device >= count)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_create: device %d out of range [0,%d)", desc->device, count);
    HIPCHK(hipSetDevice(desc->device));
    std::unique_ptr<gsx_viewer> v(new gsx_viewer());
    v->device = desc->device;
    if (desc->stream) {
        v->stream = reinterpret_cast<hipStream_t>(desc->stream);
    } else {
        HIPCHK(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
        v->own_stream = true;
    }
    gsx_spec_params_default(&v->params);
    v->width = std::max(1u, desc->width);
    v->height = std::max(1u, desc->height);
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(v->view, ident, sizeof ident);
    memcpy(v->proj, ident, sizeof ident);
    *out = v.release();
    return GSX_OK;
}

void gsx_viewer_destroy(gsx_viewer* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    (void)hipStreamSynchronize(v->stream);
    for (auto& t : v->timers) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto& p : v->event_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    v->models.clear();
    if (v->own_stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

void gsx_render_options_default(gsx_render_options* o) {
    if (!o) return;
    o->progressive = 1;
    o->first_slab_divisor = 16;
    o->min_slab = 131072;
    o->growth = 2;
    o->speculative = 1;
    o->spec_margin = 0.25f;
    o->spec_radius = 3;
}

gsx_status gsx_viewer_set_render_options(gsx_viewer* v, const gsx_render_options* o) {
    if (!v || !o) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: null argument");
    if (o->first_slab_divisor == 0 || o->growth < 2 || o->min_slab == 0)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: first_slab_divisor >= 1, growth >= 2, min_slab >= 1");
    if (!(o->spec_margin >= 0.0f) || o->spec_radius > 16)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: spec_margin >= 0, spec_radius <= 16");
    v->options = *o;
    return GSX_OK;
}

gsx_status gsx_viewer_set_spec_params(gsx_viewer* v, const gsx_spec_params* p) {
    if (!v || !p) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_spec_params: null argument");
    if (!(p->max_std_dev > 0.0f) || !(p->cull_margin > 0.0f) || !(p->alpha_max > 0.0f))
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_spec_params: max_std_dev, cull_margin, alpha_max must be > 0");
    v->params = *p;
    return GSX_OK;
}

gsx_status gsx_model_create(gsx_viewer* v, const char* key, uint64_t count, gsx_sh_kind sh, gsx_cov3d_kind cov3d) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!key) return fail(GSX_ERR_INVALID_ARG, "gsx_model_create: key is null");
    if (count >= 0xFFFFFFF0ull) return fail(GSX_ERR_INVALID_ARG, "gsx_model_create: count %llu too large", (unsigned long long)count);
    if (v->models.count(key)) return fail(GSX_ERR_INVALID_ARG, "gsx_model_create: model '%s' exists", key);
    if ((int)sh < 0 || (int)sh > GSX_SH_NONE || (int)cov3d < 0 || (int)cov3d > GSX_COV3D_HALF)
        return fail(GSX_ERR_INVALID_ARG, "gsx_model_create: unknown pod kind Sh%d/Cov3d%d", (int)sh, (int)cov3d);
    std::unique_ptr<Model> m(new Model());
    m->key = key;
    m->n = count;
    m->sh_kind = sh;
    m->cov_kind = cov3d;
    m->has_sh = sh != GSX_SH_NONE;
    const size_t n = std::max<uint64_t>(count, 1);
    HIPCHK(m->pc.ensure(16 * n));
    if (cov3d == GSX_COV3D_SINGLE) {
        HIPCHK(m->cov_a.ensure(16 * n));
        HIPCHK(m->cov_b.ensure(8 * n));
    } else {
        HIPCHK(m->cov_h.ensure(8 * n));
        HIPCHK(m->cov_h2.ensure(4 * n));
    }
    // SH twice: streaming planes for frames that shade every survivor, a per-Gaussian record copy for the sparse shading
    // of speculated frames (288 GB of HBM: 1.9 GB more at 10 M Gaussians buys whole-line gathers)
    if (sh == GSX_SH_SINGLE) {
        HIPCHK(m->sh4.ensure(16 * n * kShPlanes4));
        HIPCHK(m->sh1.ensure(4 * n));
        HIPCHK(m->sh_aos.ensure(16 * n * 12));
    } else if (sh == GSX_SH_HALF) {
        HIPCHK(m->sh_h.ensure(16 * n * 6));
        HIPCHK(m->sh_aos.ensure(16 * n * 6));
    } else if (sh == GSX_SH_NORM8) {
        HIPCHK(m->sh_q.ensure(16 * n * 3));
        HIPCHK(m->sh_aos.ensure(16 * n * 3));
    }
    HIPCHK(m->mask.ensure(4 * ((n + 31) / 32)));
    {
        gsx_status rst = ensure_record_capacity(m.get(), n);
        if (rst) return rst;
        if ((rst = ensure_sortbin_capacity(m.get(), n))) return rst;
    }
    HIPCHK(m->counters.ensure(sizeof(Counters)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&m->h_counters), sizeof(Counters), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(m->counters.p, 0, sizeof(Counters), v->stream));
    // a fresh model is all-zero Gaussians (new_empty) and fully unmasked (MaskOpTree::Reset, scene.rs:2124-2131)
    for (DevBuf* b : {&m->pc, &m->cov_a, &m->cov_b, &m->cov_h, &m->cov_h2, &m->sh4, &m->sh1, &m->sh_h, &m->sh_q})
        if (b->p) HIPCHK(hipMemsetAsync(b->p, 0, b->bytes, v->stream));
    v->models[key] = std::move(m);
    return GSX_OK;
}

gsx_status gsx_model_remove(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_remove: no model '%s'", key ? key : "(null)");
    HIPCHK(hipStreamSynchronize(v->stream));
    v->models.erase(key);
    return GSX_OK;
}

gsx_status gsx_model_len(gsx_viewer* v, const char* key, uint64_t* out_count) {
    Model* m = find_model(v, key);
    if (!m || !out_count) return fail(GSX_ERR_NOT_FOUND, "gsx_model_len: no model '%s'", key ? key : "(null)");
    *out_count = m->n;
    return GSX_OK;
}

gsx_status gsx_model_upload_range(gsx_viewer* v, const char* key, uint64_t start, const gsx_gaussian* src, uint64_t n) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_range: no model '%s'", key ? key : "(null)");
    if (n == 0) return GSX_OK;
    if (!src) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_range: src is null");
    if (start > m->n || n > m->n - start)
        return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_range: range [%llu,+%llu) exceeds model length %llu",
                    (unsigned long long)start, (unsigned long long)n, (unsigned long long)m->n);
    // stage in chunks so a multi-GB stream upload needs a bounded staging buffer
    const uint64_t chunk = 1u << 20;
    for (uint64_t off = 0; off < n; off += chunk) {
        uint64_t c = std::min(chunk, n - off);
        HIPCHK(v->staging.ensure(sizeof(gsx_gaussian) * c));
        HIPCHK(hipMemcpyAsync(v->staging.p, src + off, sizeof(gsx_gaussian) * c, hipMemcpyHostToDevice, v->stream));
        HIPCHK(launch_convert(v->stream, v->staging.as<gsx_gaussian>(), c, start + off, m->n, m->pod()));
        HIPCHK(hipStreamSynchronize(v->stream));  // the caller's memory may be reused after return
    }
    return GSX_OK;
}

gsx_status gsx_model_upload_pod_device(gsx_viewer* v, const char* key, uint64_t start, uint64_t n, const float* d_pos,
                                       const uint32_t* d_color, const float* d_sh, const float* d_cov3d) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_pod_device: no model '%s'", key ? key : "(null)");
    if (n == 0) return GSX_OK;
    if (!d_pos || !d_color || !d_cov3d || (m->has_sh && !d_sh))
        return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_pod_device: null plane");
    if (start > m->n || n > m->n - start) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_pod_device: range exceeds model");
    HIPCHK(launch_pack_pod(v->stream, d_pos, d_color, m->has_sh ? d_sh : nullptr, d_cov3d, n, start, m->n, m->pod()));
    return GSX_OK;
}

gsx_status gsx_update_camera(gsx_viewer* v, const float view[16], const float proj[16], uint32_t width, uint32_t height) {
    if (!v || !view || !proj) return fail(GSX_ERR_INVALID_ARG, "gsx_update_camera: null argument");
    if (width == 0 || height == 0 || width > 65535u * GSX_TILE || height > 65535u * GSX_TILE)
        return fail(GSX_ERR_INVALID_ARG, "gsx_update_camera: size %ux%u out of range", width, height);
    memcpy(v->view, view, sizeof(float) * 16);
    memcpy(v->proj, proj, sizeof(float) * 16);
    v->width = width;
    v->height = height;
    return GSX_OK;
}

gsx_status gsx_update_model_transform(gsx_viewer* v, const char* key, const float pos[3], const float quat[4],
                                      const float scale[3]) {
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_update_model_transform: no model '%s'", key ? key : "(null)");
    if (!pos || !quat || !scale) return fail(GSX_ERR_INVALID_ARG, "gsx_update_model_transform: null argument");
    memcpy(m->mt.pos, pos, sizeof(float) * 3);
    memcpy(m->mt.quat, quat, sizeof(float) * 4);
    memcpy(m->mt.scale, scale, sizeof(float) * 3);
    return GSX_OK;
}

gsx_status gsx_update_gaussian_transform(gsx_viewer* v, float size, gsx_display_mode mode, uint32_t sh_deg, uint32_t no_sh0) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_update_gaussian_transform: viewer is null");
    // GaussianShDegree::new returns None above 3 (transform.rs:139)
    if (sh_deg > 3) return fail(GSX_ERR_INVALID_ARG, "gsx_update_gaussian_transform: sh_deg %u > 3", sh_deg);
    if ((int)mode < 0 || (int)mode > GSX_DISPLAY_POINT) return fail(GSX_ERR_INVALID_ARG, "gsx_update_gaussian_transform: bad display mode");
    v->size = size;
    v->display_mode = (uint32_t)mode;
    v->sh_deg = sh_deg;
    v->no_sh0 = no_sh0 ? 1u : 0u;
    return GSX_OK;
}

gsx_status gsx_model_upload_mask(gsx_viewer* v, const char* key, const uint32_t* words, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_mask: no model '%s'", key ? key : "(null)");
    if (!words) {  // MaskOpTree::Reset
        m->has_mask = false;
        return GSX_OK;
    }
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_mask: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    HIPCHK(hipMemcpyAsync(m->mask.p, words, 4 * n_words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipStreamSynchronize(v->stream));
    m->has_mask = true;
    return GSX_OK;
}

gsx_status gsx_model_download_mask(gsx_viewer* v, const char* key, uint32_t* words, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !words) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_mask: no model '%s'", key ? key : "(null)");
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_mask: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    if (!m->has_mask) {
        memset(words, 0xFF, 4 * n_words);
        return GSX_OK;
    }
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(words, m->mask.p, 4 * n_words, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_mask_evaluate(gsx_viewer* v, const char* key, const gsx_mask_op* ops, uint32_t n_ops,
                             const gsx_mask_shape* shapes, uint32_t n_shapes) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_mask_evaluate: no model '%s'", key ? key : "(null)");
    if (n_ops > GSX_MASK_MAX_OPS || n_shapes > GSX_MASK_MAX_SHAPES)
        return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: at most %u ops and %u shapes", GSX_MASK_MAX_OPS, GSX_MASK_MAX_SHAPES);
    if ((n_ops && !ops) || (n_shapes && !shapes)) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: null argument");
    // validate the postfix program: shape indices (validate_shapes, app.rs:1786-1813) and stack discipline
    int depth = 0;
    for (uint32_t k = 0; k < n_ops; ++k) {
        switch (ops[k].opcode) {
            case GSX_MASK_OP_SHAPE:
                if (ops[k].arg >= n_shapes) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: shape index %u out of range", ops[k].arg);
                if (++depth > 32) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: expression nests deeper than 32");
                break;
            case GSX_MASK_OP_COMPLEMENT:
                if (depth < 1) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: malformed postfix program");
                break;
            case GSX_MASK_OP_UNION: case GSX_MASK_OP_INTERSECTION: case GSX_MASK_OP_DIFFERENCE: case GSX_MASK_OP_SYMMETRIC_DIFFERENCE:
                if (depth < 2) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: malformed postfix program");
                --depth;
                break;
            default: return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: unknown opcode %u", ops[k].opcode);
        }
    }
    if (n_ops && depth != 1) return fail(GSX_ERR_INVALID_ARG, "gsx_mask_evaluate: malformed postfix program");
    if (n_ops == 0) {  // MaskOpTree::Reset
        m->has_mask = false;
        return GSX_OK;
    }
    MaskProgram prog{};
    quat_to_rows(m->mt.quat, prog.m_rot);
    memcpy(prog.m_pos, m->mt.pos, sizeof prog.m_pos);
    memcpy(prog.m_scale, m->mt.scale, sizeof prog.m_scale);
    prog.n_shapes = n_shapes;
    prog.n_ops = n_ops;
    for (uint32_t s = 0; s < n_shapes; ++s) {
        prog.shapes[s].kind = shapes[s].kind;
        memcpy(prog.shapes[s].pos, shapes[s].pos, sizeof(float) * 3);
        quat_to_rows(shapes[s].quat_xyzw, prog.shapes[s].rot);
        memcpy(prog.shapes[s].scale, shapes[s].scale, sizeof(float) * 3);
    }
    memcpy(prog.ops, ops, sizeof(gsx_mask_op) * n_ops);
    HIPCHK(launch_mask_evaluate(v->stream, m->pc.as<float4>(), (uint32_t)m->n, prog, m->mask.as<uint32_t>()));
    m->has_mask = true;
    return GSX_OK;
}

gsx_status gsx_preprocess(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_preprocess: no model '%s'", key ? key : "(null)");
    return do_preprocess(v, m);
}

gsx_status gsx_sort(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_sort: no model '%s'", key ? key : "(null)");
    return do_sort(v, m);
}

gsx_status gsx_sync(gsx_viewer* v) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    gsx_status fst = finish_frame(v);
    if (fst) return fst;
    HIPCHK(hipStreamSynchronize(v->stream));
    return GSX_OK;
}

gsx_status gsx_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    return do_render(v, keys, n_keys);
}

gsx_status gsx_render_frame(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    for (uint32_t i = 0; i < n_keys; ++i) {
        Model* m = find_model(v, keys ? keys[i] : nullptr);
        if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_render_frame: no model '%s'", keys && keys[i] ? keys[i] : "(null)");
        if ((st = do_preprocess(v, m))) return st;
        if ((st = do_sort(v, m))) return st;
    }
    return do_render(v, keys, n_keys);
}

gsx_status gsx_download_framebuffer(gsx_viewer* v, float* rgbt, uint64_t n_floats) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    const uint64_t need = 4ull * v->width * v->height;
    if (!rgbt || n_floats != need) return fail(GSX_ERR_INVALID_ARG, "gsx_download_framebuffer: expected %llu floats", (unsigned long long)need);
    if ((st = ensure_fb(v))) return st;
    if ((st = finish_frame(v))) return st;
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(rgbt, fb_ptr(v), sizeof(float) * need, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_download_rgba8(gsx_viewer* v, const float bg[3], uint8_t* rgba, uint64_t n_bytes) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    const uint64_t npx = (uint64_t)v->width * v->height;
    if (!bg || !rgba || n_bytes != 4 * npx) return fail(GSX_ERR_INVALID_ARG, "gsx_download_rgba8: expected %llu bytes", (unsigned long long)(4 * npx));
    if ((st = ensure_fb(v))) return st;
    if ((st = finish_frame(v))) return st;
    HIPCHK(v->scratch.ensure(4 * npx));
    HIPCHK(launch_resolve_rgba8(v->stream, fb_ptr(v), (uint32_t)npx, bg[0], bg[1], bg[2], v->scratch.as<uint32_t>()));
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(rgba, v->scratch.p, 4 * npx, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_framebuffer_device_ptr(gsx_viewer* v, void** out_ptr, uint32_t* out_w, uint32_t* out_h) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!out_ptr) return fail(GSX_ERR_INVALID_ARG, "gsx_framebuffer_device_ptr: null argument");
    if ((st = ensure_fb(v))) return st;
    *out_ptr = fb_ptr(v);
    if (out_w) *out_w = v->width;
    if (out_h) *out_h = v->height;
    return GSX_OK;
}

gsx_status gsx_model_frame_stats(gsx_viewer* v, const char* key, gsx_frame_stats* out) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !out) return fail(GSX_ERR_NOT_FOUND, "gsx_model_frame_stats: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_model_frame_stats: model '%s' not preprocessed this frame", key);
    if ((st = finish_frame(v))) return st;
    out->n_gaussians = m->n;
    out->n_visible = m->n_visible;
    out->n_tile_entries = m->binned ? m->n_entries : 0;  // entries actually binned by the last gsx_render
    out->n_sorted = m->n_sorted;
    out->speculated = m->binned && m->spec_round1 ? 1u : 0u;
    out->n_repair_tiles = out->speculated ? m->h_counters->spec_need : 0;
    out->n_repair_sorted = out->speculated ? m->n_sorted2 : 0;
    out->reserved = 0;
    return GSX_OK;
}

gsx_status gsx_model_download_projection(gsx_viewer* v, const char* key, uint32_t* depth_key, uint32_t* rect, float* mean2d,
                                         float* conic_opacity, float* rgb) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_projection: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_projection: model '%s' not preprocessed", key);
    if ((st = complete_records(v, m))) return st;
    HIPCHK(hipStreamSynchronize(v->stream));
    const size_t n = m->rec_n;  // == model length unless records were imported (gsx_shard_import)
    std::vector<uint32_t> k(n);
    std::vector<float4> a(n), b(n), c(n);
    const Records rr = m->rec();
    if (n) {
        HIPCHK(hipMemcpy(k.data(), rr.key, 4 * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(a.data(), rr.a, 16 * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), rr.b, 16 * n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(c.data(), rr.c, 16 * n, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; ++i) {
        const bool vis = k[i] != kCulledKey;
        if (depth_key) depth_key[i] = k[i];
        uint32_t rx, ry;
        memcpy(&rx, &a[i].z, 4);
        memcpy(&ry, &a[i].w, 4);
        if (rect) {
            rect[4 * i + 0] = vis ? (rx & 0xFFFFu) : 0;
            rect[4 * i + 1] = vis ? (ry & 0xFFFFu) : 0;
            rect[4 * i + 2] = vis ? (rx >> 16) : 0;
            rect[4 * i + 3] = vis ? (ry >> 16) : 0;
        }
        if (mean2d) {
            mean2d[2 * i] = vis ? a[i].x : 0.0f;
            mean2d[2 * i + 1] = vis ? a[i].y : 0.0f;
        }
        if (conic_opacity) {
            conic_opacity[4 * i] = vis ? b[i].x : 0.0f;
            conic_opacity[4 * i + 1] = vis ? b[i].y : 0.0f;
            conic_opacity[4 * i + 2] = vis ? b[i].z : 0.0f;
            conic_opacity[4 * i + 3] = vis ? b[i].w : 0.0f;
        }
        if (rgb) {
            rgb[3 * i] = vis ? c[i].x : 0.0f;
            rgb[3 * i + 1] = vis ? c[i].y : 0.0f;
            rgb[3 * i + 2] = vis ? c[i].z : 0.0f;
        }
    }
    return GSX_OK;
}

gsx_status gsx_model_download_sorted(gsx_viewer* v, const char* key, uint32_t* indices, uint64_t capacity, uint64_t* out_n_visible) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_sorted: no model '%s'", key ? key : "(null)");
    if (!m->sorted) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_sorted: model '%s' not sorted", key);
    if ((st = sync_counters(v))) return st;
    const uint32_t n_order = m->spec_round1 && m->binned ? m->n_sorted2 : m->n_sorted;  // a speculated frame leaves its repair order
    if (out_n_visible) *out_n_visible = n_order;
    if (indices) {
        if (capacity < n_order) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_sorted: capacity %llu < %u sorted records", (unsigned long long)capacity, n_order);
        HIPCHK(hipStreamSynchronize(v->stream));
        if (n_order) HIPCHK(hipMemcpy(indices, m->sorted_idx, 4ull * n_order, hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

gsx_status gsx_model_download_tile_lists(gsx_viewer* v, const char* key, uint32_t* tile_offsets, uint64_t n_offsets,
                                         uint32_t* list, uint64_t capacity) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_tile_lists: no model '%s'", key ? key : "(null)");
    if (!m->binned) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: model '%s' not rendered this frame", key);
    if (!m->lists_complete)
        return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: complete lists exist only for the front-most model "
                    "rendered with gsx_render_options.progressive = 0 (model '%s' was rendered in depth slabs)", key);
    if ((st = finish_frame(v))) return st;
    const uint32_t n_tiles = m->fc.tiles_x * m->fc.tiles_y;
    if (n_offsets != (uint64_t)n_tiles + 1) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: expected %u offsets", n_tiles + 1);
    HIPCHK(hipStreamSynchronize(v->stream));
    std::vector<uint2> r(n_tiles);
    HIPCHK(hipMemcpy(r.data(), m->ranges.p, sizeof(uint2) * n_tiles, hipMemcpyDeviceToHost));
    uint32_t off = 0;
    for (uint32_t t = 0; t < n_tiles; ++t) {
        if (r[t].y > r[t].x && r[t].x != off)
            return fail(GSX_ERR_HIP, "gsx_model_download_tile_lists: tile %u range [%u,%u) not contiguous at %u", t, r[t].x, r[t].y, off);
        if (tile_offsets) tile_offsets[t] = off;
        off += r[t].y - r[t].x;
    }
    if (tile_offsets) tile_offsets[n_tiles] = off;
    if (off != m->n_entries) return fail(GSX_ERR_HIP, "gsx_model_download_tile_lists: ranges cover %u entries, D = %u", off, m->n_entries);
    if (list) {
        if (capacity < m->n_entries) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: capacity too small");
        if (m->n_entries) HIPCHK(hipMemcpy(list, m->tile_list, 4ull * m->n_entries, hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

gsx_status gsx_model_download_pod(gsx_viewer* v, const char* key, float* pos, uint32_t* color, float* sh, float* cov3d) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_pod: no model '%s'", key ? key : "(null)");
    if (!pos || !color || !cov3d) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_pod: null output");
    const size_t n = m->n;
    if (!n) return GSX_OK;
    DevBuf dpos, dcol, dsh, dcov;
    HIPCHK(dpos.ensure(12 * n));
    HIPCHK(dcol.ensure(4 * n));
    HIPCHK(dcov.ensure(24 * n));
    if (sh) HIPCHK(dsh.ensure(180 * n));
    HIPCHK(launch_unpack_pod(v->stream, m->pod(), n, dpos.as<float>(), dcol.as<uint32_t>(), sh ? dsh.as<float>() : nullptr,
                             dcov.as<float>()));
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(pos, dpos.p, 12 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(color, dcol.p, 4 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cov3d, dcov.p, 24 * n, hipMemcpyDeviceToHost));
    if (sh) HIPCHK(hipMemcpy(sh, dsh.p, 180 * n, hipMemcpyDeviceToHost));
    return GSX_OK;
}

// ---- multi-GPU stage split ----------------------------------------------------------------------
static uint32_t rows_per_rank(const gsx_viewer* v, uint32_t world) {
    uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    return (tiles_y + world - 1) / world;
}

gsx_status gsx_shard_layout(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_shard_layout_t* out) {
    if (!v || !out || world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_layout: bad argument");
    const uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE, rpr = rows_per_rank(v, world);
    out->rows_per_rank = rpr;
    out->row_lo = std::min(rank * rpr, tiles_y);
    out->row_hi = std::min((rank + 1) * rpr, tiles_y);
    out->band_bytes = (uint64_t)rpr * GSX_TILE * v->width * sizeof(float4);
    out->band_offset_bytes = (uint64_t)rank * out->band_bytes;
    out->padded_framebuffer_bytes = (uint64_t)world * out->band_bytes;
    return GSX_OK;
}

gsx_status gsx_viewer_set_external_framebuffer(gsx_viewer* v, void* d_ptr, uint64_t bytes) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if ((st = finish_frame(v))) return st;
    v->ext_fb = d_ptr;
    v->ext_fb_bytes = d_ptr ? bytes : 0;
    return GSX_OK;
}

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
    HIPCHK(hipMemcpyAsync(v->query_texture.p, texels, (size_t)width * height, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipStreamSynchronize(v->stream));
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
        m->flags_kind = GSX_QUERY_NONE;  // consumed: one selection op per evaluated query
    }
    return GSX_OK;
}

gsx_status gsx_model_upload_selection(gsx_viewer* v, const char* key, const uint32_t* words, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_upload_selection: no model '%s'", key ? key : "(null)");
    if (!words) {  // clear
        m->has_selection = false;
        if (m->selection.p) HIPCHK(hipMemsetAsync(m->selection.p, 0, m->selection.bytes, v->stream));
        return GSX_OK;
    }
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_selection: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    if ((st = ensure_selection(v, m))) return st;
    HIPCHK(hipMemcpyAsync(m->selection.p, words, 4 * n_words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipStreamSynchronize(v->stream));
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
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(words, m->selection.p, 4 * n_words, hipMemcpyDeviceToHost));
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
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(bits.data(), m->edited.p, 4 * words, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(a.data(), m->edit_a.p, 16 * n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), m->edit_b.p, 16 * n, hipMemcpyDeviceToHost));
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
    if (!edits) {  // drop every stored edit
        m->has_edits = false;
        if (m->edited.p) HIPCHK(hipMemsetAsync(m->edited.p, 0, m->edited.bytes, v->stream));
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
    HIPCHK(hipMemcpyAsync(m->edited.p, bits.data(), 4 * words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipMemcpyAsync(m->edit_a.p, a.data(), 16 * n, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipMemcpyAsync(m->edit_b.p, b.data(), 16 * n, hipMemcpyHostToDevice, v->stream));
    HIPCHK(hipStreamSynchronize(v->stream));
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
    HIPCHK(hipStreamSynchronize(v->stream));
    HIPCHK(hipMemcpy(&cnt, m->hit_count.p, 4, hipMemcpyDeviceToHost));
    cnt = std::min<uint32_t>(cnt, GSX_QUERY_MAX_HITS);
    std::vector<gsx_query_hit> h(cnt);
    if (cnt) HIPCHK(hipMemcpy(h.data(), m->hits.p, sizeof(gsx_query_hit) * cnt, hipMemcpyDeviceToHost));
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

static size_t window_bytes(const gsx_viewer* v) {
    return sizeof(uint2) * (size_t)((v->width + GSX_TILE - 1) / GSX_TILE) * ((v->height + GSX_TILE - 1) / GSX_TILE);
}

gsx_status gsx_shard_pack(gsx_viewer* v, const char* key, uint32_t world, const uint32_t* d_tile_window, void* d_send,
                          uint64_t capacity_records, uint64_t* counts) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_pack: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: model '%s' has no projection this frame (gsx_preprocess first)", key);
    if (world == 0 || world > 64 || !counts) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: world must be 1..64");
    const uint32_t n = (uint32_t)m->n;
    const uint32_t nb = (uint32_t)pack_blocks(n), rpr = rows_per_rank(v, world);
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->pack_table.ensure(4 * ((size_t)64 * std::max(nb, 1u) + 64)));
    HIPCHK(m->pack_masks.ensure(8 * (size_t)std::max(n, 1u)));
    const uint2* window = nullptr;
    const uint2* list = nullptr;
    const uint32_t* d_list_n = nullptr;
    unsigned long long* travellers = nullptr;
    uint32_t* trav_counts = nullptr;
    Counters* dc = m->counters.as<Counters>();
    if (d_tile_window) {  // own copy: the caller's map need not outlive this call
        HIPCHK(m->pack_window.ensure(window_bytes(v)));
        HIPCHK(hipMemcpyAsync(m->pack_window.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
        window = m->pack_window.as<uint2>();
        if (m->lazy) {  // explicit windows on a lazily projected shard (the repair exchange): travellers may be unshaded
            HIPCHK(m->trav_ballots.ensure(8 * ((std::max<size_t>(n, 1) + 63) / 64)));
            HIPCHK(m->trav_counts.ensure(4 * std::max<size_t>(nb, 1)));
            travellers = m->trav_ballots.as<unsigned long long>();
            trav_counts = m->trav_counts.as<uint32_t>();
        }
    } else if (m->shard_win_set && m->cand_valid) {  // the windows given to gsx_shard_set_windows: only the candidates are looked at
        window = m->shard_win.as<uint2>();
        list = m->adm_pairs.as<uint2>();
        d_list_n = &dc->n_candidates;
    } else if (m->lazy) {
        if ((st = complete_records(v, m))) return st;  // everything travels: every record must be whole
    }
    uint32_t* table = m->pack_table.as<uint32_t>();
    uint32_t* totals = table + (size_t)64 * std::max(nb, 1u);
    unsigned long long* masks = m->pack_masks.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(totals, 0, 4 * 64, v->stream));
    HIPCHK(launch_pack_count(v->stream, m->proj_rec(), n, world, rpr, window, tiles_x, masks, table, list, d_list_n, travellers, trav_counts));
    if (nb) HIPCHK(launch_rowscan(v->stream, table, world, nb, totals));
    if (travellers && nb) {
        // shade the travellers the first round did not: compact their indices, k_shade skips what is shaded already
        HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(n, 1)));
        HIPCHK(launch_rowscan(v->stream, trav_counts, 1, nb, &dc->n_sorted2));
        HIPCHK(launch_admit_scatter(v->stream, m->proj_rec().key, n, travellers, trav_counts, m->adm_pairs.as<uint2>()));
        PodPlanes pod = m->pod();
        pod.mask = m->last_pod_mask;
        HIPCHK(launch_shade(v->stream, m->fc, n, pod, m->proj_rec(),
                            LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted2, m->adm_ballots.as<unsigned long long>()}));
        m->cand_valid = false;  // adm_pairs now holds the repair travellers
    }
    uint32_t h_tot[64];
    HIPCHK(hipMemcpyAsync(h_tot, totals, 4 * 64, hipMemcpyDeviceToHost, v->stream));
    HIPCHK(hipStreamSynchronize(v->stream));
    uint64_t sum = 0;
    for (uint32_t g = 0; g < world; ++g) {
        counts[g] = h_tot[g];
        sum += h_tot[g];
    }
    if (sum > capacity_records)
        return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: %llu records exceed the send capacity %llu",
                    (unsigned long long)sum, (unsigned long long)capacity_records);
    if (sum && !d_send) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: d_send is null");
    HIPCHK(launch_pack_scatter(v->stream, m->proj_rec(), n, world, masks, table, totals, d_send, capacity_records, list, d_list_n));
    return GSX_OK;
}

gsx_status gsx_viewer_set_band(gsx_viewer* v, uint32_t row_lo, uint32_t row_hi) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_band: viewer is null");
    if (row_lo > row_hi) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_band: row_lo %u > row_hi %u", row_lo, row_hi);
    v->band_lo = row_lo;
    v->band_hi = row_hi;
    return GSX_OK;
}

gsx_status gsx_shard_set_windows(gsx_viewer* v, const char* key, const uint32_t* d_tile_window) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_set_windows: no model '%s'", key ? key : "(null)");
    m->shard_win_set = d_tile_window != nullptr;
    if (!d_tile_window) return GSX_OK;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->shard_win.ensure(window_bytes(v)));
    HIPCHK(hipMemcpyAsync(m->shard_win.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
    HIPCHK(m->shard_pyr.ensure(4 * window_pyramid_words(tiles_x, tiles_y)));
    HIPCHK(launch_window_pyramid(v->stream, m->shard_win.as<uint2>(), tiles_x, tiles_y, m->shard_pyr.as<uint32_t>()));
    m->shard_tiles_x = tiles_x;
    m->shard_tiles_y = tiles_y;
    return GSX_OK;
}

gsx_status gsx_shard_import(gsx_viewer* v, const char* key, const void* d_recv, uint64_t n_records, uint32_t world,
                            uint32_t rank, const uint32_t* d_tile_window) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_import: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import: model '%s' has no frame constants (gsx_preprocess first)", key);
    if (world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import: bad world/rank %u/%u", world, rank);
    if (n_records >= 0xFFFFFFF0ull) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import: too many records");
    if (n_records && !d_recv) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import: d_recv is null");
    if ((st = ensure_import_capacity(m, n_records))) return st;
    HIPCHK(launch_import_records(v->stream, d_recv, (uint32_t)n_records, m->imp_rec()));
    // every imported record is visible by construction
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&m->counters.as<Counters>()->n_visible), (int)(uint32_t)n_records, 2,
                             v->stream));  // n_visible and n_sorted
    m->stats_pending = true;
    m->rec_n = n_records;
    m->use_imported = true;
    const uint32_t rpr = rows_per_rank(v, world);
    m->row_lo = rank * rpr;
    m->row_hi = (rank + 1) * rpr;
    m->has_window = d_tile_window != nullptr;
    if (d_tile_window) {
        HIPCHK(m->window.ensure(window_bytes(v)));
        HIPCHK(hipMemcpyAsync(m->window.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
    }
    m->sorted = m->counters_valid = m->binned = false;
    return GSX_OK;
}

// this rank's band of the per-tile saturation keys; rows below the frame read 0 (= open)
__global__ void k_shard_feedback(const uint32_t* __restrict__ tile_sat, uint32_t tiles_x, uint32_t tiles_y, uint32_t row_lo,
                                 uint32_t n_words, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t ty = row_lo + i / tiles_x;
    out[i] = ty < tiles_y ? tile_sat[ty * tiles_x + i % tiles_x] : 0u;
}

gsx_status gsx_shard_feedback_words(gsx_viewer* v, uint32_t world, uint32_t* out_words) {
    if (!v || !out_words || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback_words: bad argument");
    *out_words = rows_per_rank(v, world) * ((v->width + GSX_TILE - 1) / GSX_TILE);
    return GSX_OK;
}

gsx_status gsx_shard_feedback(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, void* d_out_u32) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_out_u32) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_feedback: no model '%s'", key ? key : "(null)");
    if (!m->binned) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback: model '%s' not rendered this frame", key);
    if (!v->options.progressive) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback needs gsx_render_options.progressive = 1");
    if (world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback: bad world/rank %u/%u", world, rank);
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    const uint32_t row_words = (tiles_x + 31) / 32, rpr = rows_per_rank(v, world), n_words = rpr * tiles_x;
    const uint32_t* tile_sat = v->done_bits.as<uint32_t>() + 1 + (size_t)row_words * tiles_y;
    hipLaunchKernelGGL(k_shard_feedback, dim3((n_words + 255) / 256), dim3(256), 0, v->stream, tile_sat, tiles_x, tiles_y,
                       rank * rpr, n_words, static_cast<uint32_t*>(d_out_u32));
    HIPCHK(hipGetLastError());
    return GSX_OK;
}

gsx_status gsx_render_more(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    return do_render(v, keys, n_keys, true);
}

gsx_status gsx_set_pass_timing(gsx_viewer* v, uint32_t enabled) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_set_pass_timing: viewer is null");
    v->timing = enabled == 1u ? 0xFFFFFFFFu : (enabled >> 1);  // 1 = every pass; otherwise bit (p + 1) selects pass p
    return GSX_OK;
}

gsx_status gsx_get_pass_timing(gsx_viewer* v, float ms[GSX_PASS_COUNT], uint32_t launches[GSX_PASS_COUNT]) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    HIPCHK(hipStreamSynchronize(v->stream));
    for (auto& t : v->timers) {
        float e = 0.0f;
        if (hipEventElapsedTime(&e, t.start, t.stop) == hipSuccess) v->pass_ms[t.pass] += e;
        v->event_pool.push_back({t.start, t.stop});
    }
    v->timers.clear();
    for (int i = 0; i < GSX_PASS_COUNT; ++i) {
        if (ms) ms[i] = v->pass_ms[i];
        if (launches) launches[i] = v->pass_launches[i];
        v->pass_ms[i] = 0.0f;
        v->pass_launches[i] = 0;
    }
    return GSX_OK;
}

}  // extern "C"
