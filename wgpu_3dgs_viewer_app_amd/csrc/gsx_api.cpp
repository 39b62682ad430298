// gsx_api.cpp — the C ABI of include/gsx.h: viewer / model lifetime, uploads, uniforms, frame entry points, readback.
// (Selection / edits / queries: gsx_api_edit.cpp; multi-GPU: gsx_api_shard.cpp; what a frame enqueues: gsx_frame.cpp.)
// There is NO CPU fallback: without a HIP device every entry point fails with GSX_ERR_NO_DEVICE.
#include "gsx_state.h"
#include <atomic>
#include <cstdlib>

using namespace gsx;

// ---- frames in flight (gsx_render_options::frames_in_flight) ----
// A lane is a gsx_viewer of its own — stream, framebuffer, per-model records / sort / tile buffers, speculation windows —
// created by the viewer it belongs to; its models are shadows that VIEW the owner's Gaussian data (DevBuf::borrow).
// gsx_render_frame deals frames round-robin to the viewer and its lanes; nothing else in the library knows about lanes
// except viewer_bind (gsx_state.h), which orders the viewer's stream after the lanes' frames before any other call.
static std::atomic<uint64_t> g_model_serial{0};  // (viewers of different host threads create models concurrently)
static std::atomic<int> g_viewers_on_device[64];  // live top-level viewers of this process per device (lane_create's probe asks)

// per-model results: the lane that rendered the newest frame — unless that frame did not include the model (a key that was
// last rendered in an earlier frame lives where that frame ran; the viewer itself is the best answer left)
static gsx_viewer* result_lane_of(gsx_viewer* v, const char* key) {
    gsx_viewer* l = result_lane(v);
    return (l != v && !find_model(l, key)) ? v : l;
}

// Does work on `candidate` run while each of `busy` is occupied?  HIP multiplexes its streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default) in creation order; two streams on one queue execute one after the other, and a lane that
// shares its queue with the viewer (or with another lane) overlaps nothing: measured on cfg4 with a second RCCL communicator in
// the process (its streams shifted the lane onto the viewer's queue): 1729 -> 1245 fps with two frames in flight.
// Probe: occupy every busy stream with a spinning wave, run an empty kernel on the candidate, and see whether it completed
// while all the spinners were still at it.
static gsx_status stream_runs_beside(hipStream_t candidate, const std::vector<hipStream_t>& busy, bool* out) {
    *out = true;
    std::vector<hipEvent_t> ev(busy.size() + 1, nullptr);
    gsx_status st = GSX_OK;
    auto run = [&]() -> gsx_status {
        for (auto& e : ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipStream_t b : busy) HIPCHK(gsx::op::StreamSynchronize(b));
        HIPCHK(gsx::op::StreamSynchronize(candidate));
        for (size_t i = 0; i < busy.size(); ++i) {
            HIPCHK(launch_spin(busy[i], 400));
            HIPCHK(gsx::op::EventRecord(ev[i], busy[i]));
        }
        HIPCHK(launch_spin(candidate, 1));
        HIPCHK(gsx::op::EventRecord(ev.back(), candidate));
        HIPCHK(gsx::op::EventSynchronize(ev.back()));
        for (size_t i = 0; i < busy.size(); ++i)
            if (hipEventQuery(ev[i]) == hipSuccess) *out = false;  // that spinner finished first: the candidate waited behind it
        for (hipStream_t b : busy) HIPCHK(gsx::op::StreamSynchronize(b));
        return GSX_OK;
    };
    st = run();
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    return st;
}

static gsx_status lane_create(gsx_viewer* v, gsx_viewer** out) {
    std::unique_ptr<gsx_viewer> l(new gsx_viewer());
    l->device = v->device;
    l->parent = v;
    l->validate = v->validate;
    l->tile_cap_fixed = v->tile_cap_fixed;
    l->bin_mode = v->bin_mode;
    l->edit_cache = v->edit_cache;
    l->blocks_max = v->blocks_max;
    l->blocks_adaptive = v->blocks_adaptive;
    l->tile_order_on = v->tile_order_on;
    l->bucket_sort = v->bucket_sort;
    l->bin_fused = v->bin_fused;
    l->sorted_records = v->sorted_records;
    l->tile_profile = v->tile_profile;
    // a stream that does not share its hardware queue with the viewer's or another lane's: streams that do are kept (parked)
    // until the viewer goes, so that the next one created lands on the next queue
    std::vector<hipStream_t> busy{v->stream};
    for (gsx_viewer* o : v->lanes) busy.push_back(o->stream);
    // The probe decides from timing: with other viewers at work on the same device (ranks as threads of one process, several
    // processes per GPU) their kernels delay the candidate and the verdict is noise — every false "shares a queue" would park a
    // stream for the viewer's lifetime and spin 400 us on every busy stream.  So: only while this is the one viewer with lanes on
    // its device, and never more than kMaxParked parked streams per viewer.
    constexpr size_t kMaxParked = 8;
    const bool probe = g_viewers_on_device[v->device & 63].load() <= 1;
    for (int attempt = 0; attempt < 8; ++attempt) {
        HIPCHK(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
        bool beside = true;
        if (probe && v->parked_streams.size() < kMaxParked) {
            const gsx_status pst = stream_runs_beside(l->stream, busy, &beside);
            if (pst) return pst;
        }
        if (beside || attempt == 7) break;  // (eight tries without luck: fewer hardware queues than lanes — frames still come out right)
        v->parked_streams.push_back(l->stream);
        l->stream = nullptr;
    }
    l->own_stream = true;
    HIPCHK(hipEventCreateWithFlags(&l->lane_event, hipEventDisableTiming));
    *out = l.release();
    return GSX_OK;
}

// may this frame go to a lane?  (a frame with a query — its flags feed gsx_postprocess — and everything multi-GPU stays on the
// viewer; selections, edits and the highlight travel: a lane's shadow models view the owner's selection and edit buffers, which
// the owner prepares before the frame is dealt out, prepare_edits_for_lanes)
static bool frame_may_overlap(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    if (v->parent || v->options.frames_in_flight < 2 || v->query.kind != GSX_QUERY_NONE || v->ext_fb || v->band_lo != 0 ||
        v->band_hi != 0xFFFFFFFFu)
        return false;
    for (uint32_t i = 0; i < n_keys; ++i) {
        Model* m = find_model(v, keys ? keys[i] : nullptr);
        if (!m || m->shard_win_set || m->shard_limit_valid || m->shard_next_valid) return false;
    }
    return n_keys > 0;
}

// The edit records and the keep-bitset of the frame's models live with the owner and are read by whichever lane renders a
// frame.  When k_edit_prepare has to run again (a selection, mask, edit or selection-edit change since it last did) it runs on
// the owner's stream, AFTER every frame in flight that still reads the old state and BEFORE every later frame on any lane.
namespace gsx {
gsx_status prepare_edits_for_lanes(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    for (uint32_t i = 0; i < n_keys; ++i) {
        Model* m = find_model(v, keys[i]);
        if (!m || !edits_need_prepare(v, m)) continue;
        for (gsx_viewer* l : v->lanes)
            if (l->lane_busy) {
                HIPCHK(gsx::op::StreamWaitEvent(v->stream, l->lane_event, 0));
                l->lane_busy = false;
            }
        gsx_status st = prepare_edits(v, m, nullptr);
        if (st) return st;
        v->epoch += 1;  // every lane's next frame waits for the owner's stream (lane_sync)
    }
    return GSX_OK;
}
}  // namespace gsx

// bring lane l up to date with viewer v for a frame of `keys`: uniforms, options, and a shadow of every model
static gsx_status lane_sync(gsx_viewer* v, gsx_viewer* l, const char* const* keys, uint32_t n_keys) {
    l->params = v->params;
    memcpy(l->view, v->view, sizeof l->view);
    memcpy(l->proj, v->proj, sizeof l->proj);
    l->width = v->width;
    l->height = v->height;
    l->size = v->size;
    l->display_mode = v->display_mode;
    l->sh_deg = v->sh_deg;
    l->no_sh0 = v->no_sh0;
    memcpy(l->highlight, v->highlight, sizeof l->highlight);
    l->sel_edit = v->sel_edit;
    l->band_lo = v->band_lo;
    l->band_hi = v->band_hi;
    if (l->options.progressive != v->options.progressive || l->options.speculative != v->options.speculative)
        for (auto& kv : l->models) kv.second->spec_round1 = kv.second->sorted = false;
    l->options = v->options;
    l->options.frames_in_flight = 1;
    l->timing = v->timing;
    for (uint32_t i = 0; i < n_keys; ++i) {
        Model* pm = find_model(v, keys[i]);
        Model* sm = find_model(l, keys[i]);
        if (sm && sm->shadow_of != pm->serial) {  // the key names another model now
            HIPCHK(gsx::op::StreamSynchronize(l->stream));
            l->models.erase(keys[i]);
            sm = nullptr;
        }
        if (!sm) {
            std::unique_ptr<Model> m(new Model());
            m->key = pm->key;
            m->shadow_of = pm->serial;
            m->n = pm->n;
            m->sh_kind = pm->sh_kind;
            m->cov_kind = pm->cov_kind;
            m->has_sh = pm->has_sh;
            HIPCHK(m->counters.ensure(sizeof(Counters)));
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&m->h_counters), sizeof(Counters), hipHostMallocDefault));
            memset(m->h_counters, 0, sizeof(Counters));
            HIPCHK(gsx::op::MemsetAsync(m->counters.p, 0, sizeof(Counters), l->stream));
            sm = m.get();
            l->models[pm->key] = std::move(m);
        }
        for (auto pr : {std::make_pair(&sm->pc, &pm->pc), std::make_pair(&sm->cov_a, &pm->cov_a), std::make_pair(&sm->cov_b, &pm->cov_b),
                        std::make_pair(&sm->sh4, &pm->sh4), std::make_pair(&sm->sh1, &pm->sh1), std::make_pair(&sm->sh_h, &pm->sh_h),
                        std::make_pair(&sm->sh_q, &pm->sh_q), std::make_pair(&sm->sh_aos, &pm->sh_aos), std::make_pair(&sm->cov_h, &pm->cov_h),
                        std::make_pair(&sm->cov_h2, &pm->cov_h2), std::make_pair(&sm->mask, &pm->mask),
                        std::make_pair(&sm->selection, &pm->selection), std::make_pair(&sm->edited, &pm->edited),
                        std::make_pair(&sm->edit_a, &pm->edit_a), std::make_pair(&sm->edit_b, &pm->edit_b), std::make_pair(&sm->keep, &pm->keep)})
            pr.first->borrow(*pr.second);
        sm->has_selection = pm->has_selection;
        sm->has_edits = pm->has_edits;
        sm->tuner_ref = &pm->tuner;  // one speculate-or-not cycle per model, whichever lane renders the frame
        sm->has_mask = pm->has_mask;
        sm->mask_program_hash = pm->mask_program_hash;
        sm->mt = pm->mt;
        sm->show_unedited = pm->show_unedited;
        sm->slot_force = pm->slot_force;  // gsx_shard_set_slot_records may have been called before this lane (or this shadow) existed
    }
    // whatever the caller enqueued on the viewer's stream since this lane's last frame (uploads, masks) comes first
    if (l->seen_epoch != v->epoch) {
        if (!v->lane_event) HIPCHK(hipEventCreateWithFlags(&v->lane_event, hipEventDisableTiming));
        HIPCHK(gsx::op::EventRecord(v->lane_event, v->stream));
        HIPCHK(gsx::op::StreamWaitEvent(l->stream, v->lane_event, 0));
        l->seen_epoch = v->epoch;
    }
    return GSX_OK;
}

namespace gsx {
// May an index-sharded frame of these models run on a lane?  A lane's shadow models view the Gaussian data, the mask, the
// transform and the owner's selection / edit records (prepare_edits_for_lanes); a query's flags live with the viewer itself (as
// for gsx_render_frame: frame_may_overlap).  tests/test_gpu_shard_lib.py::test_fuzz_sharded_against_single found the hole when
// lanes did not see selections and edits at all.
bool shard_frame_may_use_lanes(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    if (v->query.kind != GSX_QUERY_NONE) return false;
    for (uint32_t i = 0; i < n_keys; ++i)
        if (!find_model(v, keys[i])) return false;
    return true;
}

gsx_status lane_acquire(gsx_viewer* v, uint32_t index, const char* const* keys, uint32_t n_keys, gsx_viewer** out) {
    *out = v;
    if (index == 0) return GSX_OK;
    gsx_status st = GSX_OK;
    while (v->lanes.size() < index) {
        gsx_viewer* l = nullptr;
        if ((st = lane_create(v, &l))) return st;
        v->lanes.push_back(l);
        l->lane_index = (uint32_t)v->lanes.size();
    }
    *out = v->lanes[index - 1];
    return lane_sync(v, *out, keys, n_keys);
}
}  // namespace gsx

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
    v->validate = getenv("GSX_VALIDATE") != nullptr;
    if (const char* tc = getenv("GSX_TILE_CAP")) v->tile_cap_fixed = std::max<long long>(atoll(tc), 1);
    if (const char* bm = getenv("GSX_BIN")) v->bin_mode = atoi(bm) ? 1 : 0;
    v->edit_cache = getenv("GSX_NO_EDIT_CACHE") == nullptr;
    v->shard_pair_slots = true;   // (round 3's one slot size for every pair lost the A/B of round 4: profiles/r04_*; gsx_shard_set_slot_records still forces a size)
    v->tile_profile = getenv("GSX_TILE_PROFILE") != nullptr;
    if (const char* e = getenv("GSX_TILE_ORDER")) v->tile_order_on = atoi(e) != 0;
    if (const char* e = getenv("GSX_BUCKET_SORT")) v->bucket_sort = atoi(e) != 0;
    if (const char* e = getenv("GSX_BIN_FUSED")) v->bin_fused = atoi(e) != 0;
    if (const char* e = getenv("GSX_BIN_BIG_RECT")) block_bin_set_big_rect((uint32_t)atoi(e));   // tests / tuning
    if (const char* e = getenv("GSX_BIN_BIG_SLAB")) block_bin_set_big_slab((uint32_t)atoi(e));   // tests: smaller slabs take the eight-per-lane tiles
    if (const char* e = getenv("GSX_BUCKET_CAP")) bucket_sort_set_cap((uint32_t)atoi(e));   // tests: buckets above this take the global-memory path
    if (const char* e = getenv("GSX_SORTED_RECORDS")) v->sorted_records = atoi(e) != 0 ? 1 : 0;
    if (const char* bx = getenv("GSX_BLOCKS_MAX")) {
        v->blocks_max = (uint32_t)std::max(16, std::min(1024, atoi(bx)));
        v->blocks_adaptive = false;
    }
    (void)radix_lane_ordered_adds();  // probes THIS device once per process (the answer is kept per device)
    v->device = desc->device;
    if (desc->stream) {
        v->stream = reinterpret_cast<hipStream_t>(desc->stream);
    } else {
        HIPCHK(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
        v->own_stream = true;
    }
    gsx_spec_params_default(&v->params);
    gsx_render_options_default(&v->options);   // (one source for the defaults: a field added to the struct cannot be forgotten here)
    v->width = std::max(1u, desc->width);
    v->height = std::max(1u, desc->height);
    static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(v->view, ident, sizeof ident);
    memcpy(v->proj, ident, sizeof ident);
    g_viewers_on_device[v->device & 63].fetch_add(1);
    *out = v.release();
    return GSX_OK;
}

// GSX_MEM_DEBUG=1: what every model of a viewer (and of its lanes) holds on the device, buffer by buffer, when the viewer goes
static void mem_debug(const gsx_viewer* v, const char* who) {
    for (const auto& kv : v->models) {
        const Model* m = kv.second.get();
#define GSX_MB(b) if (m->b.bytes && !m->b.borrowed) fprintf(stderr, "[gsx mem] %s model '%s' %-16s %9.1f MB  %6.1f B / Gaussian\n", who, m->key.c_str(), #b, m->b.bytes / 1e6, (double)m->b.bytes / (double)std::max<uint64_t>(m->n, 1));
        GSX_MB(pc) GSX_MB(cov_a) GSX_MB(cov_b) GSX_MB(sh4) GSX_MB(sh1) GSX_MB(sh_h) GSX_MB(sh_q) GSX_MB(sh_aos) GSX_MB(cov_h) GSX_MB(cov_h2) GSX_MB(mask)
        GSX_MB(key_buf) GSX_MB(rec_a) GSX_MB(rec_b) GSX_MB(rec_c) GSX_MB(rect8) GSX_MB(imp_key) GSX_MB(imp_a) GSX_MB(imp_b) GSX_MB(imp_c)
        GSX_MB(dp_a) GSX_MB(dp_b) GSX_MB(sk_out) GSX_MB(sv_out) GSX_MB(sort_ws) GSX_MB(cnt) GSX_MB(block_sums) GSX_MB(srect) GSX_MB(block_vis)
        GSX_MB(tp_src) GSX_MB(tp_a) GSX_MB(tp_b) GSX_MB(tk_out) GSX_MB(tv_out) GSX_MB(tsort_ws) GSX_MB(brec_sorted) GSX_MB(adm_pairs) GSX_MB(adm_ballots)
        GSX_MB(adm_counts) GSX_MB(adm_offsets) GSX_MB(adm_ballots2) GSX_MB(adm_counts2) GSX_MB(pack_masks) GSX_MB(edit_a) GSX_MB(edit_b) GSX_MB(msd_ws) GSX_MB(bin_ws)
#undef GSX_MB
    }
}

void gsx_viewer_destroy(gsx_viewer* v) {
    if (!v) return;
    if (getenv("GSX_MEM_DEBUG")) mem_debug(v, v->parent ? "lane" : "viewer");
    (void)hipSetDevice(v->device);
    (void)gsx_viewer_comm_destroy(v);  // first: it drains the lanes' streams, then destroys the communicators (a lane has none of its own)
    for (gsx_viewer* l : v->lanes) gsx_viewer_destroy(l);  // (synchronises the lane's stream first)
    v->lanes.clear();
    (void)gsx::op::StreamSynchronize(v->stream);
    if (v->lane_event) (void)hipEventDestroy(v->lane_event);
    for (hipStream_t ps : v->parked_streams) (void)hipStreamDestroy(ps);
    if (v->h_shard_verdict) (void)hipHostFree(v->h_shard_verdict);
    if (v->h_verdict_ring) (void)hipHostFree(v->h_verdict_ring);
    for (auto& t : v->timers) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto& p : v->event_pool) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    v->models.clear();
    trace_destroy(v->trace);
    if (v->h_verdict) (void)hipHostFree(v->h_verdict);
    if (v->own_stream) (void)hipStreamDestroy(v->stream);
    if (!v->parent) g_viewers_on_device[v->device & 63].fetch_sub(1);
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
    o->host_verify = 0;
    o->frames_in_flight = 1;
    o->slab_shading = 1;
    if (const char* e = getenv("GSX_SLAB_SHADING")) o->slab_shading = atoi(e) != 0 ? 1u : 0u;   // (A/B: tools/ab_env.sh)
}

void gsx_debug_set_radix_rank_mode(int32_t mode) { radix_set_rank_override(mode); }

gsx_status gsx_viewer_set_render_options(gsx_viewer* v, const gsx_render_options* o) {
    if (!v || !o) return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: null argument");
    if (o->first_slab_divisor == 0 || o->growth < 2 || o->min_slab == 0)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: first_slab_divisor >= 1, growth >= 2, min_slab >= 1");
    if (!(o->spec_margin >= 0.0f) || o->spec_radius > 16 || o->host_verify > 2)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: spec_margin >= 0, spec_radius <= 16, host_verify 0 | 1 | 2");
    if (o->frames_in_flight < 1 || o->frames_in_flight > 4)
        return fail(GSX_ERR_INVALID_ARG, "gsx_viewer_set_render_options: frames_in_flight 1 .. 4");
    {
        gsx_status bst = viewer_bind(v);  // frames in flight finish under the options they were enqueued with
        if (bst) return bst;
    }
    // what gsx_preprocess decided (speculated round, lazy shading) belongs to the options it saw: a model preprocessed under
    // other scheduling options must go through gsx_preprocess + gsx_sort again before it is rendered
    if (o->progressive != v->options.progressive || o->speculative != v->options.speculative || o->slab_shading != v->options.slab_shading)
        for (auto& kv : v->models) {
            kv.second->sorted = false;
            kv.second->spec_round1 = false;
        }
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
    m->serial = g_model_serial.fetch_add(1) + 1;
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
        HIPCHK(m->sh_aos.ensure(16 * n * 16));
    } else if (sh == GSX_SH_HALF) {
        HIPCHK(m->sh_h.ensure(16 * n * 6));
        HIPCHK(m->sh_aos.ensure(16 * n * (cov3d == GSX_COV3D_SINGLE ? 12 : 8)));
    } else if (sh == GSX_SH_NORM8) {
        HIPCHK(m->sh_q.ensure(16 * n * 3));
        HIPCHK(m->sh_aos.ensure(16 * n * 8));
    }
    HIPCHK(m->mask.ensure(4 * ((n + 31) / 32)));
    {
        gsx_status rst = ensure_record_capacity(m.get(), n);
        if (rst) return rst;
        if ((rst = ensure_sortbin_capacity(m.get(), n))) return rst;
    }
    HIPCHK(m->counters.ensure(sizeof(Counters)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&m->h_counters), sizeof(Counters), hipHostMallocDefault));
    HIPCHK(gsx::op::MemsetAsync(m->counters.p, 0, sizeof(Counters), v->stream));
    // a fresh model is all-zero Gaussians (new_empty) and fully unmasked (MaskOpTree::Reset, scene.rs:2124-2131)
    for (DevBuf* b : {&m->pc, &m->cov_a, &m->cov_b, &m->cov_h, &m->cov_h2, &m->sh4, &m->sh1, &m->sh_h, &m->sh_q, &m->sh_aos})
        if (b->p) HIPCHK(gsx::op::MemsetAsync(b->p, 0, b->bytes, v->stream));
    v->models[key] = std::move(m);
    return GSX_OK;
}

gsx_status gsx_model_remove(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_remove: no model '%s'", key ? key : "(null)");
    HIPCHK(gsx::op::StreamSynchronize(v->stream));  // (ordered after the lanes' frames by viewer_bind)
    for (gsx_viewer* l : v->lanes) l->models.erase(key);
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
        HIPCHK(gsx::op::MemcpyAsync(v->staging.p, src + off, sizeof(gsx_gaussian) * c, hipMemcpyHostToDevice, v->stream));
        HIPCHK(launch_convert(v->stream, v->staging.as<gsx_gaussian>(), c, start + off, m->n, m->pod()));
        HIPCHK(gsx::op::StreamSynchronize(v->stream));  // the caller's memory may be reused after return
    }
    m->tuner.reset();
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
    m->tuner.reset();
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
    m->edit_epoch += 1;
    if (!words) {  // MaskOpTree::Reset
        if (m->has_mask) m->tuner.reset();
        m->has_mask = false;
        m->mask_program_hash = 0;
        return GSX_OK;
    }
    if (n_words != (m->n + 31) / 32) return fail(GSX_ERR_INVALID_ARG, "gsx_model_upload_mask: expected %llu words", (unsigned long long)((m->n + 31) / 32));
    HIPCHK(gsx::op::MemcpyAsync(m->mask.p, words, 4 * n_words, hipMemcpyHostToDevice, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    m->has_mask = true;
    m->mask_program_hash = 0;
    m->tuner.reset();
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
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(words, m->mask.p, 4 * n_words, hipMemcpyDeviceToHost));
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
    m->edit_epoch += 1;
    if (n_ops == 0) {  // MaskOpTree::Reset
        if (m->has_mask) m->tuner.reset();
        m->has_mask = false;
        m->mask_program_hash = 0;
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
        for (int c = 0; c < 3; ++c) prog.shapes[s].box_lim[c] = mask_box_limit(shapes[s].scale[c]);
    }
    memcpy(prog.ops, ops, sizeof(gsx_mask_op) * n_ops);
    HIPCHK(launch_mask_evaluate(v->stream, m->pc.as<float4>(), (uint32_t)m->n, prog, m->mask.as<uint32_t>()));
    // another mask is another scene as far as the speculation tuner's timings go (an app that re-evaluates the same
    // program every frame keeps them)
    uint64_t h = 1469598103934665603ull;
    for (size_t b = 0; b < sizeof prog; ++b) h = (h ^ reinterpret_cast<const unsigned char*>(&prog)[b]) * 1099511628211ull;
    h |= 1;
    if (!m->has_mask || h != m->mask_program_hash) m->tuner.reset();
    m->mask_program_hash = h;
    m->has_mask = true;
    return GSX_OK;
}

gsx_status gsx_preprocess(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_preprocess: no model '%s'", key ? key : "(null)");
    v->latest = nullptr;
    TraceScope trace(v, TRACE_PREPROCESS);
    if ((st = do_preprocess(v, m))) return st;
    return trace.finish();
}

gsx_status gsx_sort(gsx_viewer* v, const char* key) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_sort: no model '%s'", key ? key : "(null)");
    TraceScope trace(v, TRACE_SORT);
    if ((st = do_sort(v, m))) return st;
    return trace.finish();
}

gsx_status gsx_sync(gsx_viewer* v) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    for (gsx_viewer* l : v->lanes) {
        if ((st = finish_frame(l))) return st;
        HIPCHK(gsx::op::StreamSynchronize(l->stream));
    }
    gsx_status fst = finish_frame(v);
    if (fst) return fst;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    return GSX_OK;
}

gsx_status gsx_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v->latest = nullptr;
    TraceScope trace(v, TRACE_RENDER);
    if ((st = do_render(v, keys, n_keys))) return st;
    v->host_waited = false;   // (until the app waits again)
    return trace.finish();
}

gsx_status gsx_render_frame(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "viewer is null");
    gsx_status st = GSX_OK;
    gsx_viewer* lane = v;
    if (frame_may_overlap(v, keys, n_keys)) {
        HIPCHK(hipSetDevice(v->device));  // NOT viewer_bind: frames in flight stay in flight
        if ((st = prepare_edits_for_lanes(v, keys, n_keys))) return st;
        const uint32_t turn = v->lane_turn++ % v->options.frames_in_flight;
        if ((st = lane_acquire(v, turn, keys, n_keys, &lane))) return st;
    } else if ((st = viewer_bind(v))) {
        return st;
    }
    {
        TraceScope trace(lane, TRACE_RENDER_FRAME);  // the frame's launches leave as (cached, patched) HIP graphs when the scope ends
        for (uint32_t i = 0; i < n_keys; ++i) {
            Model* m = find_model(lane, keys ? keys[i] : nullptr);
            if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_render_frame: no model '%s'", keys && keys[i] ? keys[i] : "(null)");
            if ((st = do_preprocess(lane, m, true))) return st;  // the sort's admission scan sums N_vis: one launch less
            if ((st = do_sort(lane, m))) return st;
        }
        if ((st = do_render(lane, keys, n_keys))) return st;
        v->host_waited = false;   // (until the app waits again: gsx_sync, a blocking readback)
        if ((st = trace.finish())) return st;
    }
    if (lane != v) {
        HIPCHK(gsx::op::EventRecord(lane->lane_event, lane->stream));
        lane->lane_busy = true;
    }
    lane->held_w = lane->width;
    lane->held_h = lane->height;
    v->latest = lane == v ? nullptr : lane;
    return GSX_OK;
}

gsx_status gsx_download_framebuffer(gsx_viewer* v, float* rgbt, uint64_t n_floats) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v = result_lane(v);  // the newest frame may be a lane's
    const uint64_t need = 4ull * v->width * v->height;
    if (!rgbt || n_floats != need) return fail(GSX_ERR_INVALID_ARG, "gsx_download_framebuffer: expected %llu floats", (unsigned long long)need);
    if ((st = ensure_fb(v))) return st;
    if ((st = finish_frame(v))) return st;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(rgbt, fb_ptr(v), sizeof(float) * need, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_debug_download_lane_framebuffer(gsx_viewer* v, uint32_t lane, float* rgbt, uint64_t n_floats) {
    if (!v || v->parent || lane > v->lanes.size()) return fail(GSX_ERR_INVALID_ARG, "gsx_debug_download_lane_framebuffer: no such lane");
    HIPCHK(hipSetDevice(v->device));
    gsx_viewer* l = lane == 0 ? v : v->lanes[lane - 1];
    // the viewport of the frame the lane HOLDS: the owner's own may have been changed by gsx_update_camera since that frame was enqueued
    const uint64_t need = l->held_w ? 4ull * l->held_w * l->held_h : 4ull * l->width * l->height;
    if (!rgbt || n_floats != need || !fb_ptr(l)) return fail(GSX_ERR_INVALID_ARG, "gsx_debug_download_lane_framebuffer: expected %llu floats of a lane that has rendered", (unsigned long long)need);
    HIPCHK(gsx::op::StreamSynchronize(l->stream));  // (NOT viewer_bind: the frames in flight stay as they are)
    HIPCHK(gsx::op::Memcpy(rgbt, fb_ptr(l), sizeof(float) * need, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_download_rgba8(gsx_viewer* v, const float bg[3], uint8_t* rgba, uint64_t n_bytes) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v = result_lane(v);  // the newest frame may be a lane's
    const uint64_t npx = (uint64_t)v->width * v->height;
    if (!bg || !rgba || n_bytes != 4 * npx) return fail(GSX_ERR_INVALID_ARG, "gsx_download_rgba8: expected %llu bytes", (unsigned long long)(4 * npx));
    if ((st = ensure_fb(v))) return st;
    if ((st = finish_frame(v))) return st;
    HIPCHK(v->scratch.ensure(4 * npx));
    HIPCHK(launch_resolve_rgba8(v->stream, fb_ptr(v), (uint32_t)npx, bg[0], bg[1], bg[2], v->scratch.as<uint32_t>()));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(rgba, v->scratch.p, 4 * npx, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_framebuffer_device_ptr(gsx_viewer* v, void** out_ptr, uint32_t* out_w, uint32_t* out_h) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v = result_lane(v);  // the newest frame may be a lane's
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
    v = result_lane_of(v, key);  // the newest frame may be a lane's
    Model* m = find_model(v, key);
    if (!m || !out) return fail(GSX_ERR_NOT_FOUND, "gsx_model_frame_stats: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_model_frame_stats: model '%s' not preprocessed this frame", key);
    if ((st = finish_frame(v))) return st;
    out->n_gaussians = m->n;
    out->n_visible = m->n_visible;
    out->n_tile_entries = m->binned ? m->n_entries : 0;  // entries actually binned by the last gsx_render
    out->n_sorted = m->n_sorted;
    const bool spec_local = m->binned && m->spec_round1;
    // (an index-sharded frame whose exchange was windowed by last frame's limits is speculated in the same sense)
    out->speculated = spec_local || (m->binned && m->use_imported && m->shard_frame_limited) ? 1u : 0u;
    out->n_repair_tiles = spec_local ? m->h_counters->spec_need : (m->use_imported ? m->h_counters->shard_need : 0);
    out->n_repair_sorted = spec_local ? m->n_sorted2 : 0;
    out->overflow_slabs = (uint32_t)std::min<uint64_t>(m->overflow_slabs, 0xFFFFFFFFull);
    return GSX_OK;
}

gsx_status gsx_model_download_projection(gsx_viewer* v, const char* key, uint32_t* depth_key, uint32_t* rect, float* mean2d,
                                         float* conic_opacity, float* rgb) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v = result_lane_of(v, key);  // the newest frame may be a lane's
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_projection: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_projection: model '%s' not preprocessed", key);
    if ((st = complete_records(v, m))) return st;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    const size_t n = m->rec_n;  // == model length unless records were imported (gsx_shard_import)
    std::vector<uint32_t> k(n);
    std::vector<float4> a(n), b(n), c(n);
    const Records rr = m->rec();
    if (n) {
        HIPCHK(gsx::op::Memcpy(k.data(), rr.key, 4 * n, hipMemcpyDeviceToHost));
        HIPCHK(gsx::op::Memcpy(a.data(), rr.a, 16 * n, hipMemcpyDeviceToHost));
        HIPCHK(gsx::op::Memcpy(b.data(), rr.b, 16 * n, hipMemcpyDeviceToHost));
        HIPCHK(gsx::op::Memcpy(c.data(), rr.c, 16 * n, hipMemcpyDeviceToHost));
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
    v = result_lane_of(v, key);  // the newest frame may be a lane's
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_sorted: no model '%s'", key ? key : "(null)");
    if (!m->sorted) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_sorted: model '%s' not sorted", key);
    if ((st = sync_counters(v))) return st;
    const uint32_t n_order = m->spec_round1 && m->binned ? m->n_sorted2 : m->n_sorted;  // a speculated frame leaves its repair order
    if (out_n_visible) *out_n_visible = n_order;
    if (indices) {
        if (capacity < n_order) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_sorted: capacity %llu < %u sorted records", (unsigned long long)capacity, n_order);
        HIPCHK(gsx::op::StreamSynchronize(v->stream));
        if (n_order) HIPCHK(gsx::op::Memcpy(indices, m->sorted_idx, 4ull * n_order, hipMemcpyDeviceToHost));
    }
    return GSX_OK;
}

gsx_status gsx_model_download_tile_lists(gsx_viewer* v, const char* key, uint32_t* tile_offsets, uint64_t n_offsets,
                                         uint32_t* list, uint64_t capacity) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    v = result_lane_of(v, key);  // the newest frame may be a lane's
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_model_download_tile_lists: no model '%s'", key ? key : "(null)");
    if (!m->binned) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: model '%s' not rendered this frame", key);
    if (!m->lists_complete)
        return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: complete lists exist only for the front-most model "
                    "rendered with gsx_render_options.progressive = 0 (model '%s' was rendered in depth slabs)", key);
    if ((st = finish_frame(v))) return st;
    const uint32_t n_tiles = m->fc.tiles_x * m->fc.tiles_y;
    if (n_offsets != (uint64_t)n_tiles + 1) return fail(GSX_ERR_INVALID_ARG, "gsx_model_download_tile_lists: expected %u offsets", n_tiles + 1);
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    std::vector<uint2> r(n_tiles);
    HIPCHK(gsx::op::Memcpy(r.data(), m->ranges.p, sizeof(uint2) * n_tiles, hipMemcpyDeviceToHost));
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
        if (m->n_entries) HIPCHK(gsx::op::Memcpy(list, m->tile_list, 4ull * m->n_entries, hipMemcpyDeviceToHost));
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
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    HIPCHK(gsx::op::Memcpy(pos, dpos.p, 12 * n, hipMemcpyDeviceToHost));
    HIPCHK(gsx::op::Memcpy(color, dcol.p, 4 * n, hipMemcpyDeviceToHost));
    HIPCHK(gsx::op::Memcpy(cov3d, dcov.p, 24 * n, hipMemcpyDeviceToHost));
    if (sh) HIPCHK(gsx::op::Memcpy(sh, dsh.p, 180 * n, hipMemcpyDeviceToHost));
    return GSX_OK;
}

gsx_status gsx_set_pass_timing(gsx_viewer* v, uint32_t enabled) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "gsx_set_pass_timing: viewer is null");
    v->timing = enabled == 1u ? 0xFFFFFFFFu : (enabled >> 1);  // 1 = every pass; otherwise bit (p + 1) selects pass p
    return GSX_OK;
}

gsx_status gsx_get_pass_timing(gsx_viewer* v, float ms[GSX_PASS_COUNT], uint32_t launches[GSX_PASS_COUNT]) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    std::vector<gsx_viewer*> all{v};
    all.insert(all.end(), v->lanes.begin(), v->lanes.end());
    for (gsx_viewer* l : all) {
        for (auto& t : l->timers) {
            float e = 0.0f;
            if (hipEventElapsedTime(&e, t.start, t.stop) == hipSuccess) v->pass_ms[t.pass] += e;
            l->event_pool.push_back({t.start, t.stop});
        }
        l->timers.clear();
        if (l != v)
            for (int i = 0; i < GSX_PASS_COUNT; ++i) {
                v->pass_launches[i] += l->pass_launches[i];
                l->pass_launches[i] = 0;
            }
    }
    for (int i = 0; i < GSX_PASS_COUNT; ++i) {
        if (ms) ms[i] = v->pass_ms[i];
        if (launches) launches[i] = v->pass_launches[i];
        v->pass_ms[i] = 0.0f;
        v->pass_launches[i] = 0;
    }
    return GSX_OK;
}

}  // extern "C"
