// gsx_frame.cpp — frame scheduling of libgsx.so: what gsx_preprocess / gsx_sort / gsx_render enqueue on the viewer's stream.
//
// Mirrors the per-frame protocol the app drives (src/tab/scene.rs:699-874 and 2263-2326): per visible model preprocess (K1)
// + radix sort (K2), then the far -> near render loop (K3).  Nothing here waits for the device: counts stay in HBM, the host
// plans upper bounds, overflow and statistics are looked at lazily (finish_frame).
#include "gsx_state.h"
#include <sched.h>
#include <chrono>
#include <thread>
#include <cstdlib>

namespace gsx {

thread_local std::string g_err;

gsx_status fail(gsx_status st, const char* fmt, ...) {
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


// host_verify: spin on the pinned verdict word until k_spec_verify of this frame has posted {seq, need}.  The wait is
// bounded by the stream itself: if the stream drains (or fails) and the word still is not there, something upstream
// went wrong and that is reported instead of spinning forever.
// GSX_SPEC_DEBUG=1 / 2 / 3: tuner, per-frame, per-slab lines on stderr (read once)
static int spec_debug_level() {
    static const int level = getenv("GSX_SPEC_DEBUG") ? std::max(1, atoi(getenv("GSX_SPEC_DEBUG"))) : 0;
    return level;
}

static gsx_status wait_verdict(gsx_viewer* v, uint32_t seq, uint32_t* need) {
    trace_flush();  // the kernel that posts the word may still be in a recorded segment
    for (uint64_t spin = 1;; ++spin) {
        const unsigned long long w = __atomic_load_n(v->h_verdict, __ATOMIC_ACQUIRE);
        if ((uint32_t)(w >> 32) == seq) {
            *need = (uint32_t)w;
            return GSX_OK;
        }
        if ((spin & 0xFFFu) == 0) {
            const hipError_t e = gsx::op::StreamQuery(v->stream);
            if (e == hipSuccess) {
                const unsigned long long w2 = __atomic_load_n(v->h_verdict, __ATOMIC_ACQUIRE);
                if ((uint32_t)(w2 >> 32) == seq) continue;
                return fail(GSX_ERR_HIP, "speculation verdict %u never arrived (stream idle)", seq);
            }
            if (e != hipErrorNotReady) return fail(GSX_ERR_HIP, "stream failed while waiting for the speculation verdict: %s", hipGetErrorString(e));
        }
        // (spinning, like the sharded frame's ring wait: a sleep's wake-up is tens of microseconds of timer slack, i.e. a bubble on the
        //  device behind every verdict; a wait that has outlasted any frame yields its core between looks)
        if (spin > 200000u) sched_yield();
        else __builtin_ia32_pause();
    }
}


// ---- does speculating pay on THIS scene, along THIS camera path?  Measured, not guessed. ----
// Temporal occlusion speculation wins when frames are coherent and the scene occludes (cfg4: 2x), and loses when most frames
// need the repair round anyway (cfg2: 1 M sparse Gaussians, 90 % of the frames repair; random camera poses).  Either path
// gives the same pixels, so the viewer simply times them: some frames are bracketed by a pair of HIP events (recorded on the
// stream, read back when they have completed; the host waits for none of them except a probe's, eight frames after it —
// kSettleWait), one running mean per mode, and a four-phase cycle per model:
//   SPEC (len_spec frames) -> PROBE_PLAIN (5 frames, unspeculated; the windows keep being updated) -> SETTLE (speculated
//   frames until the probe's timings have arrived) -> decide;   PLAIN -> PROBE_SPEC -> SETTLE -> decide likewise.
// A decision that confirms the current mode doubles its phase — quadruples it when the verdict is clear — (64 ... 2048
// frames: the probes then cost < 1 %), one that flips it starts over at 64.  By construction the result stays within a few per cent of the better of the two paths.
// chunks of 128 candidates: finer blocks above kWalkFineOn / coarse blocks again below kWalkFineOff (measured under the finer ones).
// Round 6: 220 / 70 (160 / 30 before).  cfg4's orbit walks 10-15 chunks a frame with rare poses of 130-170 under coarse blocks and 6-14
// with spikes of 50 under fine ones — either state held itself and the first spike chose: - 10 % with one frame in flight whenever the
// fine state won.  An open horizon walks 275-350 coarse, 104-146 fine (GSX_SPEC_DEBUG=3 prints the statistic).
constexpr uint32_t kWalkFineOn = 220, kWalkFineOff = 70;
constexpr uint32_t kSlabShadingMaxPercent = 35;  // slab shading pays while the slabs shade less than this share of the visible records (break-even ~45 % on cfg4)
constexpr uint32_t kBucketSortMax = 1500000;  // pairs: above, the LSD depth sort (256 buckets of 8192 pairs fit the LDS; at 1 M the bucket sort takes half the LSD sort's time)
constexpr uint32_t kProbeFrames = 5;   // the first is not timed (the switch itself is atypical), the other four are
constexpr uint32_t kSettleWait = 8;    // frames enqueued behind a probe before the host waits for its timings
constexpr uint32_t kSettleFrames = 64; // at most this many frames between a probe and the decision it feeds (normally: until its timings are in)

static void tuner_collect(Model* m) {
    SpecTuner& t = m->tuner_ref ? *m->tuner_ref : m->tuner;
    for (auto& s : t.slots) {
        if (s.state < 2 || hipEventQuery(s.stop) != hipSuccess) continue;
        if (s.state == 3) {  // bracketed before a reset()
            s.state = 0;
            continue;
        }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess && ms > 0.0f) {
            double& mean = s.spec ? t.mean_spec : t.mean_plain;
            uint32_t& n = s.spec ? t.n_spec : t.n_plain;
            mean = n == 0 ? ms : mean + 0.25 * (ms - mean);
            n += 1;
        }
        if (s.probe && t.probe_pending) t.probe_pending -= 1;
        s.state = 0;
    }
}

// called once per gsx_preprocess of a model that could speculate; returns whether this frame should
static bool tuner_wants_speculation(Model* m) {
    SpecTuner& t = m->tuner_ref ? *m->tuner_ref : m->tuner;
    tuner_collect(m);
    static const bool debug = spec_debug_level() >= 1;
    // a settle phase ends as soon as the probe's timings are in.  Eight frames after the probe the host stops running ahead
    // until they are: it waits for the probe's last event — with eight frames queued behind it the device never idles, and a
    // host that is dozens of short frames ahead (a 1 M-Gaussian scene on two lanes) would otherwise spend that long in the
    // mode it is about to leave
    const bool settling = t.phase == SpecTuner::SETTLE_SPEC || t.phase == SpecTuner::SETTLE_PLAIN;
    if (settling && t.probe_pending && kSettleFrames - t.left >= kSettleWait) {
        for (auto& s : t.slots)
            if (s.state == 2 && s.probe) (void)gsx::op::EventSynchronize(s.stop);
        tuner_collect(m);
    }
    if (settling && t.probe_pending == 0) t.left = 0;
    if (t.left == 0) {
        switch (t.phase) {
            case SpecTuner::SPEC:
                t.phase = SpecTuner::PROBE_PLAIN; t.left = kProbeFrames;
                break;
            case SpecTuner::PLAIN:
                t.phase = SpecTuner::PROBE_SPEC; t.left = kProbeFrames;
                break;
            // a host that does not wait for the device is several frames ahead of it: the probe's timings arrive while the
            // frames after it are being enqueued, so the decision is taken a dozen frames later, in the old mode meanwhile
            case SpecTuner::PROBE_PLAIN:
                t.phase = SpecTuner::SETTLE_SPEC; t.left = kSettleFrames;
                break;
            case SpecTuner::PROBE_SPEC:
                t.phase = SpecTuner::SETTLE_PLAIN; t.left = kSettleFrames;
                break;
            case SpecTuner::SETTLE_SPEC:
            case SpecTuner::SETTLE_PLAIN: {
                const bool was_spec = t.phase == SpecTuner::SETTLE_SPEC;
                const bool have = t.n_spec >= 2 && t.n_plain >= 2;  // (running means over every bracketed frame so far, newest weighted most)
                const bool spec_better = have ? (was_spec ? t.mean_spec <= 1.03 * t.mean_plain : t.mean_spec < 0.97 * t.mean_plain) : was_spec;
                if (debug) fprintf(stderr, "[gsx spec] model '%s' frame %u: speculated %.3f ms (%u samples), plain %.3f ms (%u) -> %s\n", m->key.c_str(),
                                   t.frame_no, t.mean_spec, t.n_spec, t.mean_plain, t.n_plain, spec_better ? "speculate" : "plain");
                // a clear verdict (the other path costs half as much again, or more) is asked for again four times later, a close
                // one twice later: on cfg4 a probe is five frames at twice the cost, on cfg2 the two paths are within 5 %
                const double ratio = !have ? 1.0 : (spec_better ? t.mean_plain / std::max(t.mean_spec, 1e-6) : t.mean_spec / std::max(t.mean_plain, 1e-6));
                const uint32_t grow = ratio >= 1.5 ? 4u : 2u;
                if (spec_better) {
                    t.len_spec = was_spec ? std::min<uint32_t>(grow * t.len_spec, 2048u) : 64u;
                    t.len_plain = 64;
                    t.phase = SpecTuner::SPEC; t.left = t.len_spec;
                } else {
                    t.len_plain = was_spec ? 64u : std::min<uint32_t>(grow * t.len_plain, 2048u);
                    t.len_spec = 64;
                    t.phase = SpecTuner::PLAIN; t.left = t.len_plain;
                }
                break;
            }
        }
    }
    t.left -= 1;
    t.frame_no += 1;
    return t.phase == SpecTuner::SPEC || t.phase == SpecTuner::PROBE_SPEC || t.phase == SpecTuner::SETTLE_SPEC;
}

// bracket this model's frame with events?  every frame of a probe but its first (the switch itself is atypical), every
// fourth frame otherwise (an event pair costs a few microseconds of stream gap)
static void tuner_frame_begin(gsx_viewer* v, Model* m, bool speculated) {
    SpecTuner& t = m->tuner_ref ? *m->tuner_ref : m->tuner;
    if (t.active) {  // the last bracket was never closed (a gsx_preprocess without its gsx_render): take the slot back
        if (t.active->state == 1) {
            t.active->state = 0;
            if (t.active->probe && t.probe_pending) t.probe_pending -= 1;
        }
        t.active = nullptr;
    }
    const bool probe = t.phase == SpecTuner::PROBE_PLAIN || t.phase == SpecTuner::PROBE_SPEC;
    if (probe ? t.left == kProbeFrames - 1 : (t.frame_no & 3u) != 0) return;
    for (auto& s : t.slots) {
        if (s.state != 0) continue;
        if (!s.start && (hipEventCreate(&s.start) != hipSuccess || hipEventCreate(&s.stop) != hipSuccess)) return;
        if (gsx::op::EventRecord(s.start, v->stream) != hipSuccess) return;
        s.spec = speculated;
        s.probe = probe;
        s.state = 1;
        if (probe) t.probe_pending += 1;
        t.active = &s;
        return;
    }
}

static void tuner_frame_end(gsx_viewer* v, Model* m) {
    SpecTuner& t = m->tuner_ref ? *m->tuner_ref : m->tuner;
    if (!t.active) return;
    t.active->state = gsx::op::EventRecord(t.active->stop, v->stream) == hipSuccess ? 2 : 0;
    if (t.active->state == 0 && t.active->probe && t.probe_pending) t.probe_pending -= 1;
    t.active = nullptr;
}

// The host's view of the device-side overflow bookkeeping (SlabStats::overflow_events / max_needed_ever never reset): when
// slabs spilled since the last look, the pair buffers grow for the frames to come.  The frames that spilled were composited
// completely on the device (k_composite_spill) — this is about speed, not correctness.
static void note_overflow(Model* m) {
    const uint32_t ev = m->h_counters->overflow_events;
    if (ev == m->overflow_seen) return;
    m->overflow_slabs += ev - m->overflow_seen;
    m->overflow_seen = ev;
    if (spec_debug_level() >= 1)
        fprintf(stderr, "[gsx overflow] model '%s': %u events so far, largest slab wanted %u entries, capacity %llu, n_sorted %u n_sorted2 %u entries_total %u speculated-copy %d\n",
                m->key.c_str(), ev, m->h_counters->max_needed_ever, (unsigned long long)m->tile_cap, m->h_counters->n_sorted, m->h_counters->n_sorted2,
                m->h_counters->n_entries_total, (int)m->stats_copy_speculated);
    m->tile_cap = std::max<uint64_t>(2 * m->tile_cap, (uint64_t)m->h_counters->max_needed_ever + 1024);
}

// Frames are enqueued without any host round trip; this is where the host catches up: wait for the
// stream and mirror the per-model statistics.
gsx_status finish_frame(gsx_viewer* v) {
    (v->parent ? v->parent : v)->host_waited = true;   // (every caller goes on to wait for the device: gsx_sync, the blocking readbacks)
    for (int attempt = 0; attempt < 8; ++attempt) {
        bool pending = false;
        for (auto& kv : v->models) pending |= kv.second->stats_pending;
        if (!pending) return GSX_OK;
        for (auto& kv : v->models) {
            Model* m = kv.second.get();
            if (m->stats_pending)
                HIPCHK(gsx::op::MemcpyAsync(m->h_counters, m->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, v->stream));
        }
        HIPCHK(gsx::op::StreamSynchronize(v->stream));
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
            if (m->spec_round1) {
                m->last_spec_sorted = m->h_counters->n_sorted;
                m->last_repair_sorted = m->h_counters->n_sorted2;
            }
            if (m->slab_shading && m->binned) m->slab_shading_off = (uint64_t)m->h_counters->n_shaded_total * 100u > (uint64_t)m->h_counters->n_visible * kSlabShadingMaxPercent;
            if (m->binned && !m->spec_round1) m->slabs_hint = m->h_counters->slabs_used;   // (a speculated frame is ONE slab: it says nothing about how many a plain frame needs)
            m->stats_copy_inflight = false;
            note_overflow(m);
            // The pixels of a frame that spilled are complete (k_composite_spill); only its tile LISTS are not, and only a
            // single-slab frame promises those (gsx_model_download_tile_lists): that one is redone with the grown buffers.
            if (m->h_counters->overflow && m->binned && m->lists_complete && !v->tile_cap_fixed && !v->last_render_cont && m->rec_n == m->n)
                redo = true;
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


// buffers sized by the model (projection outputs)
gsx_status ensure_record_capacity(Model* m, uint64_t count) {
    if (count <= m->rec_cap) return GSX_OK;
    const size_t n = std::max<uint64_t>(count, 1);
    HIPCHK(m->key_buf.ensure(4 * n));
    HIPCHK(m->rec_a.ensure(16 * n));
    HIPCHK(m->rec_b.ensure(16 * n));
    HIPCHK(m->rec_c.ensure(16 * n));
    HIPCHK(m->block_vis.ensure(4 * (project_blocks(n) + 4)));  // (k_admit_scan reads it in uint4 steps)
    m->rec_cap = n;
    return GSX_OK;
}

// buffers sized by the frame's active record set (depth sort + per-slab binning)
gsx_status ensure_sortbin_capacity(Model* m, uint64_t count) {
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
            HIPCHK(gsx::op::Memset(m->sort_ws.p, 0, m->sort_ws.bytes));  // status words must not alias a live epoch
        }
    }
    HIPCHK(m->srect.ensure(16 * n));  // uint2 tile rectangles (tile lists) or uint4 {rect, key, index} (block lists)
    // (cnt / block_sums — per-record counts and their chunk sums — belong to the three-launch binning: per-tile lists, GSX_BIN_FUSED=0;
    //  the slabs that take that path ask for them: ensure_count_buffers)
    m->sortbin_cap = n;
    return GSX_OK;
}

static gsx_status ensure_count_buffers(Model* m) {
    HIPCHK(m->cnt.ensure(4 * m->sortbin_cap));
    HIPCHK(m->block_sums.ensure(4 * (scan_blocks(m->sortbin_cap) + 1)));
    return GSX_OK;
}

gsx_status ensure_import_capacity(Model* m, uint64_t count) {
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

gsx_status ensure_selection(gsx_viewer* v, Model* m) {
    const size_t bytes = 4 * std::max<size_t>(((size_t)m->n + 31) / 32, 1);
    if (m->selection.bytes < bytes) {
        HIPCHK(m->selection.ensure(bytes));
        HIPCHK(gsx::op::MemsetAsync(m->selection.p, 0, bytes, v->stream));
    }
    return GSX_OK;
}

gsx_status ensure_edit_buffers(gsx_viewer* v, Model* m) {
    const size_t words = std::max<size_t>(((size_t)m->n + 31) / 32, 1), n = std::max<size_t>(m->n, 1);
    if (m->edited.bytes < 4 * words) {
        HIPCHK(m->edited.ensure(4 * words));
        HIPCHK(gsx::op::MemsetAsync(m->edited.p, 0, 4 * words, v->stream));
        HIPCHK(m->keep.ensure(4 * words));
        HIPCHK(m->edit_a.ensure(16 * n));
        HIPCHK(m->edit_b.ensure(16 * n));
        m->prep_epoch = 0;  // new buffers: nothing prepared
    }
    return GSX_OK;
}

// k_edit_prepare for this model if its inputs changed since it last ran (Model::edit_epoch; idempotent otherwise): persists the
// selection edit into the selected Gaussians' edit records and derives keep = mask & ~hidden.  *launched: it ran.
bool edits_need_prepare(const gsx_viewer* v, const Model* m) {
    const bool sel_edit_on = m->has_selection && (v->sel_edit.flag & GSX_EDIT_ENABLED);
    if (m->show_unedited || !(m->has_edits || sel_edit_on)) return false;  // no edit pass this frame
    const uint32_t* mask = m->has_mask ? m->mask.as<uint32_t>() : nullptr;
    const bool buffers = m->edited.bytes >= 4 * std::max<size_t>(((size_t)m->n + 31) / 32, 1);
    return !(v->edit_cache && buffers && m->prep_epoch == m->edit_epoch && m->prep_has_selection == m->has_selection && m->prep_mask == mask &&
             memcmp(&m->prep_sel_edit, &v->sel_edit, sizeof v->sel_edit) == 0);
}

gsx_status prepare_edits(gsx_viewer* v, Model* m, bool* launched) {
    if (launched) *launched = false;
    if (!edits_need_prepare(v, m)) return GSX_OK;
    gsx_status st = ensure_edit_buffers(v, m);
    if (st) return st;
    const uint32_t* mask = m->has_mask ? m->mask.as<uint32_t>() : nullptr;
    HIPCHK(launch_edit_prepare(v->stream, (uint32_t)m->n, m->has_selection ? m->selection.as<uint32_t>() : nullptr, m->edited.as<uint32_t>(),
                               m->edit_a.as<float4>(), m->edit_b.as<float4>(), v->sel_edit, mask, m->keep.as<uint32_t>()));
    m->prep_epoch = m->edit_epoch;
    m->prep_has_selection = m->has_selection;
    m->prep_mask = mask;
    m->prep_sel_edit = v->sel_edit;
    m->has_edits = true;
    if (launched) *launched = true;
    return GSX_OK;
}

static gsx_status ensure_msd(gsx_viewer* v, Model* m, DevBuf& ws);

gsx_status do_preprocess(gsx_viewer* v, Model* m, bool defer_visible_count) {
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
    m->rows_nominal = 0;
    // selection edit / stored edits / highlight: only when something of the kind exists (spec §7)
    const uint32_t n32 = (uint32_t)m->n;
    const size_t words = ((size_t)m->n + 31) / 32;
    const bool sel_edit_on = m->has_selection && (v->sel_edit.flag & GSX_EDIT_ENABLED);
    const bool edits_on = !m->show_unedited && (m->has_edits || sel_edit_on);
    const bool highlight_on = m->has_selection && v->highlight[3] > 0.0f;
    PodPlanes pod = m->pod();
    if (edits_on) {
        // (a lane's shadow model views the owner's edit buffers: the owner prepared them before the frame was dealt out, gsx_render_frame)
        if (!v->parent && (st = prepare_edits(v, m, nullptr))) return st;
        m->has_edits = true;
        pod.mask = m->keep.as<uint32_t>();
    }
    // admission is decided inside the projection kernel: every visible Gaussian, or — when this model has windows from
    // its previous frame — the conservative max-pyramid test of the temporal occlusion speculation
    m->spec_round1 = v->options.progressive && v->options.speculative && m->spec_valid && m->spec_tiles_x == m->fc.tiles_x &&
                     m->spec_tiles_y == m->fc.tiles_y;
    // ... and whether speculating pays here is measured (SpecTuner): plain frames while it does not, windows kept up to date
    const bool could_speculate = v->options.progressive && v->options.speculative && !(m->shard_win_set && m->shard_tiles_x == m->fc.tiles_x && m->shard_tiles_y == m->fc.tiles_y);
    if (could_speculate && !tuner_wants_speculation(m)) m->spec_round1 = false;
    // deep inside a plain phase nobody reads the windows this frame would leave behind (its last frame does: a probe follows)
    {
        // (with frames in flight every lane needs ITS windows for the probe: the last L frames of the phase keep them)
        const SpecTuner& tn = m->tuner_ref ? *m->tuner_ref : m->tuner;
        const uint32_t lanes = v->parent ? v->parent->options.frames_in_flight : v->options.frames_in_flight;
        m->windows_unwanted = could_speculate && tn.phase == SpecTuner::PLAIN && tn.left >= lanes;
    }
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
    // (edits and the highlight are colour ops on shaded records: shade_admitted applies them to what k_shade writes; a rect /
    // brush / texture query is answered by the geometry-only kernel itself from the projected centre; a hit query reads conics)
    const bool geometric_query = v->query.kind == GSX_QUERY_RECT || v->query.kind == GSX_QUERY_BRUSH || v->query.kind == GSX_QUERY_TEXTURE;
    // slab shading (gsx_render_options): a progressive frame without windows projects geometry only as well; its depth slabs then shade
    // exactly the records some block of tiles still takes (k_block_bin's list).  Only frames that bin by blocks, slab by slab:
    // a model small enough for ONE slab keeps complete per-tile lists and its full records (gsx_model_download_tile_lists).
    m->slab_shading = v->options.slab_shading && v->options.progressive && v->bin_mode == 1 && v->bin_fused && !m->spec_round1 && !shard_lazy &&
                      m->n > v->options.min_slab && m->pod().sh_aos != nullptr;
    // ... and only while it pays: a scene where next to nothing saturates (translucent) has every visible record taken by some block — then
    // the streaming projection of everything (k_project: 0.67 of HBM peak) beats gathering the same records slab by slab.  Measured, lazily:
    // the last slab-shaded frame's count (Counters::n_shaded_total); tried again every 256th plain frame.
    if (m->slab_shading && m->slab_shading_off) {
        if (++m->slab_shading_retry < 256u) m->slab_shading = false;
        else m->slab_shading_retry = 0, m->slab_shading_off = false;
    }
    m->lazy = (m->spec_round1 || shard_lazy || m->slab_shading) && (v->query.kind == GSX_QUERY_NONE || geometric_query);
    if (!m->lazy) m->slab_shading = false;
    {
        static const bool debug = spec_debug_level() >= 2;
        if (debug) fprintf(stderr, "[gsx frame] model '%s': speculated %d, slab shading %d (off %d), lazy %d, n %llu\n", m->key.c_str(), (int)m->spec_round1,
                           (int)m->slab_shading, (int)m->slab_shading_off, (int)m->lazy, (unsigned long long)m->n);
    }
    if (m->lazy && geometric_query) {
        HIPCHK(m->query_flags.ensure(4 * std::max<size_t>(words, 1)));
        if (v->query.kind == GSX_QUERY_TEXTURE && (v->query_tex_w != v->width || v->query_tex_h != v->height))
            return fail(GSX_ERR_INVALID_ARG, "gsx_preprocess: texture query without a viewport-sized query texture (gsx_update_query_texture)");
        adm.query = ProjectQuery{v->query, v->query_texture.as<uint8_t>(), v->query_tex_w, v->query_tex_h, m->query_flags.as<uint32_t>()};
    }
    m->frame_edits = edits_on;
    m->frame_highlight = highlight_on;
    m->cand_valid = false;
    adm.lazy = m->lazy ? 1u : 0u;
    // a lazy projection writes four bytes of tile rectangle per Gaussian instead of the 16-byte `a` record (grids up to 255 x 255 tiles)
    // (Round 5 measured the unlazy kernel writing this 4-byte plane as well, so that the binning of an unspeculated frame could gather its
    //  rectangles from a 40 MB plane instead of the 160 MB `a` plane: binning 200 -> 195-200 us, projection + 2-8 us: nothing.  A gather by
    //  depth order touches one line per record whatever the record's size.  profiles/r05_ab_rect8.txt)
    m->rect8_active = m->lazy && m->fc.tiles_x <= 255u && m->fc.tiles_y <= 255u;
    if (m->rect8_active) HIPCHK(m->rect8.ensure(4 * std::max<size_t>(m->n, 1)));
    // slab shading: a byte a Gaussian, the coarse cells of its rectangle — the depth sort carries them into depth order (do_sort)
    m->code8_active = m->slab_shading && m->rect8_active;
    if (m->code8_active) HIPCHK(m->code8.ensure(std::max<size_t>(m->n, 1)));
    m->last_pod_mask = pod.mask;
    m->last_pyramid = adm.pyramid.data;
    if (could_speculate && !m->use_imported) tuner_frame_begin(v, m, m->spec_round1);
    {
        const int pass = m->lazy ? GSX_PASS_PROJECT_GEOM : GSX_PASS_PROJECT;  // two kernels, two averages
        ScopedPass t(v, pass);  // brackets the projection kernel alone (bench.py's roofline kernel)
        HIPCHK(launch_project(v->stream, m->fc, n32, pod, m->proj_rec(), m->block_vis.as<uint32_t>(), adm));
        v->pass_launches[pass] += m->n ? 1 : 0;
    }
    // N_vis: summed by the admission scan when a compaction follows anyway (gsx_render_frame, a lazily projected shard)
    m->visible_count_pending = defer_visible_count || shard_lazy;
    if (!m->visible_count_pending) HIPCHK(launch_sum_counts(v->stream, m->block_vis.as<uint32_t>(), n32, &m->counters.as<Counters>()->n_visible));
    if (shard_lazy) {
        // the candidates of the coming exchange (a conservative superset of the travellers): compact them and give
        // exactly those their conic / colour records; gsx_shard_pack then looks at nothing else
        Counters* dcx = m->counters.as<Counters>();
        HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(m->n, 1)));
        if (v->bucket_sort) {  // one launch (look-back over 65536-Gaussian tiles) instead of scan + scatter
            if ((st = ensure_msd(v, m, m->msd_ws))) return st;
            HIPCHK(launch_admit_compact(v->stream, m->proj_rec().key, n32, m->adm_ballots.as<unsigned long long>(), &dcx->n_candidates, m->adm_pairs.as<uint2>(),
                                        m->block_vis.as<uint32_t>(), &dcx->n_visible, m->msd_ws.as<uint32_t>(), 0, nullptr, false));
        } else {
            HIPCHK(m->adm_offsets.ensure(m->adm_counts.bytes));
            HIPCHK(launch_admit_from_project(v->stream, m->proj_rec().key, n32, m->adm_ballots.as<unsigned long long>(),
                                             m->adm_counts.as<uint32_t>(), m->adm_offsets.as<uint32_t>(), &dcx->n_candidates,
                                             m->adm_pairs.as<uint2>(), true, m->block_vis.as<uint32_t>(), &dcx->n_visible));
        }
        m->visible_count_pending = false;
        if (m->lazy && (st = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dcx->n_candidates, nullptr, m->rect8_active}))) return st;
        m->cand_valid = true;
    }
    if ((edits_on || highlight_on) && !m->lazy)
        HIPCHK(launch_edit_apply(v->stream, n32, m->proj_rec(), highlight_on ? m->selection.as<uint32_t>() : nullptr,
                                 edits_on ? m->edited.as<uint32_t>() : nullptr, m->edit_a.as<float4>(), m->edit_b.as<float4>(),
                                 v->highlight));
    m->flags_kind = GSX_QUERY_NONE;
    if (adm.query.flags) {  // answered by the projection kernel
        m->flags_kind = v->query.kind;
        m->flags_op = v->query.selection_op;
    } else if (v->query.kind != GSX_QUERY_NONE) {
        if (v->query.kind == GSX_QUERY_HIT) {
            HIPCHK(m->hits.ensure(sizeof(gsx_query_hit) * (size_t)GSX_QUERY_MAX_HITS));
            HIPCHK(m->hit_count.ensure(4));
            HIPCHK(gsx::op::MemsetAsync(m->hit_count.p, 0, 4, v->stream));
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
gsx_status complete_records(gsx_viewer* v, Model* m) {
    if (!m->lazy || !m->preprocessed) return GSX_OK;
    ProjectAdmission adm{};
    adm.pyramid = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->last_pyramid);
    adm.ballots = m->adm_ballots.as<unsigned long long>();
    HIPCHK(m->block_sums.ensure(4 * std::max<size_t>((m->n + 255) / 256, 1)));
    adm.block_counts = m->block_sums.as<uint32_t>();  // scratch: the admission counts were consumed by the compaction
    PodPlanes pod = m->pod();
    pod.mask = m->last_pod_mask;
    m->rect8_active = false;  // the unlazy kernel writes every record whole
    m->code8_active = false;
    HIPCHK(launch_project(v->stream, m->fc, (uint32_t)m->n, pod, m->proj_rec(), m->block_vis.as<uint32_t>(), adm));
    m->lazy = false;
    if (m->frame_edits || m->frame_highlight)  // every record was written again: the frame's colour ops on all of them
        HIPCHK(launch_edit_apply(v->stream, (uint32_t)m->n, m->proj_rec(), m->frame_highlight ? m->selection.as<uint32_t>() : nullptr,
                                 m->frame_edits ? m->edited.as<uint32_t>() : nullptr, m->edit_a.as<float4>(), m->edit_b.as<float4>(),
                                 v->highlight));
    return GSX_OK;
}

// the bucket sort's workspace of this model (main round / repair round): allocated and initialised on first use
static gsx_status ensure_msd(gsx_viewer* v, Model* m, DevBuf& ws) {
    const size_t words = msd_workspace_words(std::max<uint64_t>(m->rec_cap, std::max<uint64_t>(m->n, 1)));
    if (ws.bytes >= 4 * words) return GSX_OK;
    HIPCHK(ws.ensure(4 * words));
    HIPCHK(msd_workspace_init(v->stream, ws.as<uint32_t>(), ws.bytes / 4));
    return GSX_OK;
}

gsx_status shade_admitted(gsx_viewer* v, Model* m, const LateProjection& late) {
    ScopedPass t(v, GSX_PASS_SHADE);   // (suspends the depth sort's / the binning scope it is called from)
    v->pass_launches[GSX_PASS_SHADE] += (m->frame_edits || m->frame_highlight) ? 2 : 1;
    PodPlanes pod = m->pod();
    pod.mask = m->last_pod_mask;
    const uint32_t n = (uint32_t)m->n;
    HIPCHK(launch_shade(v->stream, m->fc, n, pod, m->proj_rec(), late));
    if (m->frame_edits || m->frame_highlight)
        HIPCHK(launch_edit_apply_list(v->stream, n, m->proj_rec(), late.pairs, late.d_n, late.shaded,
                                      m->frame_highlight ? m->selection.as<uint32_t>() : nullptr,
                                      m->frame_edits ? m->edited.as<uint32_t>() : nullptr, m->edit_a.as<float4>(), m->edit_b.as<float4>(),
                                      v->highlight));
    return GSX_OK;
}

// (A single-launch depth sort for the ~0.3 M pairs of a speculated frame — one persistent grid, device-wide barriers between
// the digit passes — was built and measured this round: 146 us against 66 us for one launch per digit at 0.3 M pairs, 499
// against 95 at 1 M.  Seven grid barriers plus the per-tile offset lookups cost more than the five kernel boundaries they
// replace, and a persistent grid that needs every workgroup resident is a deadlock risk next to other streams.  Not shipped.)

// force_full: ignore the admission the projection pass made (a speculated frame being redone) and sort every visible record
gsx_status do_sort(gsx_viewer* v, Model* m, bool force_full) {
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_sort('%s') before gsx_preprocess", m->key.c_str());
    const uint32_t n = (uint32_t)m->rec_n;
    Counters* dc = m->counters.as<Counters>();
    uint32_t launches_sort = 4;
    m->sorted_code_valid = false;
    {
        ScopedPass t(v, GSX_PASS_DEPTH_SORT);
        if (m->use_imported) {  // every imported record is visible: sort the keys as they lie
            m->spec_round1 = false;
            RadixBuffers rb{m->rec().key, nullptr, nullptr, m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                            m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
            if (v->bucket_sort && m->has_window) {  // a windowed exchange: a band's share of the admitted records (tens of thousands): three launches, not five
                gsx_status mst = ensure_msd(v, m, m->msd_ws);
                if (mst) return mst;
                HIPCHK(launch_bucket_sort(v->stream, rb, n, &dc->n_sorted, true, m->msd_ws.as<uint32_t>(), m->msd_seq++, false));
                launches_sort = 3;
            } else {
                HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted, 32, true));  // n: an upper bound (slot import); the count is on the device
            }
        } else {
            // speculated frames: compact the (key, index) pairs the projection pass admitted, then sort only those
            // (counting the sort's digit histograms inside the compaction / tile-emit kernels — LDS atomics where the pairs are
            // written, one flush per workgroup — removes two histogram launches per frame and was measured: 1200 vs 1198 fps
            // speculated, 602 vs 613 unspeculated on cfg4; the counting costs what the histogram kernels cost.  Not kept.)
            if (force_full) {
                gsx_status stc = complete_records(v, m);
                if (stc) return stc;
                m->spec_round1 = false;
            }
            // every visible record enters the sort (an unspeculated frame, a redone one): the first radix pass reads the
            // projection's key plane as it lies and skips the culled records — no compaction pass in front of the sort
            // (k_admit_scan + k_admit_scatter_dense: 55 us and 176 MB at 10 M Gaussians)
            const bool dense = force_full || (!m->spec_round1 && (!m->lazy || m->slab_shading) && m->last_pyramid == nullptr);
            if (dense) {
                if (m->visible_count_pending) {
                    HIPCHK(launch_sum_counts(v->stream, m->block_vis.as<uint32_t>(), n, &dc->n_visible));
                    m->visible_count_pending = false;
                }
                RadixBuffers rb{m->proj_rec().key, nullptr, nullptr, m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                                m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
                const bool codes = m->code8_active && !force_full && m->n <= (1u << 24);   // (the code travels in the top byte of the sort's values)
                if (codes) HIPCHK(m->sorted_code.ensure(std::max<size_t>(m->sortbin_cap, 1)));
                HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted, 32, true, true, nullptr, nullptr, nullptr, false, codes ? m->code8.as<uint8_t>() : nullptr,
                                         codes ? m->sorted_code.as<uint8_t>() : nullptr));
                m->sorted_code_valid = codes;
            } else {
                HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(n, 1)));
                // a speculated frame admits a few per cent of the visible records: ONE compaction launch that also counts the bucket
                // sort's histogram, then a partition pass and a launch of in-LDS bucket sorts (gsx_internal.h "bucket sort") — four
                // launches for what scan + scatter + histogram + four digit passes did in seven
                // (a frame that admits millions — stale windows after a camera jump, poses in random order — sorts faster in four full
                //  digit passes: the bucket sort's buckets then outgrow the LDS.  The host only knows the count of an earlier frame: good enough)
                const bool bucket = v->bucket_sort && m->spec_round1 && !(m->last_spec_sorted > kBucketSortMax);
                uint32_t seq = 0;
                if (bucket) {
                    gsx_status mst = ensure_msd(v, m, m->msd_ws);
                    if (mst) return mst;
                    seq = m->msd_seq++;
                    HIPCHK(launch_admit_compact(v->stream, m->proj_rec().key, n, m->adm_ballots.as<unsigned long long>(), &dc->n_sorted, m->adm_pairs.as<uint2>(),
                                                m->visible_count_pending ? m->block_vis.as<uint32_t>() : nullptr, &dc->n_visible, m->msd_ws.as<uint32_t>(), seq));
                } else {
                    HIPCHK(m->adm_offsets.ensure(m->adm_counts.bytes));
                    HIPCHK(launch_admit_from_project(v->stream, m->proj_rec().key, n, m->adm_ballots.as<unsigned long long>(),
                                                     m->adm_counts.as<uint32_t>(), m->adm_offsets.as<uint32_t>(), &dc->n_sorted,
                                                     m->adm_pairs.as<uint2>(), m->spec_round1,
                                                     m->visible_count_pending ? m->block_vis.as<uint32_t>() : nullptr, &dc->n_visible));
                }
                m->visible_count_pending = false;
                if (m->lazy) {  // the projection pass was geometry only: shade what it admitted
                    gsx_status sst = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted, nullptr, m->rect8_active});
                    if (sst) return sst;
                }
                RadixBuffers rb{nullptr, nullptr, m->adm_pairs.as<uint2>(), m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                                m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
                if (bucket) {
                    HIPCHK(launch_bucket_sort(v->stream, rb, n, &dc->n_sorted, false, m->msd_ws.as<uint32_t>(), seq, true));
                    launches_sort = 2;
                } else {
                    HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted, 32, false));
                }
            }
        }
        m->sorted_idx = m->sv_out.as<uint32_t>();
        v->pass_launches[GSX_PASS_DEPTH_SORT] += n ? launches_sort : 0;
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
// (round 6: TWO slabs behind the kept ones — the next as planned, then the rest.  With everything behind in one slab, a pose at which the
//  tiles do not saturate where the last plain frame's did handed that one slab every record: 12 M entries where the plan makes 1.5 M.)
static void merge_tail_slabs(std::vector<uint32_t>* bounds, uint32_t used) {
    if (used == 0) return;
    const size_t keep = (size_t)used + 2;  // slabs kept as planned: the ones that found work, and one more
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
// frame_zero (nullable; the frame's first model): the frame's saturation state, to be zeroed together with this model's per-frame
// totals before anything of the frame reads them — folded into the first slab's block-table kernel where there is one.
static gsx_status do_bin_and_composite(gsx_viewer* v, Model* m, bool carry, const ZeroJob* frame_zero = nullptr, bool cont = false) {
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
    if (m->stats_copy_inflight && hipEventQuery(m->stats_event) == hipSuccess) {
        m->stats_copy_inflight = false;
        if (progressive) {
            // (the slab plan of the next PLAIN frame — a probe of the speculation tuner — from the last plain frame: taken from a speculated
            //  frame's single slab the hint merged everything behind the second slab into one, 11 M entries where four slabs make 1.5 M)
            if (!m->stats_copy_speculated) m->slabs_hint = m->h_counters->slabs_used;
            if (spec_debug_level() >= 3)
                fprintf(stderr, "[gsx stats] viewer %p: copy of a %s frame: slabs_used %u, n_sorted %u, entries %u, max_needed %u, walk_max %u chunks (fine blocks: speculated %d, plain %d)\n", (void*)v, m->stats_copy_speculated ? "speculated" : "plain",
                        m->h_counters->slabs_used, m->h_counters->n_sorted, m->h_counters->n_entries_total, m->h_counters->max_needed, m->h_counters->walk_max, (int)m->blocks_fine_spec, (int)m->blocks_fine_plain);
            m->n_sorted = m->h_counters->n_sorted;
        }
        if (m->stats_copy_speculated) {
            m->last_spec_sorted = m->h_counters->n_sorted;
            m->last_repair_sorted = m->h_counters->n_sorted2;
        }
        if (m->stats_copy_slab_shading) m->slab_shading_off = (uint64_t)m->h_counters->n_shaded_total * 100u > (uint64_t)m->h_counters->n_visible * kSlabShadingMaxPercent;
        note_overflow(m);  // a free-running loop learns here that some earlier frame spilled: larger pair buffers from now on
        // long block lists (a scene where little saturates): the block sort carries the lists' records along (k_composite_blocks
        // SORTED); with hysteresis, from whatever frame's statistics arrived last — either way the pixels are the same
        const uint64_t per_tile = (uint64_t)m->h_counters->n_entries_total / std::max<uint32_t>(n_tiles, 1u);
        m->lists_long = m->lists_long ? per_tile > 500u : per_tile > 900u;
        // a tile that never saturates walks its block's whole list, 128 candidates per ~1.5 us, and the launch is as slow as that tile
        // (an open horizon: 425 chunks for 1900 takers with blocks of 8 x 4 tiles).  Blocks of a quarter the size cut the walk ~3.5x
        // and cost a second digit in the block sort and ~2x the entries: - 12 % on cfg4's orbit, + 16 % under an open sky
        // (tools/ab_blocks.py, round 5).  So the block size follows the longest walk of an earlier frame (SlabStats::walk_max, left
        // by tile_order_job), with hysteresis; either way the pixels are the same.
        // (one flag per schedule: an unspeculated frame — a probe of the tuner among speculated ones — walks its first slab's whole lists,
        //  several times the walk of the speculated frames around it; its statistic used to put THEM into fine blocks, where the walk
        //  then stayed above the way-back threshold: the same bench command read 2040 or 1830 fps with one frame in flight, 20.0 or 21.6
        //  launches a frame, depending on whether a stats copy happened to catch a probe frame — round 6)
        const uint32_t walk = m->h_counters->walk_max;
        bool& fine = m->stats_copy_speculated ? m->blocks_fine_spec : m->blocks_fine_plain;
        fine = fine ? walk > kWalkFineOff : walk > kWalkFineOn;
    }
    std::vector<uint32_t> bounds;
    const bool imported_windows = m->use_imported && m->has_window && progressive;
    if (m->spec_round1 || imported_windows) {
        // a speculated round is ONE slab: the windows already bound what every tile takes to little more than it needs,
        // and the compositor stops a saturated tile by itself; more slabs only add launches (measured on cfg4: 551 fps
        // with one slab, 487 with three).  The kernels stride over what exists on the device, so the bound is free.
        // (Same for the imported records of an index-sharded frame whose exchange was windowed.)
        bounds = {0u, (uint32_t)m->rec_n};
    } else {
        plan_slabs(v->options, (uint32_t)m->rec_n, &bounds);
        if (progressive) merge_tail_slabs(&bounds, m->slabs_hint);
    }
    if (spec_debug_level() >= 3) {
        fprintf(stderr, "[gsx slabs] viewer %p model '%s': speculated %d, slab shading %d, hint %u, slabs", (void*)v, m->key.c_str(), (int)m->spec_round1, (int)m->slab_shading, m->slabs_hint);
        for (uint32_t b : bounds) fprintf(stderr, " %u", b);
        fprintf(stderr, "\n");
    }
    Counters* dc = m->counters.as<Counters>();
    const uint32_t row_lo = std::min(m->row_lo, m->fc.tiles_y), row_hi = std::min(m->row_hi, m->fc.tiles_y);
    const uint32_t owned_tiles = (row_hi > row_lo ? row_hi - row_lo : 0) * m->fc.tiles_x;
    const uint2* window = (m->use_imported && m->has_window) ? (m->window_ptr ? m->window_ptr : m->window.as<uint2>()) : nullptr;
    if (m->spec_round1) window = m->spec_win.as<uint2>();
    uint32_t* tile_sat = progressive ? done + row_words * m->fc.tiles_y : nullptr;  // [count | bitmap | saturation keys | row work]
    // multi-GPU: what the tiles of every tile row walked, summed — the next frame's bands are balanced by it (gsx_shard_frame.cpp)
    uint32_t* row_work = (progressive && m->use_imported) ? tile_sat + (size_t)n_tiles : nullptr;

    // pair capacity to begin with: 16 entries per record for small models (per-tile lists), 6 for large ones (block lists need
    // ~3 per record on speculated frames, 0.2 on depth slabs; a frame that wants more spills on the device and the host grows
    // the buffers when it learns of it) — 32 bytes of pair / sort buffers per entry
    // (a frame without depth slabs keeps per-tile lists of the whole model: 16 as well)
    {
        // (round 6: large progressive models start at TWO entries per record — cfg4's largest slab makes 0.2, 32 bytes of pair / sort buffers an
        //  entry were 192 bytes a Gaussian and lane at the 6 entries of rounds 2-5; a scene that wants more — large splats, nothing saturating —
        //  spills on the device for the few frames it takes the host to learn of it: note_overflow doubles, or jumps to what was needed)
        const uint64_t per_record = (!progressive || m->rec_n <= (1u << 18)) ? 16u : 2u;
        m->tile_cap = std::max<uint64_t>(m->tile_cap, std::max<uint64_t>(1u << 20, per_record * m->rec_n));
    }
    if (v->tile_cap_fixed) m->tile_cap = v->tile_cap_fixed;  // GSX_TILE_CAP (tests): a capacity that overflows on purpose
    m->tile_cap = std::min<uint64_t>(m->tile_cap, 0xFFFFF000ull);
    const uint32_t cap = (uint32_t)m->tile_cap;
    {
        const size_t bytes = sizeof(uint32_t) * (size_t)cap;
        HIPCHK(m->tp_src.ensure(2 * bytes));
        HIPCHK(m->tk_out.ensure(bytes));
        HIPCHK(m->tv_out.ensure(bytes));
        const size_t ws = 4 * radix_workspace_words(cap);
        if (ws > m->tsort_ws.bytes) {
            HIPCHK(m->tsort_ws.ensure(ws));
            HIPCHK(gsx::op::MemsetAsync(m->tsort_ws.p, 0, m->tsort_ws.bytes, v->stream));
        }
        if (sizeof(uint2) * (size_t)n_tiles > m->ranges.bytes) m->ranges_clean = false;
        HIPCHK(m->ranges.ensure(sizeof(uint2) * (size_t)std::max<uint32_t>(n_tiles, 1024u)));  // (block lists: up to 1024 block ranges)
    }
    // reset this model's per-frame totals (n_visible and n_sorted stay) — and, for the frame's first model, the frame's saturation
    // state: one launch, or none where the first slab's block-table kernel can do it on its way (below)
    ZeroJob zero{&dc->n_entries, (uint32_t)((sizeof(Counters) - offsetof(Counters, n_entries)) / 4), nullptr, 0};
    if (frame_zero) {
        zero.b = frame_zero->a;
        zero.nb = frame_zero->na;
    }
    bool zero_pending = true;
    const uint32_t* done_before = nullptr;
    uint32_t keep_done_words = 0;
    if (speculate) {
        const size_t bm = 4 * (size_t)row_words * m->fc.tiles_y;
        HIPCHK(m->spec_win.ensure(sizeof(uint2) * (size_t)n_tiles));
        HIPCHK(m->spec_win2.ensure(sizeof(uint2) * (size_t)n_tiles));
        if (carry) {  // nearer models already saturated some tiles: remember which, they say nothing about this model
            HIPCHK(m->spec_done_before.ensure(bm));
            done_before = m->spec_done_before.as<uint32_t>();
            keep_done_words = (uint32_t)(bm / 4);
        }
    }
    const int bits = std::max<int>(1, (int)ceil_log2(n_tiles));
    // Progressive frames bin by BLOCKS of tiles (<= 256 of them: one 8-bit sort pass) and let the compositor decide per tile
    // (kernels_bin.hip "block lists").
    // a single-slab front model keeps complete per-tile lists (gsx_model_download_tile_lists): that frame bins by tile
    const bool lists_wanted = bounds.size() == 2 && !carry && !m->spec_round1 && !imported_windows;
    const bool blocks = progressive && v->bin_mode == 1 && !lists_wanted;
    if (!blocks || bounds.size() < 2 || bounds[1] == bounds[0]) {  // no block-table kernel ahead (or no slab at all): zero here
        HIPCHK(launch_zero_words(v->stream, zero.a, zero.na, zero.b, zero.nb));
        zero_pending = false;
        if (keep_done_words)
            HIPCHK(gsx::op::MemcpyAsync(m->spec_done_before.p, done, 4 * (size_t)keep_done_words, hipMemcpyDeviceToDevice, v->stream));
    } else if (keep_done_words) {   // ... the first slab's block-table kernel copies the bitmap on its way (it runs before anything sets a bit)
        zero.copy_src = done;
        zero.copy_dst = m->spec_done_before.as<uint32_t>();
        zero.n_copy = keep_done_words;
    }
    uint32_t bsx = 0, bsy = 0;
    // most blocks of this frame: GSX_BLOCKS_MAX when it was given, otherwise 256 (one 8-bit sort pass) — or 1024 while some tile's walk is long
    const uint32_t blocks_max = v->blocks_adaptive && (m->spec_round1 ? m->blocks_fine_spec : m->blocks_fine_plain) ? 1024u : v->blocks_max;
    if (blocks) {
        // The grid covers the rows this viewer composites (block_grid).  An index-sharded rank sizes its blocks for the taller of its
        // own band and an EQUAL band: every rank whose band is no taller than that bins by the same block size, so what a tile row
        // costs — the figure the next frame's bands are balanced by — does not change with the band it happens to lie in (sized by
        // the own band alone, a row was cheaper in a short band than in a tall one and the balance settled at 1.46 x the mean).
        const uint32_t grid_rows = std::max(row_hi > row_lo ? row_hi - row_lo : 1u, m->use_imported ? m->rows_nominal : 0u);
        auto count = [&]() { const BlockGrid g = block_grid(bsx, bsy, m->fc.tiles_x, 0, grid_rows); return (uint64_t)g.blocks_x * g.blocks_y; };
        while (count() > blocks_max) (bsx <= bsy ? bsx : bsy) += 1;
        HIPCHK(m->block_table.ensure(sizeof(uint4) * 1024));
        if (v->tile_profile) HIPCHK(v->tile_prof.ensure(sizeof(uint4) * (size_t)n_tiles));
    }
    {   // the pair sort's ping-pong buffers: only a sort of two digits has an intermediate, only one of three a second (block lists of
        // <= 256 blocks — every slab of a large model — are ONE digit: emitted pairs -> sorted keys / values, 32 bytes a Gaussian and lane less)
        const int pair_bits = blocks ? (int)std::max<uint32_t>(1u, ceil_log2(blocks_max)) : bits;
        const size_t pair_bytes = 2 * sizeof(uint32_t) * (size_t)cap;
        if (pair_bits > 8) HIPCHK(m->tp_a.ensure(pair_bytes));
        if (pair_bits > 16) HIPCHK(m->tp_b.ensure(pair_bytes));
    }
    // Block compositor: the tiles that were expensive in the model's frame before are dispatched first (k_composite_blocks; a
    // schedule, not data).  The order is made by one more workgroup of the frame's first block-table kernel; a second round of the
    // same frame (cont) keeps the first round's, and what it costs counts towards the next frame's.
    uint32_t* order_buf = nullptr;
    bool order_build = false;
    if (blocks && v->tile_order_on && n_tiles <= kTileOrderMax && bounds.size() >= 2 && bounds[1] > bounds[0]) {
        if (m->tile_order_tiles != n_tiles) {
            HIPCHK(m->tile_order.ensure(4 * (1 + 2 * (size_t)n_tiles)));
            HIPCHK(gsx::op::MemsetAsync(m->tile_order.p, 0, 4 * (1 + (size_t)n_tiles), v->stream));
            m->tile_order_tiles = n_tiles;
            m->tile_order_valid = false;
        }
        order_buf = m->tile_order.as<uint32_t>();
        order_build = !cont || !m->tile_order_valid;
    }
    // a single-slab front model keeps its complete tile lists for gsx_model_download_tile_lists
    const bool clear_ranges = progressive && !(bounds.size() == 2 && !carry && !m->spec_round1 && !imported_windows);
    // one depth slab [j0, j1) of the current depth order: bin -> tile sort -> ranges -> composite
    auto run_slab = [&](uint32_t j0, uint32_t j1, bool later, const uint2* win, const uint32_t* d_n, uint32_t slab_index,
                        const WindowPyramid* min_ends = nullptr, bool table_ready = false) -> gsx_status {
        // the very first slab of the frame sees no saturated tile: plain rectangle areas
        const uint32_t* done_in = later ? done : nullptr;
        bool sorted_records = false;  // block lists: the compositor's candidates travel through the block sort (long lists)
        // a slab of S splats can produce at most S * n_tiles entries; size the sort launch by the smaller bound
        const uint32_t slab_cap = (uint32_t)std::min<uint64_t>(cap, (uint64_t)(j1 - j0) * std::min<uint64_t>(owned_tiles, 1u << 16));
        if (blocks) {
            // (the block sort's one-digit histogram — entries per block — counted by the emit kernel on its way instead of by a launch of
            //  its own was measured, round 4: two launches less per frame, and slower — every emit workgroup flushes up to 256 bins to
            //  the same 256 addresses: binning 69 -> 86 us against block sort 37 -> 25 on a speculated cfg4 frame, 218 -> 327 against
            //  69 -> 45 unspeculated.  Not kept.)
            const int block_bits = (int)std::max<uint32_t>(1u, ceil_log2(blocks_max));
            ZeroJob jobs = zero_pending ? zero : ZeroJob{};
            if (order_build && !table_ready) {
                jobs.order_buf = order_buf;
                jobs.order_tiles = n_tiles;
                m->tile_order_valid = true;
                order_build = false;
            }
            const bool fused = v->bin_fused;
            const bool slab_shade = fused && m->slab_shading && m->lazy && !m->spec_round1;
            if (slab_shade) HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(m->rec_n, 1)));
            {
                ScopedPass t(v, GSX_PASS_BIN);
                if (fused) {
                    const size_t bw = 4 * bin_workspace_words(m->sortbin_cap);
                    if (m->bin_ws.bytes < bw) {
                        HIPCHK(m->bin_ws.ensure(bw));
                        HIPCHK(gsx::op::MemsetAsync(m->bin_ws.p, 0, m->bin_ws.bytes, v->stream));
                    }
                    HIPCHK(launch_block_bin_fused(v->stream, j0, j1, d_n, m->sorted_idx, m->rec(), m->sk_out.as<uint32_t>(), m->srect.as<uint4>(), dc, cap,
                                                  row_lo, row_hi, done_in, row_words, (progressive && later) ? done_count : nullptr, owned_tiles, slab_index,
                                                  win, m->fc.tiles_x, m->fc.tiles_y, bsx, bsy, m->block_table.as<uint4>(), m->tp_src.as<uint2>(),
                                                  m->ranges.as<uint2>(), jobs, table_ready, m->bin_ws.as<uint32_t>(), m->tsort_ws.as<uint32_t>(), block_bits,
                                                  slab_shade ? m->adm_pairs.as<uint2>() : nullptr,
                                                  (slab_shade && m->sorted_code_valid) ? m->sorted_code.as<uint8_t>() : nullptr));
                    // slab shading: conic / colour records for exactly the records of this slab some block takes (and the frame's colour ops on them)
                    if (slab_shade) {
                        const gsx_status sst = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dc->n_slab_shade, nullptr, m->rect8_active});
                        if (sst) return sst;
                    }
                } else {
                    gsx_status cst = ensure_count_buffers(m);
                    if (cst) return cst;
                    HIPCHK(launch_block_bin(v->stream, j0, j1, d_n, m->sorted_idx, m->rec(), m->sk_out.as<uint32_t>(), m->srect.as<uint4>(),
                                            m->cnt.as<uint32_t>(), m->block_sums.as<uint32_t>(), dc, cap, row_lo, row_hi, done_in, row_words,
                                            (progressive && later) ? done_count : nullptr, owned_tiles, slab_index, win, m->fc.tiles_x,
                                            m->fc.tiles_y, bsx, bsy, m->block_table.as<uint4>(), m->tp_src.as<uint2>(), m->ranges.as<uint2>(),
                                            jobs, table_ready));
                }
                zero_pending = false;
                v->pass_launches[GSX_PASS_BIN] += 1;
            }
            {
                ScopedPass t(v, GSX_PASS_TILE_SORT);
                RadixBuffers rb{nullptr, nullptr, m->tp_src.as<uint2>(), m->tk_out.as<uint32_t>(), m->tv_out.as<uint32_t>(),
                                m->tp_a.as<uint2>(), m->tp_b.as<uint2>(), m->tsort_ws.as<uint32_t>()};
                const uint32_t block_cap = (uint32_t)std::min<uint64_t>(cap, (uint64_t)(j1 - j0) * 256u);
                // (<= 256 blocks: ONE digit, and the block ranges are the scan of its histogram — no k_tile_ranges launch;
                //  k_block_table zeroed the ranges, which is what stays when the slab made no entry at all)
                // (<= 256 blocks: ONE digit, and the block ranges are the scan of its histogram — no k_tile_ranges launch;
                //  k_block_table zeroed the ranges, which is what stays when the slab made no entry at all)
                // sorted_records: the write-out also carries every entry's {rect, key, index} record along
                sorted_records = v->sorted_records >= 0 ? v->sorted_records == 1 : m->lists_long;  // (the LAST digit pass carries them)
                if (sorted_records) HIPCHK(m->brec_sorted.ensure(sizeof(uint4) * (size_t)cap));
                HIPCHK(launch_radix_sort(v->stream, rb, block_cap, &dc->n_entries, block_bits, false, false, block_bits <= 8 ? m->ranges.as<uint2>() : nullptr,
                                         sorted_records ? m->srect.as<uint4>() : nullptr, sorted_records ? m->brec_sorted.as<uint4>() : nullptr, fused));
                m->tile_keys = m->tk_out.as<uint32_t>();
                m->tile_list = m->tv_out.as<uint32_t>();
                v->pass_launches[GSX_PASS_TILE_SORT] += 1;
                if (block_bits > 8) {  // GSX_BLOCKS_MAX above 256: two digits, ranges from the sorted keys
                    ScopedPass t2(v, GSX_PASS_BIN);
                    HIPCHK(launch_tile_ranges(v->stream, block_cap, &dc->n_entries, m->tile_keys, 1024u, m->ranges.as<uint2>(), true));
                }
                m->ranges_clean = false;
            }
        } else {
            {
                ScopedPass t(v, GSX_PASS_BIN);
                gsx_status cst = ensure_count_buffers(m);
                if (cst) return cst;
                HIPCHK(launch_tile_counts(v->stream, j0, j1, d_n, m->sorted_idx, m->rec(), m->srect.as<uint2>(),
                                          m->cnt.as<uint32_t>(), m->block_sums.as<uint32_t>(), dc, cap, row_lo, row_hi, done_in,
                                          row_words, (progressive && later) ? done_count : nullptr, owned_tiles, slab_index,
                                          win, m->sk_out.as<uint32_t>(), m->fc.tiles_x, min_ends));
                HIPCHK(launch_tile_emit(v->stream, j0, j1, m->sorted_idx, m->srect.as<uint2>(), m->cnt.as<uint32_t>(),
                                        m->block_sums.as<uint32_t>(), m->fc.tiles_x, m->tp_src.as<uint2>(), row_lo, row_hi,
                                        done_in, row_words, d_n, &dc->n_entries, cap, win, m->sk_out.as<uint32_t>(), &dc->slab_cut));
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
                HIPCHK(launch_tile_ranges(v->stream, slab_cap, &dc->n_entries, m->tile_keys, (uint32_t)(m->ranges.bytes / sizeof(uint2)),
                                          m->ranges.as<uint2>(), m->ranges_clean));
            }
        }
        if (v->validate) {  // debug: check what the compositor will dereference, on the host, before it runs
            HIPCHK(v->scratch.ensure(64));
            HIPCHK(gsx::op::MemsetAsync(v->scratch.p, 0, 64, v->stream));
            // (block lists: one range per block, list values are positions in the slab)
            const uint32_t n_ranges = blocks ? block_grid(bsx, bsy, m->fc.tiles_x, row_lo, row_hi).blocks_x * block_grid(bsx, bsy, m->fc.tiles_x, row_lo, row_hi).blocks_y : n_tiles;
            HIPCHK(launch_validate_tiles(v->stream, m->ranges.as<uint2>(), n_ranges, m->tile_list, &dc->n_entries, blocks ? cap : slab_cap,
                                         blocks ? j1 - j0 : (uint32_t)m->rec_n, v->scratch.as<uint32_t>()));
            uint32_t rep[8];
            HIPCHK(gsx::op::MemcpyAsync(rep, v->scratch.p, 32, hipMemcpyDeviceToHost, v->stream));
            HIPCHK(gsx::op::StreamSynchronize(v->stream));
            if (rep[0])
                return fail(GSX_ERR_HIP, "GSX_VALIDATE: model '%s' slab %u: %s (tile %u: %u, %u, %u); n_tiles %u, tiles %ux%u, ranges_clean %d, "
                            "clear_ranges %d, speculated %d, later %d, slab_cap %u, rec_n %llu", m->key.c_str(), slab_index,
                            rep[0] == 1 ? "tile range outside the sorted entries" : "list index outside the records", rep[1], rep[2], rep[3], rep[4],
                            n_tiles, m->fc.tiles_x, m->fc.tiles_y, (int)m->ranges_clean, (int)clear_ranges, (int)m->spec_round1, (int)later, slab_cap,
                            (unsigned long long)m->rec_n);
        }
        {
            ScopedPass t(v, GSX_PASS_COMPOSITE);
            if (blocks) {
                // (a slab whose entries did not fit the pair buffers — decided on the device — has its tail composited pair-free by the
                //  same launch: the frame is complete without a host round trip)
                HIPCHK(launch_composite_blocks(v->stream, m->fc, m->ranges.as<uint2>(), sorted_records ? nullptr : m->tile_list,
                                               sorted_records ? m->brec_sorted.as<uint4>() : m->srect.as<uint4>(), m->rec(), fb_ptr(v),
                                               later, done, row_words, done_count, tile_sat, win, row_lo, row_hi, bsx, bsy, row_work, dc, j1,
                                               d_n, m->sorted_idx, m->sk_out.as<uint32_t>(),
                                               (v->tile_profile && slab_index == 0) ? v->tile_prof.as<uint4>() : nullptr,
                                               (order_buf && m->tile_order_valid) ? order_buf + 1 + n_tiles : nullptr,
                                               order_buf ? order_buf + 1 : nullptr, m->rec().rect8 /* (null for imported records) */));
            } else {
                HIPCHK(launch_composite(v->stream, m->fc, m->ranges.as<uint2>(), m->tile_list, m->rec(), fb_ptr(v),
                                        later, done, row_words, done_count, clear_ranges, tile_sat, row_work));
                m->ranges_clean = clear_ranges;  // the compositor zeroed every range it consumed
            }
            v->pass_launches[GSX_PASS_COMPOSITE] += 1;
            // per-tile lists: the slab's entries did not fit the pair buffers (decided on the device): its tail is composited
            // pair-free, so the frame is complete without a host round trip; otherwise this launch falls through
            if (!blocks)
                HIPCHK(launch_composite_spill(v->stream, m->fc, dc, j1, d_n, m->sorted_idx, m->sk_out.as<uint32_t>(), m->rec(), fb_ptr(v),
                                              done, row_words, done_count, tile_sat, row_lo, row_hi, win));
        }
        return GSX_OK;
    };
    gsx_status st = GSX_OK;
    // a speculated round's windows all start at 0 and come with the min-pyramid of their ends (enqueue_next_windows)
    WindowPyramid min_ends{};
    bool have_min_ends = false;
    if (m->spec_round1) {
        const size_t pw = window_pyramid_words(m->fc.tiles_x, m->fc.tiles_y);
        min_ends = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse.as<uint32_t>() + pw);
        have_min_ends = true;
    } else if (imported_windows && m->import_min_ends) {  // round 0 of an index-sharded frame: windows [0, limit), pyramid built at frame begin
        min_ends = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->import_min_ends);
        have_min_ends = true;
    }
    for (size_t sl = 0; sl + 1 < bounds.size(); ++sl)
        if ((st = run_slab(bounds[sl], bounds[sl + 1], carry || sl > 0, window, &dc->n_sorted, (uint32_t)sl,
                           have_min_ends ? &min_ends : nullptr)))
            return st;
    bool windows_enqueued = false;
    auto enqueue_next_windows = [&]() -> gsx_status {  // this model's windows for its next frame
        ScopedPass t(v, GSX_PASS_COMPOSITE);
        HIPCHK(launch_spec_next(v->stream, tile_sat, done, done_before, row_words, m->fc.tiles_x, m->fc.tiles_y,
                                v->options.spec_margin, v->options.spec_radius, m->spec_win.as<uint2>(), row_lo, row_hi));
        // [max-pyramid of the window ends: admission in k_project | min-pyramid: "every tile takes it" in the binning]
        // (the two as ONE launch — the last workgroup of k_spec_next building the pyramids — was measured, round 4: 20 us against
        //  4.9 + 7.2: the device-scope fence in front of the ticket writes back what the compositor has just left dirty in the L2)
        const size_t pw = window_pyramid_words(m->fc.tiles_x, m->fc.tiles_y);
        HIPCHK(m->spec_coarse.ensure(8 * pw));
        HIPCHK(launch_window_pyramid(v->stream, m->spec_win.as<uint2>(), m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse.as<uint32_t>(), false,
                                     nullptr, m->spec_coarse.as<uint32_t>() + pw));
        m->spec_valid = true;
        m->spec_tiles_x = m->fc.tiles_x;
        m->spec_tiles_y = m->fc.tiles_y;
        return GSX_OK;
    };
    if (m->spec_round1) {
        // verification on the device: tiles with a bounded window that are still open get, in one more round, exactly
        // the records they were refused, composited behind what they hold.
        const uint32_t n = (uint32_t)m->rec_n;
        bool repair = true;
        {
            ScopedPass t(v, GSX_PASS_DEPTH_SORT);
            // auto (2): ask while repairs are rare AND the host waits for its frames anyway (it has called gsx_sync or a blocking readback
            // since the frame before: the app's protocol, scene.rs:614, 873).  A host that streams frames without waiting is better off with
            // the repair round always enqueued and decided on the device — eight launches that fall through, ~38 us of stream time since
            // round 6 (22 launches, ~100 us, when this rule was made): cfg4, two / one frames in flight streaming / waiting per frame,
            // always-device 2350 / 2064 / 1902 fps, always-ask 1980 / 2089 / 1950, the round-5 rule (ask whenever repairs are rare)
            // 2200 / 1836 / 1955 — the worst of both while streaming (tools/ab_host_verify.py, profiles/r06_ab_host_verify.txt).
            // While repairs are not rare, the verdicts are still posted and the host merely
            // LOOKS at the latest one each frame (it lags by the frames in flight, and costs nothing): eight repair-free
            // verdicts in a row and the host asks again.  (A blocking probe here cost milliseconds: a host that does not wait
            // is ~10 frames ahead of the device.)
            const bool automatic = v->options.host_verify == 2;
            if (automatic && !m->hv_active && v->h_verdict) {
                const unsigned long long w = __atomic_load_n(v->h_verdict, __ATOMIC_ACQUIRE);
                const uint32_t wseq = (uint32_t)(w >> 32);
                if (wseq != 0 && wseq != m->hv_seen_seq) {
                    m->hv_seen_seq = wseq;
                    m->hv_quiet = (uint32_t)w == 0u ? m->hv_quiet + 1u : 0u;
                    if (m->hv_quiet >= 8) m->hv_active = true;
                }
            }
            const bool ask = v->options.host_verify == 1 || (automatic && m->hv_active && (v->parent ? v->parent : v)->host_waited);
            const bool post = ask || automatic;
            if (post && !v->h_verdict) {
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&v->h_verdict), 64, hipHostMallocDefault));
                *v->h_verdict = 0;
            }
            const uint32_t seq = post ? ++v->verify_seq : 0;
            HIPCHK(m->spec_need.ensure(4 * (size_t)row_words * m->fc.tiles_y));
            // ... and in the same launch, when something needs repair: the min-pyramid of the repair windows' starts (the repair
            // round's conservative admission test: four loads per record; an exact per-tile scan of every visible record cost
            // 260-350 us here) and the repair slab's block table
            HIPCHK(m->spec_coarse2.ensure(4 * window_pyramid_words(m->fc.tiles_x, m->fc.tiles_y)));
            const BlockGrid grid = block_grid(bsx, bsy, m->fc.tiles_x, row_lo, row_hi);
                HIPCHK(launch_spec_verify(v->stream, m->spec_win.as<uint2>(), done, row_words, m->fc.tiles_x, m->fc.tiles_y,
                                      m->spec_win2.as<uint2>(), m->spec_need.as<uint32_t>(), &dc->spec_need, row_lo, row_hi,
                                      post ? v->h_verdict : nullptr, seq, m->spec_coarse2.as<uint32_t>(), blocks ? &grid : nullptr,
                                      blocks ? m->block_table.as<uint4>() : nullptr, blocks ? m->ranges.as<uint2>() : nullptr));
            if (ask) {
                // Nothing to repair (most frames): the ~20 launches of the second round would all fall through, at a few
                // microseconds of stream time each.  So the verdict comes to the host: one pinned word, written by the
                // verification kernel.  The next frame's windows are enqueued first — they are what follows when there is
                // nothing to repair, and they keep the stream busy while the word travels; after a repair they are redone.
                if ((st = enqueue_next_windows())) return st;
                uint32_t need = 0;
                if ((st = wait_verdict(v, seq, &need))) return st;
                repair = need != 0;
                windows_enqueued = !repair;
                // auto: a frame that repairs costs more with the wait than without (the host enqueues the second round
                // while the device idles); stop asking when half of the last eight frames repaired
                m->hv_history = (m->hv_history << 1) | (repair ? 1u : 0u);
                if (automatic && __builtin_popcount(m->hv_history & 0xFFu) >= 6) {
                    m->hv_active = false;
                    m->hv_history = 0;
                    m->hv_quiet = 0;
                }
            }
        }
        if (repair) {
            ScopedPass t(v, GSX_PASS_DEPTH_SORT);
            HIPCHK(m->adm_ballots2.ensure(8 * ((std::max<size_t>(n, 1) + 63) / 64)));
            HIPCHK(m->adm_counts2.ensure(4 * (std::max<size_t>(admit_blocks(n), 1) + 4)));
            // conservative admission against the min-pyramid of the repair windows' starts (launch_spec_verify built it; the
            // binning applies the exact windows)
            WindowPyramid pyr2 = window_pyramid_layout(m->fc.tiles_x, m->fc.tiles_y, m->spec_coarse2.as<uint32_t>());
            pyr2.min_of_starts = 1;
            uint32_t seq2 = 0;
            const bool bucket2 = v->bucket_sort && !(m->last_repair_sorted > kBucketSortMax);
            if (bucket2) {  // (the repair round's keys lie behind the windows: a population, and a key range, of their own)
                if ((st = ensure_msd(v, m, m->msd_ws2))) return st;
                seq2 = m->msd_seq2++;
            }
            HIPCHK(launch_admit(v->stream, m->proj_rec(), n, nullptr, m->fc.tiles_x, nullptr,
                                row_words, pyr2, &dc->spec_need,
                                m->adm_ballots2.as<unsigned long long>(), m->adm_counts2.as<uint32_t>(), &dc->n_sorted2,
                                m->adm_pairs.as<uint2>(), bucket2 ? m->msd_ws2.as<uint32_t>() : nullptr, seq2));
            if (m->lazy) {  // the repair round needs records the lazy projection did not shade
                gsx_status sst = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted2, m->adm_ballots.as<unsigned long long>(), m->rect8_active});
                if (sst) return sst;
            }
            RadixBuffers rb{nullptr, nullptr, m->adm_pairs.as<uint2>(), m->sk_out.as<uint32_t>(), m->sv_out.as<uint32_t>(),
                            m->dp_a.as<uint2>(), m->dp_b.as<uint2>(), m->sort_ws.as<uint32_t>()};
            if (bucket2) {
                HIPCHK(launch_bucket_sort(v->stream, rb, n, &dc->n_sorted2, false, m->msd_ws2.as<uint32_t>(), seq2, true));
            } else {
                HIPCHK(launch_radix_sort(v->stream, rb, n, &dc->n_sorted2, 32, false));
            }
        }
        if (repair && (st = run_slab(0, n, true, m->spec_win2.as<uint2>(), &dc->n_sorted2, (uint32_t)bounds.size(), nullptr, blocks))) return st;
        m->order_consumed = true;
    }
    if (speculate && m->windows_unwanted) m->spec_valid = false;
    else if (speculate && !windows_enqueued && (st = enqueue_next_windows())) return st;
    // feed the next frames' slab plan without waiting — every fourth frame is plenty (the copy is two runtime kernels)
    if (!m->stats_copy_inflight && (m->stats_copy_tick++ & 3u) == 0) {
        if (!m->stats_event) HIPCHK(hipEventCreateWithFlags(&m->stats_event, hipEventDisableTiming));
        HIPCHK(gsx::op::MemcpyAsync(m->h_counters, m->counters.p, sizeof(Counters), hipMemcpyDeviceToHost, v->stream));
        HIPCHK(gsx::op::EventRecord(m->stats_event, v->stream));
        m->stats_copy_inflight = true;
        m->stats_copy_speculated = m->spec_round1;
        m->stats_copy_slab_shading = m->slab_shading && m->lazy;
    }
    tuner_frame_end(v, m);
    m->binned = true;
    m->stats_pending = true;
    m->lists_complete = bounds.size() == 2 && !carry && !m->spec_round1 && !imported_windows;
    return GSX_OK;
}

// cont: a second round of the same frame (multi-GPU back set): keep the framebuffer, the saturated-tile state
// and carry (C, T) into the first model.
gsx_status do_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys, bool cont) {
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
    ZeroJob frame_zero{};
    if (!cont) {   // one memset: [saturated-tile counter | saturated-tile bitmap | per-tile saturation depth keys | per-row work]
        const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
        const uint32_t row_words = (tiles_x + 31) / 32;
        const size_t bytes = 4 * (1 + (size_t)tiles_y * row_words + (size_t)tiles_y * tiles_x + tiles_y);
        HIPCHK(v->done_bits.ensure(bytes));
        // ... zeroed together with the per-frame totals of the model composited first (do_bin_and_composite)
        frame_zero = ZeroJob{v->done_bits.as<uint32_t>(), (uint32_t)(bytes / 4), nullptr, 0};
    }
    // the reference paints far -> near with "over"; front-to-back accumulation walks the same list backwards
    bool carry = cont;
    for (auto it = order.rbegin(); it != order.rend(); ++it) {
        if ((st = do_bin_and_composite(v, *it, carry, (!cont && it == order.rbegin()) ? &frame_zero : nullptr, cont))) return st;
        carry = true;
    }
    return GSX_OK;
}

}  // namespace gsx
