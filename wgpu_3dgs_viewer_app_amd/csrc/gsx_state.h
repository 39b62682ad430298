// gsx_state.h — host-side state of libgsx.so shared by its translation units: device buffers, the per-model and
// per-viewer records, and the frame scheduling entry points (gsx_frame.cpp) the C ABI files call.
// Build-internal; the public surface is include/gsx.h.
#pragma once
#include <atomic>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gsx_internal.h"

namespace gsx {

extern thread_local std::string g_err;  // gsx_last_error_string()
gsx_status fail(gsx_status st, const char* fmt, ...);

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return fail(_e == hipErrorOutOfMemory ? GSX_ERR_OOM : GSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                   \
    } while (0)

// device bytes held by every DevBuf of the process (gsx_debug_device_bytes: the bench line's resident_bytes)
inline std::atomic<uint64_t> g_dev_bytes{0};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool borrowed = false;  // p belongs to another DevBuf (a lane's view of the model data, gsx_api.cpp)
    ~DevBuf() { release(); }
    void release() {
        if (p && !borrowed) {
            (void)gsx::op::Free(p);
            g_dev_bytes.fetch_sub(bytes, std::memory_order_relaxed);
        }
        p = nullptr;
        bytes = 0;
        borrowed = false;
    }
    void borrow(const DevBuf& o) {
        if (!borrowed) release();
        p = o.p;
        bytes = o.bytes;
        borrowed = o.p != nullptr;
    }
    // grow-only; contents are NOT preserved
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (borrowed) return hipErrorInvalidValue;  // a view never grows what it does not own
        release();
        // a quarter more than asked for, so that a buffer that creeps up is not reallocated every frame — for SMALL buffers: the large ones are
        // sized by the model's Gaussian count or by a capacity plan that carries its own margin (round 6: the slack alone was 120 bytes a Gaussian)
        size_t want = need + (need < (size_t(32) << 20) ? need / 4 : 0) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) {
            bytes = want;
            g_dev_bytes.fetch_add(want, std::memory_order_relaxed);
        }
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

using Counters = SlabStats;  // device copy + pinned host mirror

// Per model: which of the two equivalent schedules (speculated / plain progressive) is faster here — see gsx_frame.cpp.
struct SpecTuner {
    enum Phase { SPEC, PROBE_PLAIN, SETTLE_SPEC, PLAIN, PROBE_SPEC, SETTLE_PLAIN } phase = SPEC;
    uint32_t left = 32;                 // frames left in the phase (the first SPEC phase is short: decide early)
    uint32_t len_spec = 64, len_plain = 64, frame_no = 0;
    double mean_spec = 0.0, mean_plain = 0.0;  // running means of the bracketed frames, milliseconds
    uint32_t n_spec = 0, n_plain = 0;
    struct Slot {
        hipEvent_t start = nullptr, stop = nullptr;
        bool spec = false;
        int state = 0;  // 0 free, 1 start recorded, 2 stop recorded (in flight), 3 in flight but stale (reset() since)
        bool probe = false;  // a frame of a probe phase: the decision waits for these
    } slots[16];
    Slot* active = nullptr;
    uint32_t probe_pending = 0;         // probe frames whose timings have not arrived yet
    // the scene changed under the model (another mask, new Gaussians): what was measured belongs to the old scene
    void reset() {
        phase = SPEC; left = 32; len_spec = len_plain = 64;
        n_spec = n_plain = 0; mean_spec = mean_plain = 0.0; probe_pending = 0;
        for (auto& s : slots) {
            if (s.state == 2) s.state = 3;
            if (s.state == 1) s.state = 0;  // bracket opened, never closed (the frame was not rendered): nothing in flight
        }
        active = nullptr;
    }
    ~SpecTuner() {
        for (auto& s : slots) {
            if (s.start) (void)hipEventDestroy(s.start);
            if (s.stop) (void)hipEventDestroy(s.stop);
        }
    }
};

struct Model {
    std::string key;
    uint64_t serial = 0;                        // unique per gsx_model_create (a lane's shadow model remembers whose data it views)
    uint64_t shadow_of = 0;                     // != 0: this is a lane's shadow of the model with that serial
    uint64_t n = 0;
    gsx_sh_kind sh_kind = GSX_SH_SINGLE;
    gsx_cov3d_kind cov_kind = GSX_COV3D_SINGLE;
    bool has_sh = true;
    bool has_mask = false;
    ModelTransform mt;
    FrameConsts fc{};

    DevBuf pc, cov_a, cov_b, sh4, sh1, sh_h, sh_q, sh_aos, cov_h, cov_h2, mask;
    DevBuf key_buf, rec_a, rec_b, rec_c;        // projection records of the model's own Gaussians
    DevBuf code8, sorted_code;                  // slab shading: coarse screen cells of every Gaussian's rectangle (Records::code8), and the same in depth order
    bool code8_active = false, sorted_code_valid = false;
    DevBuf rect8;                               // packed tile rectangles of a lazily projected frame (Records::rect8)
    bool rect8_active = false;                  // this frame's projection wrote rect8 instead of the `a` records
    DevBuf imp_key, imp_a, imp_b, imp_c;        // records imported from other ranks (kept apart: a frame may pack twice)
    bool use_imported = false;
    uint64_t sortbin_cap = 0, imp_cap = 0;
    DevBuf dp_a, dp_b, sk_out, sv_out, sort_ws; // depth sort: pair scratch, sorted keys / indices, workspace
    DevBuf cnt, block_sums, srect;              // per slab: tile counts in depth order, scan partials, tile rects
    DevBuf block_vis;                           // per-workgroup visible counts of the projection pass
    DevBuf tp_src, tp_a, tp_b, tk_out, tv_out, tsort_ws;  // tile pairs: emitted, scratch, sorted (split), workspace
    DevBuf brec_sorted;                         // block lists: the {rect, key, index} records in list order (the block sort's write-out gathers them)
    bool blocks_fine_spec = false, blocks_fine_plain = false;   // some tile of the model walks a long list: blocks of a quarter the size (1024 instead of 256), per schedule — speculated / unspeculated frames (gsx_frame.cpp)
    bool lists_long = false;                    // ... for models whose lists are long (decided from the last frame statistics that arrived)
    DevBuf ranges;
    DevBuf tile_order;                          // block compositor's dispatch order: {threshold, tile_cost[n_tiles], tile_order[n_tiles]} (tile_order_job)
    uint32_t tile_order_tiles = 0;              // the tile count it is laid out for
    bool tile_order_valid = false;              // tile_order holds a permutation
    DevBuf block_table;                         // block lists: per block {min window start, max window end, live}
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
    uint32_t overflow_seen = 0;                 // SlabStats::overflow_events as last mirrored
    uint64_t overflow_slabs = 0;                // slabs that spilled over the model's lifetime (gsx_frame_stats)
    uint32_t slabs_hint = 0;                    // slabs the last observed frame needed (0 = unknown)
    hipEvent_t stats_event = nullptr;           // completion of the asynchronous statistics copy
    bool stats_copy_inflight = false;
    uint32_t stats_copy_tick = 0;
    bool stats_copy_speculated = false;         // the frame whose statistics are in flight was speculated
    // the per-frame record set: the model's own projection (rec_n == n) or records imported from the
    // other ranks (gsx_shard_import); binning is restricted to the band of tile rows [row_lo, row_hi)
    uint64_t rec_n = 0, rec_cap = 0;
    uint32_t row_lo = 0, row_hi = 0xFFFFFFFFu;  // band of tile rows this viewer bins (clamped to tiles_y)
    uint32_t rows_nominal = 0;  // index-sharded frames: the height of an equal band, ceil(tiles_y / world) — the block size is chosen for max(own rows, this)
    DevBuf pack_table;
    // selection / edits / query (kernels_edit.hip); all allocated on first use
    DevBuf selection, edited, edit_a, edit_b, keep, query_flags, hits, hit_count;
    bool has_selection = false, has_edits = false, show_unedited = false;
    uint32_t flags_kind = GSX_QUERY_NONE, flags_op = GSX_SELECTION_SET;  // what the last preprocess evaluated
    // temporal occlusion speculation (kernels_spec.hip): this model's per-tile windows for its next frame, the repair
    // windows of the current one, the saturated-tile bitmap as it was before this model was composited
    DevBuf spec_win, spec_win2, spec_done_before, spec_need, spec_coarse, spec_coarse2;
    bool spec_valid = false, spec_round1 = false;
    // speculation that keeps repairing does not pay (sparse scenes whose tiles hover around saturation): the lazily read
    // statistics keep a history of "this frame needed the repair round"; too many -> unspeculated frames for a while
    uint64_t mask_program_hash = 0;    // of the last gsx_mask_evaluate program (0: none / uploaded words)
    bool windows_unwanted = false;     // this frame need not leave windows for the next (SpecTuner: a plain phase)
    SpecTuner* tuner_ref = nullptr;    // a lane's shadow model: the owner's tuner decides for every lane (one phase cycle per model)
    SpecTuner tuner;                   // speculate or not? decided by timing both paths (gsx_frame.cpp)
    // host_verify = 2 (auto): ask the device for its verdict only while repairs are rare
    bool hv_active = true;             // currently asking
    uint32_t hv_history = 0;           // bit k: the k-th latest verdict needed the repair round
    uint32_t hv_quiet = 0;             // while not asking: consecutive probes / sampled frames without a repair
    uint32_t hv_seen_seq = 0;          // while not asking: the last verdict the host has looked at
    bool order_consumed = false;   // a speculated render overwrote the depth order with its repair round's
    uint32_t spec_tiles_x = 0, spec_tiles_y = 0, shard_tiles_x = 0, shard_tiles_y = 0;
    // lazily projected shard (gsx_shard_set_windows): the windows of the coming exchange, their max-pyramid, and whether the
    // last preprocess left a candidate list in adm_pairs
    DevBuf shard_win, shard_pyr, trav_ballots, trav_counts;
    bool shard_win_set = false, cand_valid = false;
    // device-resident exchange (gsx_shard_frame_begin ... gsx_shard_next_windows): next frame's per-tile limits, the
    // repair round's windows, and what the host knows (late, never waited for) about how full the exchange slots get
    DevBuf shard_limit, shard_limit_next, shard_win2;
    DevBuf shard_win_next;                     // the windows [0, limit) of shard_limit_next, where the kernel that made the limits wrote them too
    bool shard_win_current = false, shard_win_next_valid = false;  // shard_win holds the windows of shard_limit / shard_win_next those of shard_limit_next
    uint32_t shard_win_tiles = 0;
    DevBuf shard_limit_override;               // gsx_shard_set_limits: limits for the model's NEXT sharded frame (whichever lane renders it)
    uint32_t shard_override_tiles = 0;         // != 0: an override is waiting, made for a grid of that many tiles
    bool shard_limit_valid = false, shard_next_valid = false, shard_frame_limited = false;
    bool shard_behind = false;                 // this frame's imported records were composited behind nearer models (spec_done_before holds their tiles)
    uint32_t shard_limit_tx = 0, shard_limit_ty = 0;
    std::vector<uint32_t> pair_counts;         // owner's model: records rank s wanted to send to rank d in round 0 of the model's last frame
                                               // whose verdict was read ([s * world + d]; empty: unknown) — next frame's slots, pair by pair
    bool pair_limited = false;                 // ... and whether that frame's exchange was limited by windows
    std::vector<uint32_t> pair_edges;          // ... and the band edges it ran with
    uint32_t slot_force = 0;                   // gsx_shard_set_slot_records: round-0 slot size instead of the policy's (0 = policy)
    uint32_t repair_hint = 0;                  // most records any (rank, destination) pair had in the repair round of the last frame whose verdict was read
    bool repair_overflowed = false;            // ... and whether a repair slot was too small for them
    uint32_t slot_hint = 0;                    // records the busiest (rank, destination) pair wanted in round 0 of the last frame: a GLOBAL
                                               // figure from that frame's verdict, so every rank sizes the next slots identically; 0 = unknown
    bool slot_hint_limited = false;            // ... of a frame whose exchange was limited by windows
    bool slot_hint_known = false;              // a verdict has been read since the limits were last replaced from outside (0 wanted is a figure too)
    DevBuf adm_ballots2;           // the repair round's ballots (the first round's stay: they say which records are shaded)
    bool lazy = false;             // this frame's projection shaded only the admitted Gaussians
    // k_edit_prepare is a function of (selection, stored edits, mask, the viewer's selection edit) and idempotent: it runs again
    // only after one of them changed (edit_epoch: bumped by every call that writes one of the buffers)
    uint64_t edit_epoch = 1, prep_epoch = 0;
    gsx_gaussian_edit prep_sel_edit{};
    bool prep_has_selection = false;
    const uint32_t* prep_mask = nullptr;
    bool frame_edits = false, frame_highlight = false;  // this frame applies stored / selection edits, the selection highlight (to whatever gets shaded)
    bool visible_count_pending = false;  // N_vis of this projection has not been summed yet (the admission scan will)
    const uint32_t* last_pyramid = nullptr;  // the admission pyramid the projection pass used
    uint32_t* last_pod_mask = nullptr;  // the keep-bitset the projection pass used (mask, or mask & ~hidden)
    DevBuf adm_offsets, adm_counts2;  // scan output of adm_counts; counts of the admission passes that run outside the projection
    DevBuf adm_pairs, adm_ballots, adm_counts;  // admission pass: compacted (key, index) pairs, per-wave ballots, per-workgroup counts
    DevBuf bin_ws;                   // k_block_bin: ticket + one status word per 2048-record tile
    DevBuf msd_ws, msd_ws2;          // bucket sort (gsx_internal.h): fine histogram, key-range cells, k_admit_compact's status words — main round / repair round
    bool slab_shading_off = false;   // the last slab-shaded frame shaded most of what it saw (a scene where nothing saturates): project in full until the data changes
    uint32_t slab_shading_retry = 0; // plain frames since: every 256th tries again
    bool stats_copy_slab_shading = false;
    bool slab_shading = false;       // this frame: geometry-only projection without windows; the depth slabs shade what their blocks take
    uint32_t last_spec_sorted = 0, last_repair_sorted = 0;  // admitted / repair records of the last SPECULATED frame whose statistics have arrived (lagging; 0: none yet)
    uint32_t msd_seq = 0, msd_seq2 = 0;  // sorts made on them so far (the cells rotate)
    DevBuf pack_masks;             // destination bit mask per record (gsx_shard_pack)
    DevBuf window, pack_window;    // per-tile depth-key windows [lo, hi): of the imported set / of the pack in flight
    bool has_window = false;
    const uint2* window_ptr = nullptr;          // != nullptr: the imported set's windows live here (no copy in `window`)
    const uint32_t* import_min_ends = nullptr;  // min-pyramid of the imported set's window ends (every window starts at 0), or nullptr
    DevBuf shard_pyr2;             // min-pyramid of the repair windows' starts (the repair round's conservative pack test)
    DevBuf shard_need_bits;        // device-resident exchange: bitmap of the tiles that need the repair round (gates its pack)
    bool repair_counted = false;
    bool verify_state_zeroed = false;   // the last feedback kernel zeroed the verification's counters and need bitmap on its way
    uint32_t pack_rounds = 16;     // tile size (x 256 records) of the last pack_count: its table layout (kernels_shard.hip)
    bool pack_list = false, pack_travellers = false;  // how the last pack_count addressed the records / whether it left travellers to shade   // gsx_shard_repair_count has left masks / table / travellers for the repair pack

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
        aos_layout((int)sh_kind, (int)cov_kind, &p.aos_stride, &p.aos_geo);
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
        r.rect8 = rect8_active ? rect8.as<uint32_t>() : nullptr;
        r.code8 = code8_active ? code8.as<uint8_t>() : nullptr;
        return r;
    }
    Records imp_rec() const {
        Records r;
        r.key = imp_key.as<uint32_t>();
        r.a = imp_a.as<float4>();
        r.b = imp_b.as<float4>();
        r.c = imp_c.as<float4>();
        r.rect8 = nullptr;
        r.code8 = nullptr;
        return r;
    }
    Records rec() const { return use_imported ? imp_rec() : proj_rec(); }  // the frame's active record set
};

// A frame-level entry point records its kernel launches and submits them as cached, patched HIP graphs (gsx_launch.h,
// gsx_graph.cpp).  scope_id: which entry point (the cache is per entry point and segment position).
enum : uint32_t { TRACE_PREPROCESS = 1, TRACE_SORT = 2, TRACE_RENDER = 3, TRACE_RENDER_FRAME = 4, TRACE_SHARD = 5 };
struct TraceScope {
    LaunchTrace* mine = nullptr;
    TraceScope(gsx_viewer* v, uint32_t scope_id);
    ~TraceScope();
    // Submits what is still recorded and reports the first launch of this scope that failed (a deferred launch cannot fail where its
    // wrapper returned): GSX_OK, or GSX_ERR_HIP with the HIP error's name.  The destructor does the same and can only drop the error.
    gsx_status finish();
    TraceScope(const TraceScope&) = delete;
    TraceScope& operator=(const TraceScope&) = delete;
};
bool launch_graphs_enabled();
void trace_destroy(LaunchTrace* t);
void trace_stats(const LaunchTrace* t, gsx_launch_stats* out);

struct PassTimer {
    hipEvent_t start, stop;
    int pass;
};

}  // namespace gsx

using namespace gsx;

// What a frame was rendered with (the uniform setters do not complete frames in flight): a frame that has to be redone after the
// caller moved the camera on is redone with ITS uniforms.
struct ShardUniforms {
    float view[16], proj[16];
    uint32_t width, height;
    float size;
    uint32_t display_mode, sh_deg, no_sh0;
    gsx_spec_params params;
    std::vector<ModelTransform> mt;  // per model, in ShardPending::order
};

// An index-sharded frame that is enqueued — every round of every model, the repair rounds included (they decide on the device whether
// they have anything to do) — and whose verdicts have not been read yet (gsx_shard_frame.cpp).
struct ShardPending {
    gsx_viewer* lane = nullptr;
    std::vector<std::string> order;           // the frame's models in COMPOSITING order: nearest first (keys_far_to_near reversed)
    std::vector<uint32_t> shard_max, slot;    // per model: largest shard over the ranks, round-0 slot size in use
    std::vector<uint32_t> vseq;               // per model: its verdict in the lane's ring (shard_post_verdict)
    std::vector<uint32_t> repair_slot;        // per model: slot size of its always-enqueued repair round (0: the frame had none — not limited by windows)
    std::vector<uint8_t> limited;             // per model: the exchange was limited by windows
    uint32_t seq = 0;                         // (the synchronous path: verdict of the model last verified)
    uint32_t speculate = 0, radius = 0;
    float margin = 0.0f;
    bool gathered = false;                    // the band gather that is enqueued shows the final frame
    bool settled = false;                     // every model's verdict has been dealt with already (a frame redone with safe slots)
    bool repaired = false;                    // some model needed its repair exchange
    uint64_t lane_frame = 0;                  // the lane's sharded-frame counter when this frame was enqueued (is it still the lane's newest?)
    ShardUniforms uniforms;
    std::vector<uint32_t> edges;              // the frame's band layout (world + 1 tile rows; empty: equal bands)
    std::vector<std::vector<uint32_t>> pair_caps;  // per model: round-0 slot sizes pair by pair ([s * world + d]; empty: uniform `slot`)
    std::vector<bool> counted;                // per model: its round 0 of THIS frame has been counted (its verdict read: Model::pair_counts)
    // layered frames with frames in flight, model by model (gsx_shard_frame.cpp, frame_step): the models [0, next_model) are enqueued,
    // the verdicts of [0, read_models) have been read (and their repairs, host-decided, exchanged)
    bool stepped = false;
    uint32_t next_model = 0, read_models = 0;
    bool overflowed = false;                  // a slot of a model read so far overflowed: the frame is redone when it is retired
};

namespace gsx {
struct ScopedPass;
}

struct gsx_viewer {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // frames in flight (gsx_render_options::frames_in_flight = L > 1): gsx_render_frame deals frames round-robin to this
    // viewer and L - 1 lanes — viewers of their own (stream, records, sort and tile buffers, framebuffer, speculation
    // state) whose models VIEW this viewer's Gaussian data.  See gsx_api.cpp.
    std::vector<gsx_viewer*> lanes;      // owned
    gsx_viewer* parent = nullptr;        // set in a lane
    gsx_viewer* latest = nullptr;        // the lane that rendered the newest frame (nullptr: this viewer itself)
    uint32_t lane_turn = 0;
    hipEvent_t lane_event = nullptr;     // lane: end of its last frame; parent: "everything enqueued so far" for the lanes to wait on
    bool lane_busy = false;              // lane: it has a frame the parent's stream has not been ordered after
    uint32_t held_w = 0, held_h = 0;     // viewport of the frame this viewer / lane holds (the owner's may have changed since: gsx_debug_download_lane_framebuffer)
    uint64_t epoch = 1, seen_epoch = 0;  // parent: bumped by every call that may touch model data; lane: the epoch it has waited for
    std::vector<hipStream_t> parked_streams;  // owner only: streams that turned out to share a hardware queue with a lane (lane_create)
    uint32_t lane_index = 0;             // 0: the viewer itself; lane i of its parent otherwise
    // sharded frames in flight (gsx_shard_render_frame with frames_in_flight > 1): every lane has a communicator of its own
    // and runs its collectives on its own stream (gsx_comm.cpp)
    std::vector<void*> lane_comms;       // owner only: ncclComm_t of lane 1, 2, ... (lane 0 uses `comm`)
    std::deque<ShardPending> shard_pending;  // owner only: oldest first
    bool shard_busy = false;             // gsx_shard_render_frame is enqueueing / completing: viewer_bind must not complete frames
    uint32_t shard_turn = 0;
    uint64_t shard_frames_enqueued = 0;  // lane (or owner as lane 0): sharded frames enqueued on it so far
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
    unsigned long long* h_verdict = nullptr;  // pinned: {seq << 32 | tiles needing repair}, posted by k_spec_verify (host_verify)
    uint32_t verify_seq = 0;
    uint32_t blocks_max = 256;    // GSX_BLOCKS_MAX: most blocks of a block-list frame (256: one 8-bit sort pass)
    bool blocks_adaptive = true;  // no GSX_BLOCKS_MAX given: 256, or 1024 for models some tile of which walks a long list (Model::blocks_fine_spec / _plain)
    int bin_mode = 1;             // GSX_BIN: 1 block lists for progressive frames (default), 0 per-tile lists always
    bool edit_cache = true;       // GSX_NO_EDIT_CACHE: run k_edit_prepare every frame (tests compare the two)
    uint64_t tile_cap_fixed = 0;  // GSX_TILE_CAP was set when the viewer was created: pair-buffer capacity that never grows (tests of the spill path)
    bool validate = false;  // GSX_VALIDATE was set when the viewer was created: check tile ranges / lists before compositing (debug, synchronous)
    uint32_t band_lo = 0, band_hi = 0xFFFFFFFFu;  // tile rows this viewer renders (gsx_viewer_set_band)
    gsx_query query{};                   // GSX_QUERY_NONE
    DevBuf query_texture;
    uint32_t query_tex_w = 0, query_tex_h = 0;
    float highlight[4]{0, 0, 0, 0};
    gsx_gaussian_edit sel_edit{0u, {0.0f, 1.0f, 1.0f}, 0.0f, 0.0f, 1.0f, 1.0f};
    unsigned long long* h_shard_verdict = nullptr;  // pinned, 2 words: {seq | need}, {busiest pair's records | overflow} (k_shard_verify / k_shard_post_counts)
    uint32_t shard_seq = 0;
    // frames that never ask the host inside a frame (gsx_shard_frame.cpp): every model of every frame posts ONE verdict block, into
    // slot (seq % ring_slots) of a pinned ring — read when the frame is retired, a call or two later
    uint32_t* h_verdict_ring = nullptr;  // pinned: ring_slots x kVerdictWords u32
    uint32_t ring_slots = 0, ring_seq = 0;
    DevBuf verdict_stage;                // device: the round-0 verdict block (k_shard_verify), merged and posted by k_shard_post_verdict
    void* comm = nullptr;                // ncclComm_t (gsx_viewer_comm_init); RCCL is loaded at run time (gsx_comm.cpp)
    uint32_t comm_world = 0, comm_rank = 0;
    bool comm_self_via_rccl = false;     // GSX_COMM_SELF_VIA_RCCL at gsx_viewer_comm_init: a rank's own exchange slot goes through RCCL too
    // a transport other than RCCL (gsx_viewer_comm_init_custom / _init_group): the two collectives as functions that enqueue
    gsx_comm_all_to_all_fn comm_a2a_fn = nullptr;
    gsx_comm_all_gather_fn comm_ag_fn = nullptr;
    void* comm_ctx = nullptr;
    // ... and, where the transport can move pieces of unequal size (gsx_viewer_comm_init_custom_v, the in-process group): then the
    // bands may be balanced and the exchange slots sized pair by pair
    gsx_comm_all_to_all_v_fn comm_a2a_v_fn = nullptr;
    gsx_comm_gather_v_fn comm_gather_v_fn = nullptr;
    gsx_comm_group* comm_group = nullptr;  // != nullptr: comm_ctx is this viewer's seat in that group (gsx_comm_group.cpp)
    gsx_shard_stats shard_stats{};       // host-side bookkeeping of the sharded frames (gsx_shard_get_stats)
    int32_t shard_gather_root = -1;  // gsx_shard_set_gather_root: -1 every rank receives every band, >= 0 only that rank
    // band layout of the sharded frames (BandEdges, gsx_internal.h).  band_edges: world + 1 tile rows, or empty = equal bands.
    // gsx_shard_render_frame sets it per frame — from next_edges, which the last completed frame's verdict posted (balanced by
    // that frame's per-row work) — when the transport moves pieces of unequal size; gsx_shard_set_band_edges sets it for callers
    // of the stage functions.
    std::vector<uint32_t> band_edges, next_edges, band_edges_forced, last_edges;
    bool shard_root_confirmed = false;   // a verdict since the last gsx_shard_set_gather_root has shown that every rank names the same root
    uint32_t next_edges_tiles_y = 0;     // the grid next_edges was made for
    bool shard_balance = true;           // gsx_shard_set_balance
    bool shard_pair_slots = true;        // size the exchange slots pair by pair (where the transport moves unequal pieces)
    DevBuf shard_fb, shard_send, shard_recv, shard_sat_band, shard_sat_all, shard_counts;  // gsx_shard_render_frame's own buffers
    void* ext_fb = nullptr;              // caller-owned framebuffer (multi-GPU: the RCCL gather target)
    uint64_t ext_fb_bytes = 0;
    gsx_render_options options{1u, 16u, 131072u, 2u, 1u, 0.25f, 3u, 0u, 1u, 1u};  // = gsx_render_options_default (a CPU test compares the two: gsx_viewer_get_render_options)
    bool host_waited = false;  // the host has waited for this viewer's device work (gsx_sync, a blocking readback) since its last frame was enqueued:
                               // the app synchronises per frame, so asking for a speculated frame's verdict costs it nothing (host_verify = 2)
    uint32_t timing = 0;  // bit p: bracket pass p with events
    gsx::ScopedPass* open_pass = nullptr;  // the innermost pass scope open on this viewer (ScopedPass: scopes nest)
    std::vector<PassTimer> timers;     // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
    float pass_ms[GSX_PASS_COUNT]{};
    uint32_t pass_launches[GSX_PASS_COUNT]{};
    DevBuf tile_prof;                    // GSX_TILE_PROFILE: what every tile of the LAST block-compositor launch of a frame's first slab cost
    bool tile_profile = false;
    bool bin_fused = true;               // GSX_BIN_FUSED=0: block binning as count + scan + emit + histogram launches (A/B)
    bool bucket_sort = true;             // GSX_BUCKET_SORT=0: speculated frames, repair rounds and imported bands keep the five-launch LSD depth sort (A/B)
    bool tile_order_on = true;           // GSX_TILE_ORDER=0: the block compositor takes its tiles in index order (A/B)
    int sorted_records = -1;             // GSX_SORTED_RECORDS=0 / 1: never / always carry the block lists' records through the block sort (-1: by list length)
    gsx::LaunchTrace* trace = nullptr;  // owned; created by the first TraceScope on this viewer (gsx_graph.cpp)
};

namespace gsx {

inline Model* find_model(gsx_viewer* v, const char* key) {
    if (!v || !key) return nullptr;
    auto it = v->models.find(key);
    return it == v->models.end() ? nullptr : it->second.get();
}

struct ScopedPass {
    // Brackets a pass with a pair of events.  Scopes nest (the slab's shading inside the binning scope, the shading of admitted
    // records inside the depth sort's): an inner scope SUSPENDS the outer one — the outer interval is closed where the inner one
    // begins and a new one opens where it ends — so every microsecond is counted for exactly one pass.
    gsx_viewer* v;
    int pass;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedPass* outer = nullptr;
    bool on = false;
    void open() {
        if (!v->event_pool.empty()) {
            a = v->event_pool.back().first;
            b = v->event_pool.back().second;
            v->event_pool.pop_back();
        } else {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
        }
        (void)gsx::op::EventRecord(a, v->stream);
    }
    void close() {
        if (!a) return;
        (void)gsx::op::EventRecord(b, v->stream);
        v->timers.push_back({a, b, pass});
        a = b = nullptr;
    }
    ScopedPass(gsx_viewer* v_, int pass_) : v(v_), pass(pass_) {
        outer = v->open_pass;
        v->open_pass = this;
        on = ((v->timing >> pass) & 1u) != 0;
        if (outer) outer->close();
        if (on) open();
    }
    ~ScopedPass() {
        close();
        v->open_pass = outer;
        if (outer && outer->on) outer->open();
    }
    ScopedPass(const ScopedPass&) = delete;
    ScopedPass& operator=(const ScopedPass&) = delete;
};

inline uint32_t ceil_log2(uint32_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

// Every entry point but gsx_render_frame and the uniform setters comes through here.  With frames in flight, the caller
// may be about to read results or to change model data the lanes are still reading: the viewer's stream is ordered after
// the lanes' frames (no host wait), and the lanes' next frames after whatever the caller enqueues (epoch).
gsx_status shard_complete_pending(gsx_viewer* v);  // gsx_comm.cpp: verdicts, redo / repair rounds of the sharded frames in flight
inline bool has_comm(const gsx_viewer* v) { return v->comm != nullptr || v->comm_a2a_fn != nullptr || v->comm_a2a_v_fn != nullptr; }
void group_leave(gsx_viewer* v);  // gsx_comm_group.cpp: give this viewer's seat in its in-process group back
// The collectives of a sharded frame with pieces of unequal size (gsx_comm.cpp): per peer p, `bytes[p]` bytes at `off[p]`.
struct PeerSpans {
    uint64_t off[kMaxRanks], bytes[kMaxRanks];
};
inline bool comm_moves_unequal(const gsx_viewer* owner) { return owner->comm != nullptr || owner->comm_a2a_v_fn != nullptr; }
// the slot-based stage calls with slots of any size (gsx_api_shard.cpp; the exported ones pass uniform slots)
gsx_status shard_pack_slots(gsx_viewer* v, const char* key, uint32_t world, uint32_t round, void* d_send, const SlotSpans& slots, bool gated = false);
// a frame that decides its repair round on the device (gsx_shard_frame.cpp): the round-0 verdict staged in device memory, then — behind
// the always-enqueued repair round — posted to slot seq % ring of the viewer's pinned verdict ring
gsx_status shard_verify_staged(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all);
gsx_status shard_feedback(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, void* d_out_u32, bool zero_verify_state);
gsx_status shard_post_verdict(gsx_viewer* v, uint32_t world, const void* d_sat_after_repair /* nullable */, uint32_t* out_seq);
gsx_status shard_next_windows_post(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, float margin, uint32_t radius,
                                   const void* d_sat_after_repair /* nullable */, uint32_t* out_seq);
gsx_status shard_wait_ring(gsx_viewer* v, uint32_t seq, gsx_shard_verdict* out, const uint32_t** block);  // spins on the ring slot of seq
gsx_status shard_ensure_ring(gsx_viewer* v, uint32_t verdicts_outstanding);
gsx_status shard_import_slots(gsx_viewer* v, const char* key, const void* d_recv, uint32_t world, uint32_t rank, uint32_t round_flags, const SlotSpans& slots);
// slot p of the send buffer (snd) goes to rank p, what rank p sends lands in slot p of the receive buffer (rcv)
gsx_status comm_all_to_all_v(gsx_viewer* v, const void* d_send, const PeerSpans& snd, void* d_recv, const PeerSpans& rcv);
// every rank's piece (send_bytes of it; rank p's lands at rcv.off[p]) to every rank (root < 0) or to `root` only
gsx_status comm_gather_v(gsx_viewer* v, const void* d_send, uint64_t send_bytes, void* d_recv, const PeerSpans& rcv, int32_t root);
gsx_status comm_ensure_lanes(gsx_viewer* v, uint32_t lanes);  // gsx_comm.cpp: one RCCL communicator per lane (collective)

inline gsx_status viewer_bind(gsx_viewer* v) {
    if (!v) return fail(GSX_ERR_INVALID_ARG, "viewer is null");
    HIPCHK(hipSetDevice(v->device));
    if (!v->shard_pending.empty() && !v->shard_busy) {
        gsx_status pst = shard_complete_pending(v);
        if (pst) return pst;
    }
    if (!v->lanes.empty() && !v->shard_busy) {  // (inside gsx_shard_render_frame the lanes' frames stay in flight)
        for (gsx_viewer* l : v->lanes)
            if (l->lane_busy) {
                HIPCHK(gsx::op::StreamWaitEvent(v->stream, l->lane_event, 0));
                l->lane_busy = false;
            }
        v->epoch += 1;
    }
    return GSX_OK;
}

inline gsx_status ensure_fb(gsx_viewer* v) {
    if (v->ext_fb) {
        if (v->ext_fb_bytes < sizeof(float4) * (size_t)v->width * v->height)
            return fail(GSX_ERR_INVALID_ARG, "external framebuffer of %llu bytes is too small for %ux%u", (unsigned long long)v->ext_fb_bytes, v->width, v->height);
        return GSX_OK;
    }
    HIPCHK(v->fb.ensure(sizeof(float4) * (size_t)v->width * v->height));
    return GSX_OK;
}
// readback entry points refer to the newest frame, whichever lane rendered it
inline gsx_viewer* result_lane(gsx_viewer* v) { return v->latest ? v->latest : v; }
inline float4* fb_ptr(gsx_viewer* v) { return v->ext_fb ? static_cast<float4*>(v->ext_fb) : reinterpret_cast<float4*>(v->fb.p); }

// lane `index` (0 = the viewer itself) brought up to date for a frame of `keys` (gsx_api.cpp)
gsx_status lane_acquire(gsx_viewer* v, uint32_t index, const char* const* keys, uint32_t n_keys, gsx_viewer** out);
bool shard_frame_may_use_lanes(gsx_viewer* v, const char* const* keys, uint32_t n_keys);  // gsx_api.cpp

// ---- frame scheduling (gsx_frame.cpp) ----
// defer_visible_count: a gsx_sort of this model follows at once (gsx_render_frame): its admission scan sums N_vis
gsx_status do_preprocess(gsx_viewer* v, Model* m, bool defer_visible_count = false);
// force_full: ignore the admission the projection pass made (a speculated frame being redone) and sort every visible record
gsx_status do_sort(gsx_viewer* v, Model* m, bool force_full = false);
// cont: a second round of the same frame (multi-GPU repair exchange): keep the framebuffer and the saturated-tile state
gsx_status do_render(gsx_viewer* v, const char* const* keys, uint32_t n_keys, bool cont = false);
gsx_status finish_frame(gsx_viewer* v);
inline gsx_status sync_counters(gsx_viewer* v) { return finish_frame(v); }
gsx_status complete_records(gsx_viewer* v, Model* m);
gsx_status prepare_edits_for_lanes(gsx_viewer* v, const char* const* keys, uint32_t n_keys);  // gsx_api.cpp: before a frame is dealt to a lane
bool edits_need_prepare(const gsx_viewer* v, const Model* m);          // gsx_frame.cpp: this frame runs an edit pass and k_edit_prepare's inputs changed
gsx_status prepare_edits(gsx_viewer* v, Model* m, bool* launched);  // gsx_frame.cpp
// k_shade over a list of admitted records of a lazily projected frame, then the frame's colour ops on exactly those (gsx_frame.cpp)
gsx_status shade_admitted(gsx_viewer* v, Model* m, const LateProjection& late);
gsx_status ensure_record_capacity(Model* m, uint64_t count);
gsx_status ensure_sortbin_capacity(Model* m, uint64_t count);
gsx_status ensure_import_capacity(Model* m, uint64_t count);
gsx_status ensure_selection(gsx_viewer* v, Model* m);
gsx_status ensure_edit_buffers(gsx_viewer* v, Model* m);

// tile rows per rank of the multi-GPU layout with equal bands, the layout in force, bytes of a per-tile window map
inline uint32_t rows_per_rank(const gsx_viewer* v, uint32_t world) {
    const uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    return (tiles_y + world - 1) / world;
}
inline BandEdges bands_of(const gsx_viewer* v, uint32_t world) {
    BandEdges b{};
    b.world = world;
    // (a lane renders with the layout its owner's frame was given: gsx_shard_frame.cpp copies it)
    if (v->band_edges.size() == (size_t)world + 1u) {
        for (uint32_t g = 0; g <= world; ++g) b.e[g] = v->band_edges[g];
    } else {
        const uint32_t rpr = rows_per_rank(v, world);
        for (uint32_t g = 0; g <= world; ++g) b.e[g] = g * rpr;
    }
    return b;
}
// A band layout set for another viewport (gsx_shard_set_band_edges validates against the height at call time; gsx_update_camera may
// enlarge the viewport later): the last edge must still cover every tile row, or no rank would own the bottom rows and the gathered
// frame would silently keep stale pixels there.  Every stage call that reads the layout checks it.
inline gsx_status check_bands(const gsx_viewer* v, uint32_t world, const char* who) {
    const uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    if (v->band_edges.size() == (size_t)world + 1u && v->band_edges[world] < tiles_y)
        return fail(GSX_ERR_INVALID_ARG, "%s: the band edges in force end at tile row %u but the viewport has %u rows (gsx_shard_set_band_edges was called "
                    "for a smaller viewport: set them again, or clear them with edges = NULL)", who, v->band_edges[world], tiles_y);
    return GSX_OK;
}
inline size_t window_bytes(const gsx_viewer* v) {
    return sizeof(uint2) * (size_t)((v->width + GSX_TILE - 1) / GSX_TILE) * ((v->height + GSX_TILE - 1) / GSX_TILE);
}

}  // namespace gsx
