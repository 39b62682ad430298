// gsx_shard_frame.cpp — gsx_shard_render_frame / gsx_shard_render_frame_keys: one index-sharded frame of this rank as ONE call
// of the C ABI, strung together from the stage calls of gsx_api_shard.cpp and the collectives of gsx_comm.cpp.  No reference
// counterpart (src/main.rs:85-98: one wgpu device); what it must reproduce is the single-GPU frame — for several models the
// reference's layering: painted far -> near, never merged (src/tab/scene.rs:533-558, 2302-2314).
//
// The frame, per model in COMPOSITING order (nearest first; the far -> near key list walked backwards, as gsx_render does):
//   gsx_shard_frame_begin of every model (projection; windows from the model's own limits of its last frame), then for each model:
//   round 0   pack -> all-to-all of fixed slots -> import + depth sort + composite into this rank's band (behind the nearer
//             models) -> feedback -> all-gather of the saturation map -> verification: which tiles were refused records they
//             still need, their repair windows, the count — all on the device (the verdict block staged in device memory).
//   verdict   one kernel posts the model's verdict block — tiles that needed the repair, a slot overflowed, the count matrix, next
//             frame's band edges — into slot (seq mod ring) of a pinned ring; next limits from the gathered saturation map.
//   round 1   the repair exchange, decided by the HOST from the verdict block: exactly sized, only where a tile needs it.
//             * ONE model: the whole frame is enqueued and the call returns; the verdict is read when the frame is retired (below) —
//               with frames in flight that look is off the critical path, and a frame that repairs nothing pays nothing.
//             * LAYERED models go out model by model (frame_step): the verdict of model i - 1 is read before model i is enqueued and
//               its repair exchanged behind model i - 1, in front of model i.  With frames in flight a call alternates between what is
//               left of the frame before and the new frame, one model per turn: the verdict a turn waits for belongs to a model that
//               went out a turn ago and the other frame's model keeps the device busy; the new frame is left with half its models out.
//             * the alternative, built first in round 5 and kept as the A/B (GSX_SHARD_LAYER_PIPELINE=0
//               for the last model too): all models at once, the repair exchange of every model that has another BEHIND it always
//               enqueued with slots of a fixed size R (twice the largest repair of the last frames, a maximum that decays by a
//               sixteenth per frame) and decided on the device — its kernels fall through when no tile needs anything, the all-to-all
//               moves the (empty) slots regardless.  No host look anywhere; ~22 fall-through launches per inner model and `world`
//               empty slots per round.  Measured slower on every count: cfg5 at world 1 396 -> 489 fps with one frame in flight,
//               497 -> 605 with two; cfg4 with the device deciding for the single model too 1100 / 1200 against 1450 / 1690.
//   then the in-place band gather.
// The verdicts (of a layered frame: the last model's) are read when the frame is RETIRED: after frame k is enqueued, every frame but the newest L - 1 is retired
// (L = gsx_render_options.frames_in_flight), on every rank alike — what they say (slot sizes pair by pair, band edges, the repair
// slot size) is what every rank plans the next frame with.  With L >= 2 the verdict of frame k - L + 1 arrived while frame k was
// being enqueued and the device has L - 1 frames queued meanwhile: nothing waits.  With L = 1 the call waits for its own frame's
// verdict — once per frame, whatever the number of models and whether or not anything was repaired (until round 5: once per model,
// twice where a repair was sized).
// A slot that overflowed (round 0: the camera jumped; round 1: more repairs than R — on cfg4's orbit they come in bursts: a few
// hundred tiles looking through a hole want 1.5 M records behind their limits, five frames after 5 000 were plenty) leaves the frame
// incomplete; its retirement redoes it, synchronously, by the host-decided path of rounds 2-4 (the whole frame with exactly sized or
// whole-shard slots, an exactly sized repair round) on its own lane, with its own uniforms and limits — a lane is never reused, and
// no frame is ever read (viewer_bind retires the frames in flight), before its frame is complete.
//
// Every branch below is taken on data that was gathered from all ranks (the verdict words), so every rank takes the same
// branches and issues the same collectives in the same order: nothing can cross.
#include <chrono>
#include <cstdlib>

#include "gsx_state.h"

namespace {

struct Ctx {
    gsx_viewer* owner;   // holds the communicator(s) and the statistics
    gsx_viewer* l;       // the lane this frame runs on (the owner itself with one frame in flight)
    ShardPending* p;
    uint32_t world, rank;
    gsx_shard_layout_t lay;
    uint32_t sat_words;  // stride of the gathered feedback pieces
    BandEdges bands;     // this frame's layout (ShardPending::edges; the lane's band_edges while the frame is worked on)
    uint32_t tiles_x, tiles_y;
};

// The frame's band layout.  Equal bands unless the transport moves pieces of unequal size; then the edges the caller forced
// (gsx_shard_set_band_edges), or the ones the last COMPLETED frame's verdict posted — balanced by that frame's per-row work, the
// same on every rank because the verdict is derived from gathered data and every rank completes its frames in the same order.
void choose_edges(gsx_viewer* owner, ShardPending& p, uint32_t world, uint32_t tiles_y) {
    p.edges.clear();
    if (!comm_moves_unequal(owner)) return;
    if (owner->band_edges_forced.size() == (size_t)world + 1u) {
        p.edges = owner->band_edges_forced;  // (a layout that no longer covers the viewport fails the frame: check_bands in frame_buffers)
    } else if (owner->shard_balance && owner->next_edges.size() == (size_t)world + 1u && owner->next_edges_tiles_y == tiles_y) {
        p.edges = owner->next_edges;
    }
}

// the padded framebuffer the bands are gathered into (owned by the library, per lane) and the saturation-map buffers
gsx_status frame_buffers(Ctx& c) {
    gsx_viewer* v = c.l;
    v->band_edges = c.p->edges;  // (empty: equal bands) — every stage call below reads the layout from the viewer it runs on
    c.tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE;
    c.tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    c.bands = bands_of(v, c.world);
    gsx_status st = gsx_shard_layout(v, c.world, c.rank, &c.lay);
    if (st) return st;
    // (whole tile rows of the equal layout AND of any balanced one: the layout may change from frame to frame, the buffer stays)
    const uint64_t fb_bytes = std::max<uint64_t>(c.lay.padded_framebuffer_bytes,
                                                 (uint64_t)std::max(c.world * rows_per_rank(v, c.world), c.tiles_y) * GSX_TILE * v->width * sizeof(float4));
    if (v->ext_fb != v->shard_fb.p || v->shard_fb.bytes < fb_bytes) {
        HIPCHK(gsx::op::StreamSynchronize(v->stream));
        if (v->shard_fb.bytes < fb_bytes) {
            HIPCHK(v->shard_fb.ensure(fb_bytes));
            HIPCHK(gsx::op::MemsetAsync(v->shard_fb.p, 0, v->shard_fb.bytes, v->stream));
        }
        v->ext_fb = v->shard_fb.p;
        v->ext_fb_bytes = v->shard_fb.bytes;
    }
    if ((st = gsx_shard_feedback_words(v, c.world, &c.sat_words))) return st;
    HIPCHK(v->shard_sat_band.ensure(4 * (size_t)c.sat_words + 16));
    HIPCHK(v->shard_sat_all.ensure((4 * (size_t)c.sat_words + 16) * c.world));
    return GSX_OK;
}

// every rank's feedback piece to every rank: at the common stride, each as long as its band needs
gsx_status feedback_gather(Ctx& c) {
    gsx_viewer* v = c.l;
    if (!comm_moves_unequal(c.owner)) return gsx_comm_all_gather(v, v->shard_sat_band.p, v->shard_sat_all.p, 4 * (uint64_t)c.sat_words);
    PeerSpans sp{};
    for (uint32_t g = 0; g < c.world; ++g) {
        sp.off[g] = 4ull * g * c.sat_words;
        sp.bytes[g] = 4ull * feedback_words(c.bands, c.tiles_x, g);
    }
    return comm_gather_v(v, v->shard_sat_band.p, sp.bytes[c.rank], v->shard_sat_all.p, sp, -1);
}

// one exchange round of model i: pack -> all-to-all -> import + sort + composite -> feedback -> all-gather.
// T: uniform slots of T records; caps (round 0, nullable): caps[s * world + d] = records the slot of the pair (s, d) holds — sized
// pair by pair from the count matrix of the model's last frame, the same table on every rank.
gsx_status exchange_round(Ctx& c, size_t i, uint32_t round, uint32_t T, const std::vector<uint32_t>* caps = nullptr, bool gated = false, bool zero_verify = false) {
    gsx_viewer* v = c.l;
    const char* key = c.p->order[i].c_str();
    gsx_shard_stats& ss = c.owner->shard_stats;
    ss.exchange_rounds += 1;
    SlotSpans snd = uniform_slots(c.world, T), rcv = snd;
    if (caps && caps->size() == (size_t)c.world * c.world) {
        uint32_t so = 0, ro = 0;
        T = 0;
        for (uint32_t p = 0; p < c.world; ++p) {
            snd.off[p] = so;
            snd.cap[p] = (*caps)[(size_t)c.rank * c.world + p];
            so += snd.cap[p] + 1u;
            rcv.off[p] = ro;
            rcv.cap[p] = (*caps)[(size_t)p * c.world + c.rank];
            ro += rcv.cap[p] + 1u;
            T = std::max(T, snd.cap[p]);
        }
    }
    (round == 0 ? ss.last_slot_records : ss.last_repair_slot_records) = T;
    const uint32_t last = c.world - 1u;
    HIPCHK(v->shard_send.ensure((uint64_t)(snd.off[last] + snd.cap[last] + 1u) * GSX_RECORD_BYTES));
    HIPCHK(v->shard_recv.ensure((uint64_t)(rcv.off[last] + rcv.cap[last] + 1u) * GSX_RECORD_BYTES));
    gsx_status st;
    if ((st = shard_pack_slots(v, key, c.world, round, v->shard_send.p, snd, gated))) return st;
    if (comm_moves_unequal(c.owner)) {
        PeerSpans bs{}, br{};
        for (uint32_t p = 0; p < c.world; ++p) {
            bs.off[p] = (uint64_t)snd.off[p] * GSX_RECORD_BYTES;
            bs.bytes[p] = (uint64_t)(snd.cap[p] + 1u) * GSX_RECORD_BYTES;
            br.off[p] = (uint64_t)rcv.off[p] * GSX_RECORD_BYTES;
            br.bytes[p] = (uint64_t)(rcv.cap[p] + 1u) * GSX_RECORD_BYTES;
        }
        st = comm_all_to_all_v(v, v->shard_send.p, bs, v->shard_recv.p, br);
    } else {
        st = gsx_comm_all_to_all(v, v->shard_send.p, v->shard_recv.p, (uint64_t)(T + 1u) * GSX_RECORD_BYTES);
    }
    if (st) return st;
    if ((st = shard_import_slots(v, key, v->shard_recv.p, c.world, c.rank, round | (i > 0 ? GSX_SHARD_BEHIND : 0u), rcv))) return st;
    // (round 0 of a frame that decides on the device: the verification follows the gather at once — its state is zeroed by this kernel)
    if ((st = shard_feedback(v, key, c.world, c.rank, v->shard_sat_band.p, zero_verify))) return st;
    return feedback_gather(c);
}

// Slots pair by pair: what the pair (s, d) wanted in round 0 of the model's last frame, a quarter more, and 512 records (the
// orbit moves the counts by a few per cent a frame; a camera jump overflows a slot, the verdict says so and the frame is redone
// with whole-shard slots — the same frame, the same pixels).  Uniform slots hold the BUSIEST pair's count x 2 for every pair:
// 19.4 MB per rank and frame on the links at 8 ranks on cfg4, of which 2.7 MB were records somebody wanted.  Needs a transport
// that moves unequal pieces and a matrix from a frame of the same kind (both limited by windows, or both not) and the same bands
// (a frame whose bands have just moved falls back to uniform slots once).
void plan_pair_slots(Ctx& c, size_t i) {
    ShardPending& p = *c.p;
    p.pair_caps[i].clear();
    gsx_viewer* o = c.owner;
    if (!comm_moves_unequal(o) || !o->shard_pair_slots || c.world < 2) return;
    const Model* om = find_model(o, p.order[i].c_str());
    const Model* lm = find_model(c.l, p.order[i].c_str());
    if (!om || !lm || om->slot_force || om->pair_counts.size() != (size_t)c.world * c.world || om->pair_limited != (p.limited[i] != 0)) return;
    if (om->pair_edges.size() != (size_t)c.world + 1u || !std::equal(om->pair_edges.begin(), om->pair_edges.end(), c.bands.e)) return;  // counted under other bands
    const uint32_t n = std::max<uint32_t>(p.shard_max[i], 1u);
    p.pair_caps[i].resize((size_t)c.world * c.world);
    for (size_t k = 0; k < p.pair_caps[i].size(); ++k) {
        const uint64_t want = om->pair_counts[k];
        p.pair_caps[i][k] = (uint32_t)std::min<uint64_t>(n, want + want / 4u + 512u);
    }
}

gsx_status next_limits(Ctx& c, size_t i) {
    return gsx_shard_next_windows(c.l, c.p->order[i].c_str(), c.world, c.l->shard_sat_all.p, c.p->margin, c.p->radius);
}

// the bands, in place: every rank's band lands where it belongs in every rank's framebuffer (or in the root's only)
gsx_status band_gather(Ctx& c) {
    char* fb = static_cast<char*>(c.l->ext_fb);
    const int32_t root = c.owner->shard_gather_root;
    gsx_status st;
    if (!comm_moves_unequal(c.owner)) {  // equal pieces only: the transport's all-gather, whatever the root
        st = gsx_comm_all_gather(c.l, fb + c.lay.band_offset_bytes, fb, c.lay.band_bytes);
    } else {
        const uint64_t row_bytes = (uint64_t)GSX_TILE * c.l->width * sizeof(float4);
        PeerSpans sp{};
        for (uint32_t g = 0; g < c.world; ++g) {
            sp.off[g] = c.bands.e[g] * row_bytes;
            sp.bytes[g] = (c.bands.e[g + 1] - c.bands.e[g]) * row_bytes;
        }
        st = comm_gather_v(c.l, fb + sp.off[c.rank], sp.bytes[c.rank], fb, sp, root);
    }
    c.p->gathered = st == GSX_OK;
    return st;
}

// what a verdict block carries besides its two words (kernels_shard.hip, k_shard_verify / k_shard_post_verdict): do the ranks agree
// about the gather root and about how they size slots and bands, the band edges for the frames to come, the count matrix of the exchange
gsx_status read_block(Ctx& c, size_t i, const uint32_t* hv) {
    gsx_viewer* o = c.owner;
    if (hv[4] & 1u) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: the ranks name different gather roots (gsx_shard_set_gather_root: this rank %d)", (int)o->shard_gather_root);
    if (hv[4] & 2u) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: the ranks size their exchange slots or bands differently (gsx_shard_set_balance, "
                                "gsx_shard_set_slot_records must be the same on every rank)");
    o->shard_root_confirmed = true;
    o->next_edges.assign(hv + kVerdictEdges, hv + kVerdictEdges + c.world + 1);
    o->next_edges_tiles_y = c.tiles_y;
    if (Model* om = find_model(o, c.p->order[i].c_str())) {
        om->pair_counts.assign(hv + kVerdictMatrix, hv + kVerdictMatrix + (size_t)c.world * c.world);
        om->pair_limited = c.p->limited[i] != 0;
        om->pair_edges.assign(c.bands.e, c.bands.e + c.world + 1);
    }
    o->shard_stats.last_entries_sum = hv[5];
    o->shard_stats.last_entries_max = hv[6];
    o->shard_stats.last_work_permille = hv[7];
    return GSX_OK;
}

// ---- the synchronous path: a frame somebody is looking at whose slots overflowed is redone with the host deciding ----
gsx_status round0_sync(Ctx& c, size_t i) {
    gsx_status st = exchange_round(c, i, 0, c.p->slot[i], &c.p->pair_caps[i]);
    if (st) return st;
    if ((st = gsx_shard_verify(c.l, c.p->order[i].c_str(), c.world, c.l->shard_sat_all.p, &c.p->seq))) return st;
    return next_limits(c, i);  // what follows when nothing needs a repair (redone after one)
}

gsx_status timed_wait(Ctx& c, const char* key, uint32_t seq, gsx_shard_verdict* out) {
    const auto t0 = std::chrono::steady_clock::now();
    const gsx_status st = gsx_shard_wait_verdict(c.l, key, seq, out);
    c.owner->shard_stats.verdict_wait_ns +=
        (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return st;
}

gsx_status repair_sized(Ctx& c, size_t i);

// the verdict of model i's round 0 and what it asks for; *overflow: a slot was too small, the frame has to be redone
gsx_status settle(Ctx& c, size_t i, bool* overflow) {
    const char* key = c.p->order[i].c_str();
    gsx_shard_verdict verdict{};
    gsx_status st = timed_wait(c, key, c.p->seq, &verdict);
    if (st) return st;
    if ((st = read_block(c, i, reinterpret_cast<const uint32_t*>(c.l->h_shard_verdict)))) return st;
    if (getenv("GSX_SHARD_DEBUG"))
        fprintf(stderr, "[gsx shard] rank %u   redo: model '%s' (slot %u): need %u, overflow %u, busiest pair wanted %u, entries %u\n", c.rank, key, c.p->slot[i],
                verdict.need_tiles, verdict.overflow, verdict.max_records, reinterpret_cast<const uint32_t*>(c.l->h_shard_verdict)[5]);
    c.p->counted[i] = true;
    *overflow = verdict.overflow != 0;
    if (*overflow || !verdict.need_tiles) return GSX_OK;
    return repair_sized(c, i);
}

// the repair round of model i, sized exactly by the host: what each destination is owed is counted, the maximum gathered and posted
gsx_status repair_sized(Ctx& c, size_t i) {
    gsx_viewer* v = c.l;
    const char* key = c.p->order[i].c_str();
    gsx_status st = GSX_OK;
    c.p->repaired = true;
    HIPCHK(v->shard_counts.ensure(16 * (size_t)(c.world + 1)));
    char* cnt = static_cast<char*>(v->shard_counts.p);
    if ((st = gsx_shard_repair_count(v, key, c.world, cnt + 16 * (size_t)c.world))) return st;
    if ((st = gsx_comm_all_gather(v, cnt + 16 * (size_t)c.world, cnt, 16))) return st;
    uint32_t seq = 0;
    if ((st = gsx_shard_post_counts(v, c.world, cnt, &seq))) return st;
    gsx_shard_verdict sized{};
    if ((st = timed_wait(c, nullptr, seq, &sized))) return st;
    if (getenv("GSX_SHARD_DEBUG")) fprintf(stderr, "[gsx shard] rank %u   redo: model '%s' repair sized %u\n", c.rank, key, sized.max_records);
    if ((st = exchange_round(c, i, 1, std::max<uint32_t>(sized.max_records, 1u)))) return st;
    c.p->gathered = false;
    return next_limits(c, i);
}

// The whole frame once more with slots that cannot overflow; every verdict is dealt with at once.  Slots of a whole shard always
// fit — and move 420 MB per rank at 8 ranks on cfg4.  But the verdict that reported the overflow also carries what every pair WANTED
// in this very round (the count matrix), the redo packs the same records through the same windows, so where the transport moves
// unequal pieces the redo's slots are exactly that: a few megabytes.
gsx_status redo_safe(Ctx& c) {
    c.owner->shard_stats.redo_frames += 1;
    // attempt 0: exact slots where this frame's counts are known; attempt 1: slots of a whole shard (they cannot overflow)
    for (int attempt = 0; attempt < 2; ++attempt) {
        c.p->gathered = false;
        bool any_exact = false;
        for (size_t i = 0; i < c.p->order.size(); ++i) {
            c.p->slot[i] = std::max<uint32_t>(c.p->shard_max[i], 1u);
            c.p->pair_caps[i].clear();
            const Model* om = find_model(c.owner, c.p->order[i].c_str());
            // (models behind the one that overflowed have not been exchanged in this frame yet: nothing is known about them)
            if (attempt == 0 && comm_moves_unequal(c.owner) && c.owner->shard_pair_slots && om && !om->slot_force && c.p->counted[i] &&
                om->pair_counts.size() == (size_t)c.world * c.world) {
                c.p->pair_caps[i] = om->pair_counts;
                any_exact = true;
            }
        }
        if (attempt == 0 && !any_exact) continue;
        bool again = false;
        for (size_t i = 0; i < c.p->order.size() && !again; ++i) {
            gsx_status st = round0_sync(c, i);  // (model 0 is not "behind": it starts from a cleared band)
            if (st) return st;
            bool overflow = false;
            if ((st = settle(c, i, &overflow))) return st;
            if (overflow) {
                if (attempt == 0) {  // (the counts moved between the two attempts: the safe size, once more)
                    c.owner->shard_stats.redo_fallbacks += 1;
                    again = true;
                } else {
                    return fail(GSX_ERR_OOM, "gsx_shard_render_frame: an exchange slot of %u records (a whole shard of '%s') overflowed: shard_records_max is wrong",
                                c.p->slot[i], c.p->order[i].c_str());
                }
            }
        }
        if (!again) break;
    }
    c.p->settled = true;
    return GSX_OK;
}

// ---- the frame: everything enqueued, nothing waited for ----

void snapshot_uniforms(const gsx_viewer* v, ShardPending& p) {
    ShardUniforms& u = p.uniforms;
    memcpy(u.view, v->view, sizeof u.view);
    memcpy(u.proj, v->proj, sizeof u.proj);
    u.width = v->width;
    u.height = v->height;
    u.size = v->size;
    u.display_mode = v->display_mode;
    u.sh_deg = v->sh_deg;
    u.no_sh0 = v->no_sh0;
    u.params = v->params;
    u.mt.clear();
    for (const std::string& k : p.order) {
        const Model* m = find_model(const_cast<gsx_viewer*>(v), k.c_str());
        u.mt.push_back(m ? m->mt : ModelTransform{});
    }
}

// the lane's uniforms <-> the frame's (a frame that is redone after the caller moved the camera on: redone as it was asked for)
void swap_uniforms(gsx_viewer* v, ShardPending& p) {
    ShardUniforms& u = p.uniforms;
    for (int k = 0; k < 16; ++k) {
        std::swap(u.view[k], v->view[k]);
        std::swap(u.proj[k], v->proj[k]);
    }
    std::swap(u.width, v->width);
    std::swap(u.height, v->height);
    std::swap(u.size, v->size);
    std::swap(u.display_mode, v->display_mode);
    std::swap(u.sh_deg, v->sh_deg);
    std::swap(u.no_sh0, v->no_sh0);
    std::swap(u.params, v->params);
    for (size_t i = 0; i < p.order.size() && i < u.mt.size(); ++i)
        if (Model* m = find_model(v, p.order[i].c_str())) std::swap(u.mt[i], m->mt);
}

// Slot size of the always-enqueued repair round: twice the most the busiest (rank, destination) pair had in the repair rounds of the
// last frames (a maximum that decays by a sixteenth per frame), and 4096 — a GLOBAL figure (every rank reads the same verdicts in the same order), never more than a
// shard.  A frame that needs more (a camera jump: thousands of tiles repair at once) overflows, says so in its verdict and is redone
// by the host-decided path if anybody looks at it; its verdict also raises the hint.
uint32_t repair_slot_policy(const Model* om, uint32_t shard_max) {
    const uint64_t want = std::max<uint64_t>(8192u, 2ull * (om ? om->repair_hint : 0u) + 4096u);
    return (uint32_t)std::min<uint64_t>(std::max<uint32_t>(shard_max, 1u), want);
}

// the frame is enqueued on its lane: results refer to it, model-changing calls on the owner's stream come after it (viewer_bind)
// (newest: the frame is the newest one enqueued — a frame that is redone at its retirement while a newer one is in flight on another lane
//  gets its lane's event again, but results keep referring to the newer frame)
gsx_status lane_mark(gsx_viewer* owner, ShardPending& p, bool newest) {
    p.lane->band_edges = owner->band_edges_forced;  // stage calls between frames see the caller's layout (or equal bands), not this frame's
    p.lane->held_w = p.lane->width;  // (every caller marks the frame while ITS uniforms are in force: frame_front, or between two swap_uniforms)
    p.lane->held_h = p.lane->height;
    if (p.lane != owner) {
        HIPCHK(gsx::op::EventRecord(p.lane->lane_event, p.lane->stream));
        p.lane->lane_busy = true;
    }
    if (newest) owner->latest = p.lane == owner ? nullptr : p.lane;
    return GSX_OK;
}

gsx_status frame_front(Ctx& c, bool eager_gather) {
    choose_edges(c.owner, *c.p, c.world, (c.l->height + GSX_TILE - 1) / GSX_TILE);
    gsx_status st = frame_buffers(c);
    if (st) return st;
    ShardPending& p = *c.p;
    const size_t n = p.order.size();
    const uint32_t n_tiles = ((c.l->width + GSX_TILE - 1) / GSX_TILE) * ((c.l->height + GSX_TILE - 1) / GSX_TILE);
    snapshot_uniforms(c.l, p);
    for (size_t i = 0; i < n; ++i) {  // every model's projection first: independent of everything that follows
        // the limits this lane's last frame computed become this frame's (kept apart until now: a redo of that frame wanted its own)
        if ((st = gsx_shard_frame_end(c.l, p.order[i].c_str()))) return st;
        // limits the caller set for this frame (gsx_shard_set_limits) live with the owner's model, whichever lane renders
        Model* om = find_model(c.owner, p.order[i].c_str());
        const uint32_t* override_limits = om && om->shard_override_tiles == n_tiles ? om->shard_limit_override.as<uint32_t>() : nullptr;
        if (om) om->shard_override_tiles = 0;
        if ((st = gsx_shard_frame_begin(c.l, p.order[i].c_str(), c.world, c.rank, p.speculate, override_limits))) return st;
    }
    p.slot.resize(n);
    p.pair_caps.assign(n, {});
    p.counted.assign(n, false);
    p.vseq.assign(n, 0u);
    p.repair_slot.assign(n, 0u);
    p.limited.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        if ((st = gsx_shard_slot_records(c.l, p.order[i].c_str(), c.world, p.shard_max[i], &p.slot[i]))) return st;
        const Model* lm = find_model(c.l, p.order[i].c_str());
        p.limited[i] = lm && lm->shard_frame_limited ? 1 : 0;
        plan_pair_slots(c, i);
        // Only a frame whose exchange was limited by windows can have refused a tile anything.  The models that have another model
        // behind them get the device-decided round (nothing may wait between two models); the LAST model's repair is decided by the
        // host when the frame is retired — exactly sized, and only in the frames that need it: with frames in flight that look is off
        // the critical path anyway, and a frame that repairs nothing pays no fall-through launches and no empty slots
        // (the device deciding for the last model too — no host look ever, at ~15 launches and `world` slots a frame — was the round-5 A/B's loser)
        // (a frame that is enqueued model by model — frame_step — reads every model's verdict before the next model goes out: host-decided
        //  repairs throughout, no fall-through launches)
        if (!p.stepped && p.limited[i] && i + 1 < n) p.repair_slot[i] = repair_slot_policy(find_model(c.owner, p.order[i].c_str()), p.shard_max[i]);
    }
    c.l->shard_frames_enqueued += 1;
    p.lane_frame = c.l->shard_frames_enqueued;
    if (p.stepped) return lane_mark(c.owner, p, true);  // (the models follow one by one: frame_step)
    for (size_t i = 0; i < n; ++i) {
        const char* key = p.order[i].c_str();
        if ((st = exchange_round(c, i, 0, p.slot[i], &p.pair_caps[i], false, true))) return st;
        if ((st = shard_verify_staged(c.l, key, c.world, c.l->shard_sat_all.p))) return st;
        // the repair round: enqueued without asking; its kernels fall through when the verification counted no tile in need
        if (p.repair_slot[i] && (st = exchange_round(c, i, 1, p.repair_slot[i], nullptr, true))) return st;
        // the model's next limits and their windows + the frame's verdict for this model: one launch
        if ((st = shard_next_windows_post(c.l, key, c.world, c.l->shard_sat_all.p, p.margin, p.radius, p.repair_slot[i] ? c.l->shard_sat_all.p : nullptr, &p.vseq[i])))
            return st;
    }
    p.next_model = (uint32_t)n;  // (everything is enqueued; the verdicts are read at retirement)
    // (a gather to ONE rank over RCCL hangs if the ranks name different roots: the first frame after gsx_shard_set_gather_root
    //  gathers only when its verdict has confirmed that they agree — the caller retires it at once)
    if (eager_gather && (st = band_gather(c))) return st;
    return lane_mark(c.owner, p, true);
}

// the context of a frame that was enqueued by an earlier call (the lane may be running a newer frame by now — with one frame in flight
// it is the owner itself: the frame's own geometry)
Ctx context_of(gsx_viewer* owner, ShardPending& p) {
    Ctx c{owner, p.lane, &p, owner->comm_world, owner->comm_rank, {}, 0, {}, 0, 0};
    c.tiles_x = (p.uniforms.width + GSX_TILE - 1) / GSX_TILE;
    c.tiles_y = (p.uniforms.height + GSX_TILE - 1) / GSX_TILE;
    c.bands.world = c.world;
    for (uint32_t g = 0; g <= c.world; ++g)
        c.bands.e[g] = p.edges.size() == (size_t)c.world + 1u ? p.edges[g] : g * ((c.tiles_y + c.world - 1) / c.world);
    return c;
}

// model i's verdict, read from the lane's ring (waits for it if it has not arrived): the hints for the frames to come, whether a slot
// overflowed, whether the model's tiles were refused records they need (*need)
gsx_status read_verdict(Ctx& c, size_t i, bool* overflow, bool* need) {
    ShardPending& p = *c.p;
    gsx_shard_stats& ss = c.owner->shard_stats;
    gsx_shard_verdict verdict{};
    const uint32_t* block = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    gsx_status st = shard_wait_ring(p.lane, p.vseq[i], &verdict, &block);
    ss.verdict_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    if (st) return st;
    if ((st = read_block(c, i, block))) return st;
    p.counted[i] = true;
    if (Model* om = find_model(c.owner, p.order[i].c_str())) {  // next frame's slots (global figures: the same on every rank)
        om->slot_hint = verdict.max_records;
        om->slot_hint_known = true;
        om->slot_hint_limited = p.limited[i] != 0;
        if (p.repair_slot[i]) {
            // (a third of cfg4's orbit frames repair, a few thousand records each, the others nothing: the hint is a maximum
            //  that decays by a sixteenth per frame — a slot sized by the last frame alone overflowed in 8 % of the frames)
            om->repair_hint = std::max(block[kVerdictRepairMax], om->repair_hint - om->repair_hint / 16u);
            ss.last_repair_records = block[kVerdictRepairMax];
        }
    }
    static const bool debug = getenv("GSX_SHARD_DEBUG") != nullptr;
    if (debug)
        fprintf(stderr, "[gsx shard] rank %u frame %llu model '%s': need %u, slot %u (busiest pair wanted %u, overflow %u), repair slot %u (busiest pair had %u, overflow %u)%s\n",
                c.rank, (unsigned long long)p.lane_frame, p.order[i].c_str(), verdict.need_tiles, p.slot[i], verdict.max_records, verdict.overflow, p.repair_slot[i],
                block[kVerdictRepairMax], block[kVerdictRepairOver], p.stepped ? " (model by model)" : "");
    if (verdict.need_tiles && p.repair_slot[i]) p.repaired = true;
    *overflow = verdict.overflow != 0 || block[kVerdictRepairOver] != 0;
    *need = verdict.need_tiles != 0 && p.limited[i] && !p.repair_slot[i];
    return GSX_OK;
}

// A layered frame with frames in flight goes out MODEL BY MODEL, its steps interleaved with another frame's (gsx_shard_render_frame_keys):
// before model i is enqueued the verdict of model i - 1 is read — it was enqueued a step ago, and meanwhile the host enqueued a model of
// the other frame, whose kernels keep the device busy — and its repair, where one is needed, is exchanged exactly sized.  No always-
// enqueued repair rounds: cfg5's three inner models paid ~22 fall-through launches each per frame, and the host, at ~3 us a launch,
// was what bounded two frames in flight.  false in *more: the frame has no model left (frame_retire reads the last model's verdict).
gsx_status frame_step(gsx_viewer* owner, ShardPending& p, bool* more) {
    const size_t n = p.order.size();
    *more = false;
    if (!p.stepped || p.next_model >= n || p.overflowed) return GSX_OK;
    Ctx c = context_of(owner, p);
    const bool own = p.lane == owner;
    if (own) swap_uniforms(p.lane, p);  // (the caller may have moved the camera on since the frame was begun)
    gsx_status st = frame_buffers(c);
    const size_t i = p.next_model;
    if (!st && i > 0 && p.read_models < i) {
        bool overflow = false, need = false;
        st = read_verdict(c, i - 1, &overflow, &need);
        p.read_models = (uint32_t)i;
        if (!st && overflow) p.overflowed = true;  // (the whole frame is redone at its retirement: nothing more of it goes out now)
        else if (!st && need) st = repair_sized(c, i - 1);
    }
    if (!st && !p.overflowed) {
        const char* key = p.order[i].c_str();
        st = exchange_round(c, i, 0, p.slot[i], &p.pair_caps[i], false, true);
        if (!st) st = shard_verify_staged(c.l, key, c.world, c.l->shard_sat_all.p);
        if (!st) st = shard_next_windows_post(c.l, key, c.world, c.l->shard_sat_all.p, p.margin, p.radius, nullptr, &p.vseq[i]);
        if (!st) p.next_model = (uint32_t)i + 1u;
    }
    if (!st) st = lane_mark(owner, p, !owner->shard_pending.empty() && &owner->shard_pending.back() == &p);
    if (own) swap_uniforms(p.lane, p);
    *more = !st && !p.overflowed && p.next_model < n;
    return st;
}

// The frame's verdicts, read (with frames in flight they arrived while the next frame was being enqueued); a frame whose slots
// overflowed is redone here, before its lane is used again and before anybody can read it.
gsx_status frame_retire(gsx_viewer* owner, ShardPending& p) {
    for (bool more = p.stepped; more;) {  // (a frame that goes out model by model: whatever has not gone out yet)
        const gsx_status sst = frame_step(owner, p, &more);
        if (sst) return sst;
    }
    Ctx c = context_of(owner, p);
    gsx_status st = GSX_OK;
    gsx_shard_stats& ss = owner->shard_stats;
    bool overflow = p.overflowed, host_repair = false;
    if (!p.settled && !p.overflowed) {
        for (size_t i = p.stepped ? p.read_models : 0u; i < p.order.size(); ++i) {
            bool over = false, need = false;
            if ((st = read_verdict(c, i, &over, &need))) return st;
            overflow = overflow || over;
            // (the last model — in a frame that went out model by model the only one not read yet: its repair waits for this look)
            if (need) host_repair = true;
        }
    }
    if (host_repair && !overflow) {
        // the last model's tiles were refused records they need: the exchange that brings them, sized exactly, behind what the band holds;
        // then its limits and the band gather once more (the early ones showed the frame without the repair)
        const bool own = p.lane == owner;
        if (own) swap_uniforms(p.lane, p);
        st = frame_buffers(c);
        if (!st) st = repair_sized(c, p.order.size() - 1);
        if (!st) st = band_gather(c);
        if (!st) st = lane_mark(owner, p, !owner->shard_pending.empty() && &owner->shard_pending.back() == &p);
        if (own) swap_uniforms(p.lane, p);
        if (st) return st;
    }
    if (overflow) {
        const bool own = p.lane == owner;
        if (own) swap_uniforms(p.lane, p);  // (a lane keeps the uniforms its frame was enqueued with until it is acquired again)
        st = frame_buffers(c);
        if (!st) st = redo_safe(c);
        if (!st && !p.gathered) st = band_gather(c);
        if (!st) st = lane_mark(owner, p, !owner->shard_pending.empty() && &owner->shard_pending.back() == &p);
        if (own) swap_uniforms(p.lane, p);
        if (st) return st;
    } else if (!p.gathered) {
        // (the frame's OWN viewport: gsx_update_camera may have resized the owner since the frame was enqueued, and a gather under the new
        //  size would use the wrong row bytes and band offsets — and frame_buffers would reallocate the framebuffer the frame lies in)
        const bool own = p.lane == owner;
        if (own) swap_uniforms(p.lane, p);
        st = frame_buffers(c);
        if (!st) st = band_gather(c);
        if (!st) st = lane_mark(owner, p, !owner->shard_pending.empty() && &owner->shard_pending.back() == &p);
        if (own) swap_uniforms(p.lane, p);
        if (st) return st;
    }
    ss.frames += 1;
    if (p.repaired) ss.repair_frames += 1;
    // this frame's next limits become the models' limits now — unless the lane has begun a newer frame, which took them over already
    if (p.lane->shard_frames_enqueued == p.lane_frame)
        for (const std::string& k : p.order)
            if ((st = gsx_shard_frame_end(p.lane, k.c_str()))) return st;
    owner->last_edges.assign(c.bands.e, c.bands.e + c.world + 1);
    return GSX_OK;
}

struct BusyGuard {
    gsx_viewer* v;
    explicit BusyGuard(gsx_viewer* v_) : v(v_) { v->shard_busy = true; }
    ~BusyGuard() { v->shard_busy = false; }
};

// whatever is in flight is retired, in order
gsx_status retire_all(gsx_viewer* v) {
    while (!v->shard_pending.empty()) {
        const gsx_status st = frame_retire(v, v->shard_pending.front());
        v->shard_pending.pop_front();
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    return GSX_OK;
}

}  // namespace

// viewer_bind (gsx_state.h): any other entry point first finishes the sharded frames in flight (on every rank alike: an
// SPMD host makes the same calls in the same order)
gsx_status gsx::shard_complete_pending(gsx_viewer* v) {
    BusyGuard guard(v);
    return retire_all(v);
}

extern "C" {

gsx_status gsx_shard_render_frame_keys(gsx_viewer* v, const char* const* keys_far_to_near, uint32_t n_keys, const uint32_t* shard_records_max,
                                       uint32_t speculate, float margin, uint32_t radius) {
    if (!v || !keys_far_to_near || !shard_records_max || n_keys == 0 || n_keys > 1024)
        return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame_keys: null argument or no keys");
    if (v->parent) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame_keys: called on a lane");
    HIPCHK(hipSetDevice(v->device));
    if (!has_comm(v)) return fail(GSX_ERR_RCCL, "gsx_shard_render_frame: no communicator (gsx_viewer_comm_init / _init_group / _init_custom)");
    for (uint32_t i = 0; i < n_keys; ++i) {
        if (!find_model(v, keys_far_to_near[i])) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_render_frame: no model '%s'", keys_far_to_near[i] ? keys_far_to_near[i] : "(null)");
        for (uint32_t j = 0; j < i; ++j)
            if (!strcmp(keys_far_to_near[i], keys_far_to_near[j])) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame_keys: model '%s' listed twice", keys_far_to_near[i]);
    }
    BusyGuard guard(v);
    gsx_status st = GSX_OK;
    uint32_t lanes = std::max(1u, std::min(v->options.frames_in_flight, 4u));
    if ((st = comm_ensure_lanes(v, lanes))) return st;
    if (lanes > 1 && !shard_frame_may_use_lanes(v, keys_far_to_near, n_keys)) {
        // a frame with a query runs on the viewer itself, alone
        if ((st = retire_all(v))) return st;
        lanes = 1;
    }
    if (lanes > 1) {
        // the models' edit records are shared by the lanes: when they have to be prepared again (a selection edit set by the host
        // since the last frame: nothing else has ordered it), the frames in flight — which read the old ones — complete first
        bool again = false;
        for (uint32_t i = 0; i < n_keys; ++i) again = again || edits_need_prepare(v, find_model(v, keys_far_to_near[i]));
        if (again && (st = retire_all(v))) return st;
        if ((st = prepare_edits_for_lanes(v, keys_far_to_near, n_keys))) return st;
    }
    while (v->shard_pending.size() > lanes - 1u) {  // (fewer lanes than the frames before: they complete first)
        st = frame_retire(v, v->shard_pending.front());
        v->shard_pending.pop_front();
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    gsx_viewer* lane = v;
    if (lanes > 1 && (st = lane_acquire(v, v->shard_turn++ % lanes, keys_far_to_near, n_keys, &lane))) return st;
    // verdict ring of the lane: a verdict per model and frame (a lane's frame is retired before the lane is used again)
    if (!lane->h_verdict_ring || lane->ring_slots < 2u * n_keys + 2u) {
        if ((st = retire_all(v))) return st;  // (a ring is only ever replaced while none of its verdicts is unread)
        if ((st = shard_ensure_ring(lane, 2u * n_keys))) return st;
    }
    v->shard_pending.emplace_back();
    ShardPending& p = v->shard_pending.back();
    p.lane = lane;
    for (uint32_t i = n_keys; i-- > 0;) {  // compositing order: nearest model first
        p.order.emplace_back(keys_far_to_near[i]);
        p.shard_max.push_back(shard_records_max[i]);
    }
    p.speculate = speculate;
    p.margin = margin;
    p.radius = radius;
    // layered models: model by model, with frames in flight interleaved with the frame before (frame_step)
    // (GSX_SHARD_LAYER_PIPELINE, read per frame — tests switch it: 0 = all models at once with device-decided repairs; 1 = model by model
    //  only with frames in flight; default 2 = always.  With ONE frame in flight nothing hides the host's look at a verdict, and model by
    //  model still wins: the look is a spin on a pinned word, ~20 us a model, against ~22 fall-through launches a model — cfg5 at world 1
    //  397 -> 475 fps, a world-8 rank alone 1.19 -> 1.13 ms, profiles/r05_ab_layer_pipeline.txt)
    const char* lp = getenv("GSX_SHARD_LAYER_PIPELINE");
    const int layer_pipeline = lp ? atoi(lp) : 2;
    p.stepped = layer_pipeline != 0 && (lanes > 1 || layer_pipeline == 2) && n_keys > 1 && speculate != 0;
    Ctx c{v, lane, &p, v->comm_world, v->comm_rank, {}, 0, {}, 0, 0};
    // The band gather is enqueued with the frame when the call is going to wait for this frame anyway (one frame in flight: the gather
    // runs while the verdict travels); with frames in flight it waits for the retirement — a third of cfg4's orbit frames repair their last
    // model there, and a gather that showed the frame without the repair would be bytes on the links for nothing.
    const bool eager_gather = lanes == 1 && (v->shard_gather_root < 0 || v->shard_root_confirmed);
    {
        // (GSX_GRAPH / gsx_debug_set_launch_graphs: the frame's launches between two collectives leave as cached, patched HIP graphs —
        //  host time only, csrc/gsx_launch.h; off by default)
        TraceScope trace(lane, TRACE_SHARD);
        st = frame_front(c, eager_gather);
        if (!st) st = trace.finish();
    }
    if (st) {
        v->shard_pending.clear();
        return st;
    }
    if (p.stepped) {
        // The new frame's models and what is left of the frames before it, one model each in turn, oldest frame first: the verdict a step
        // waits for belongs to a model that went out a turn ago, and the other frame's model keeps the device busy meanwhile.  The new frame
        // is left with (at least) half its models out: the next call finds as much left of it as it enqueues of the frame after.
        const uint32_t stop_at = (n_keys + 1u) / 2u;
        for (bool any = true; any;) {
            any = false;
            for (size_t k = 0; k + 1 < v->shard_pending.size(); ++k) {
                ShardPending& old = v->shard_pending[k];
                bool more = false;
                if ((st = frame_step(v, old, &more))) break;
                any = any || more;
            }
            if (st) break;
            ShardPending& mine = v->shard_pending.back();
            if (any || mine.next_model < stop_at) {
                bool more = false;
                if ((st = frame_step(v, mine, &more))) break;
                any = any || (more && mine.next_model < stop_at);
            }
        }
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    // (a gather to ONE rank whose root is not confirmed yet: the frame's verdict says whether the ranks agree; then the gather)
    if (v->shard_gather_root >= 0 && !v->shard_root_confirmed && (st = retire_all(v))) return st;
    // the frames in flight: every frame but the newest L - 1 is retired — with L >= 2 its verdict arrived while this frame was being
    // enqueued (and the device has this one queued); with one frame in flight the call waits for its own frame
    while (v->shard_pending.size() > lanes - 1u) {
        st = frame_retire(v, v->shard_pending.front());
        v->shard_pending.pop_front();
        if (st) {
            v->shard_pending.clear();
            return st;
        }
    }
    return GSX_OK;
}

// One model: what a host without Python calls once per frame after gsx_update_camera / gsx_update_model_transform.  Afterwards
// (gsx_sync) gsx_download_framebuffer returns the whole frame on every rank.
gsx_status gsx_shard_render_frame(gsx_viewer* v, const char* key, uint32_t shard_records_max, uint32_t speculate, float margin, uint32_t radius) {
    if (!key) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: null argument");
    const char* keys[1] = {key};
    return gsx_shard_render_frame_keys(v, keys, 1, &shard_records_max, speculate, margin, radius);
}

}  // extern "C"
