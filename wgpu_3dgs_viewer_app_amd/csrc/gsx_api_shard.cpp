// gsx_api_shard.cpp — C ABI for multi-GPU rendering: band layout, external framebuffer, screen bands, and the stage split of
// the index-sharded exchange (pack / import / feedback / second round).  No reference counterpart (src/main.rs:85-98).
#include <chrono>
#include <sched.h>
#include <thread>

#include "gsx_state.h"

using namespace gsx;

extern "C" {

// ---- multi-GPU stage split ----------------------------------------------------------------------

gsx_status gsx_shard_layout(gsx_viewer* v, uint32_t world, uint32_t rank, gsx_shard_layout_t* out) {
    if (!v || !out || world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_layout: bad argument");
    const uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    if (gsx_status cst = check_bands(v, world, "gsx_shard_layout")) return cst;
    const BandEdges b = bands_of(v, world);
    const uint64_t row_bytes = (uint64_t)GSX_TILE * v->width * sizeof(float4);
    out->rows_per_rank = band_rows_max(b);  // (equal bands: what it always was)
    out->row_lo = std::min(b.e[rank], tiles_y);
    out->row_hi = std::min(b.e[rank + 1], tiles_y);
    out->band_bytes = (uint64_t)(b.e[rank + 1] - b.e[rank]) * row_bytes;
    out->band_offset_bytes = (uint64_t)b.e[rank] * row_bytes;
    out->padded_framebuffer_bytes = (uint64_t)std::max(b.e[world], tiles_y) * row_bytes;
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


gsx_status gsx_resolve_rgba8_device(gsx_viewer* v, const float bg[3], uint32_t y0, uint32_t y1, void* d_rgba) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!bg || !d_rgba || y1 < y0) return fail(GSX_ERR_INVALID_ARG, "gsx_resolve_rgba8_device: null argument or y1 < y0");
    // the newest frame may be a lane's: ITS framebuffer, resolved on the viewer's own stream — viewer_bind has ordered that
    // stream after the lane's frame, and the lane's next frame waits for what is enqueued here (epoch)
    gsx_viewer* src = result_lane(v);
    if ((st = ensure_fb(src))) return st;
    const uint64_t rows_avail = src->ext_fb ? src->ext_fb_bytes / (sizeof(float4) * (uint64_t)src->width) : src->height;
    if (y1 > rows_avail) return fail(GSX_ERR_INVALID_ARG, "gsx_resolve_rgba8_device: rows [%u, %u) of a %llu-row framebuffer", y0, y1, (unsigned long long)rows_avail);
    const uint64_t npx = (uint64_t)(y1 - y0) * src->width;
    HIPCHK(launch_resolve_rgba8(v->stream, fb_ptr(src) + (size_t)y0 * src->width, (uint32_t)npx, bg[0], bg[1], bg[2], static_cast<uint32_t*>(d_rgba)));
    return GSX_OK;
}

gsx_status gsx_shard_pack(gsx_viewer* v, const char* key, uint32_t world, const uint32_t* d_tile_window, void* d_send,
                          uint64_t capacity_records, uint64_t* counts) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_pack: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: model '%s' has no projection this frame (gsx_preprocess first)", key);
    if (world == 0 || world > 64 || !counts) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack: world must be 1..64");
    if ((st = check_bands(v, world, "gsx_shard_pack"))) return st;
    const uint32_t n = (uint32_t)m->n;
    const bool from_list = !d_tile_window && m->shard_win_set && m->cand_valid;
    const uint32_t rounds = pack_rounds(from_list, n), tile = 256u * rounds;
    const uint32_t nb = (uint32_t)pack_blocks(n, rounds);
    const BandEdges bands = bands_of(v, world);
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
        HIPCHK(gsx::op::MemcpyAsync(m->pack_window.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
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
    HIPCHK(gsx::op::MemsetAsync(totals, 0, 4 * 64, v->stream));
    HIPCHK(launch_pack_count(v->stream, m->proj_rec(), n, bands, window, tiles_x, masks, table, list, d_list_n, travellers, trav_counts));
    if (nb) HIPCHK(launch_rowscan(v->stream, table, world, nb, totals, d_list_n, tile));
    if (travellers && nb) {
        // shade the travellers the first round did not: compact their indices, k_shade skips what is shaded already
        HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(n, 1)));
        HIPCHK(launch_rowscan(v->stream, trav_counts, 1, nb, &dc->n_sorted2));
        HIPCHK(launch_admit_scatter(v->stream, m->proj_rec().key, n, travellers, trav_counts, m->adm_pairs.as<uint2>(), nullptr, nullptr, rounds));
        if ((st = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted2, m->adm_ballots.as<unsigned long long>(), m->rect8_active}))) return st;
        m->cand_valid = false;  // adm_pairs now holds the repair travellers
    }
    uint32_t h_tot[64];
    HIPCHK(gsx::op::MemcpyAsync(h_tot, totals, 4 * 64, hipMemcpyDeviceToHost, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
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
    m->shard_win_current = false;  // (no longer the windows of the model's own limits)
    HIPCHK(gsx::op::MemcpyAsync(m->shard_win.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
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
    if ((st = check_bands(v, world, "gsx_shard_import"))) return st;
    if ((st = ensure_import_capacity(m, n_records))) return st;
    HIPCHK(launch_import_records(v->stream, d_recv, (uint32_t)n_records, m->imp_rec()));
    // every imported record is visible by construction
    HIPCHK(gsx::op::MemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&m->counters.as<Counters>()->n_visible), (int)(uint32_t)n_records, 2,
                             v->stream));  // n_visible and n_sorted
    m->stats_pending = true;
    m->rec_n = n_records;
    m->use_imported = true;
    {
        const BandEdges b = bands_of(v, world);
        m->row_lo = b.e[rank];
        m->row_hi = b.e[rank + 1];
        m->rows_nominal = (((v->height + GSX_TILE - 1) / GSX_TILE) + world - 1u) / world;
    }
    m->has_window = d_tile_window != nullptr;
    m->window_ptr = nullptr;
    m->import_min_ends = nullptr;
    if (d_tile_window) {
        HIPCHK(m->window.ensure(window_bytes(v)));
        HIPCHK(gsx::op::MemcpyAsync(m->window.p, d_tile_window, window_bytes(v), hipMemcpyDeviceToDevice, v->stream));
    }
    m->sorted = m->counters_valid = m->binned = false;
    return GSX_OK;
}

gsx_status gsx_shard_feedback_words(gsx_viewer* v, uint32_t world, uint32_t* out_words) {
    if (!v || !out_words || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback_words: bad argument");
    // the common stride of the gathered pieces (gsx_internal.h, feedback_*): statistics + the tallest band's keys + its rows' work
    *out_words = feedback_stride(bands_of(v, world), (v->width + GSX_TILE - 1) / GSX_TILE);
    return GSX_OK;
}

gsx_status gsx_shard_feedback(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, void* d_out_u32) {
    return shard_feedback(v, key, world, rank, d_out_u32, false);
}

}  // extern "C"

// zero_verify_state: the kernel also zeroes what the verification behind the gather accumulates into (its counter, its ticket, the need
// bitmap): shard_verify_staged then launches nothing in front of its kernel
gsx_status gsx::shard_feedback(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, void* d_out_u32, bool zero_verify_state) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_out_u32) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_feedback: no model '%s'", key ? key : "(null)");
    if (!m->binned) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback: model '%s' not rendered this frame", key);
    if (!v->options.progressive) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback needs gsx_render_options.progressive = 1");
    if (world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_feedback: bad world/rank %u/%u", world, rank);
    if ((st = check_bands(v, world, "gsx_shard_feedback"))) return st;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    const uint32_t row_words = (tiles_x + 31) / 32;
    const uint32_t* tile_sat = v->done_bits.as<uint32_t>() + 1 + (size_t)row_words * tiles_y;
    const uint32_t* row_work = tile_sat + (size_t)tiles_x * tiles_y;  // (do_render: behind the saturation keys)
    const gsx_viewer* o = v->parent ? v->parent : v;
    // how this rank sizes slots and bands travels with the feedback: ranks that disagree would exchange slots of different sizes
    // (over RCCL: a hang or silent truncation) — the verification posts the disagreement and the frame fails first
    const uint32_t policy = (o->shard_balance ? kPolicyBalance : 0u) | (o->shard_pair_slots ? kPolicyPairSlots : 0u) | (m->slot_force << 8);
    uint32_t *za = nullptr, *zb = nullptr;
    uint32_t nza = 0, nzb = 0;
    if (zero_verify_state) {
        HIPCHK(m->shard_need_bits.ensure(4 * (size_t)row_words * tiles_y));
        za = &m->counters.as<Counters>()->shard_need;  // shard_need + shard_ticket
        nza = 2;
        zb = m->shard_need_bits.as<uint32_t>();
        nzb = row_words * tiles_y;
    }
    m->verify_state_zeroed = zero_verify_state;
    HIPCHK(launch_shard_feedback(v->stream, tile_sat, row_work, tiles_x, tiles_y, bands_of(v, world), rank, static_cast<uint32_t*>(d_out_u32),
                                 m->counters.as<Counters>(), m->shard_behind ? m->spec_done_before.as<uint32_t>() : nullptr, row_words,
                                 (uint32_t)(o->shard_gather_root + 1), policy, za, nza, zb, nzb));
    return GSX_OK;
}

extern "C" {

gsx_status gsx_render_more(gsx_viewer* v, const char* const* keys, uint32_t n_keys) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    return do_render(v, keys, n_keys, true);
}

}  // extern "C"

// =====================================================================================================================
// Device-resident exchange protocol: the index-sharded frame with windows, verification, repair windows, next frame's limits
// and every record count ON THE DEVICE, fixed-size exchange slots whose headers carry the counts, and ONE thing the host
// waits for per frame: the verdict of round 0 (two words in pinned memory, written by the verification kernel from globally
// gathered data — every rank reads the same verdict and takes the same decision):
//
//   gsx_shard_frame_begin(key, world, rank, speculate)        windows [0, limit) from last frame's limits -> projection
//   gsx_shard_slot_records(key, world, max_shard, &T)         round-0 slot size: 2x what the LAST frame's busiest (rank,
//                                                             destination) pair wanted (a global figure from its verdict)
//   gsx_shard_pack_slots(key, world, 0, d_send, T)            world slots of (1 + T) records, header = counts
//   all-to-all of the slots                                   gsx_comm_all_to_all
//   gsx_shard_import_slots(key, d_recv, world, rank, 0, T)    import + depth sort + render this rank's band
//   gsx_shard_feedback(..) + all-gather                       saturation depth key of every tile + the ranks' slot statistics
//   gsx_shard_verify(key, world, d_sat_all, &seq)             repair windows on the device; posts the verdict
//   gsx_shard_next_windows(..); all-gather of the bands       enqueued BEFORE the wait: what follows when nothing is wrong
//   gsx_shard_wait_verdict(seq, &verdict)                     {tiles needing repair, a slot overflowed somewhere, busiest pair}
//     verdict.overflow: round 0 again with T = max_shard (a destination can be sent at most a whole shard: always fits)
//     verdict.need > 0: the repair round, sized EXACTLY — gsx_shard_repair_count + all-gather + gsx_shard_post_counts +
//                       wait -> T1; pack_slots(.., 1, T1) -> all-to-all -> import_slots(.., 1, T1) -> feedback + all-gather,
//                       then next_windows and the band all-gather once more (they replace the early ones)
// Common case (no repair, no overflow): one short wait that overlaps the band all-gather; nothing else ever blocks.
// Whatever happens the frame that leaves the GPU is complete, and equal to the single-GPU frame bit for bit.
// =====================================================================================================================
namespace {

// The two halves of a round's pack.  pack_count: which records travel where (destination masks, per-workgroup tables,
// per-destination totals; with explicit windows on a lazily projected shard also the travellers' ballots).  window ==
// nullptr: the windows announced to the projection (its candidate list), or everything.  gate (nullable): bitmap of tiles;
// a record whose rectangle holds none of them is refused before any window is looked at (the repair round: few tiles).
// d_skip (nullable): a device word; 0 = nothing travels in this round (an always-enqueued repair round that has nothing to repair)
gsx_status pack_count(gsx_viewer* v, Model* m, uint32_t world, const uint2* explicit_window, const uint32_t* gate, const WindowPyramid* pyramid,
                      const uint32_t* d_skip = nullptr) {
    const uint32_t n = (uint32_t)m->n;
    m->pack_list = !explicit_window && m->shard_win_set && m->cand_valid;
    m->pack_rounds = pack_rounds(m->pack_list, n);
    const uint32_t nb = (uint32_t)pack_blocks(n, m->pack_rounds);
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->pack_table.ensure(4 * ((size_t)64 * std::max(nb, 1u) + 64)));
    HIPCHK(m->pack_masks.ensure(8 * (size_t)std::max(n, 1u)));
    m->pack_travellers = false;
    const uint2* window = nullptr;
    const uint2* list = nullptr;
    const uint32_t* d_list_n = nullptr;
    unsigned long long* travellers = nullptr;
    uint32_t* trav_counts = nullptr;
    Counters* dc = m->counters.as<Counters>();
    gsx_status st;
    if (explicit_window) {
        window = explicit_window;
        if (m->lazy) {  // explicit windows on a lazily projected shard (the repair exchange): travellers may be unshaded
            HIPCHK(m->trav_ballots.ensure(8 * ((std::max<size_t>(n, 1) + 63) / 64)));
            HIPCHK(m->trav_counts.ensure(4 * std::max<size_t>(nb, 1)));
            travellers = m->trav_ballots.as<unsigned long long>();
            trav_counts = m->trav_counts.as<uint32_t>();
            m->pack_travellers = true;
        }
    } else if (m->pack_list) {
        window = m->shard_win.as<uint2>();
        list = m->adm_pairs.as<uint2>();
        d_list_n = &dc->n_candidates;
    } else if (m->lazy) {
        if ((st = complete_records(v, m))) return st;
    }
    uint32_t* table = m->pack_table.as<uint32_t>();
    uint32_t* totals = table + (size_t)64 * std::max(nb, 1u);
    // (the row scan writes every total — 0 when the round is skipped —: no zeroing launch in front)
    HIPCHK(launch_pack_count(v->stream, m->proj_rec(), n, bands_of(v, world), window, tiles_x, m->pack_masks.as<unsigned long long>(), table, list, d_list_n,
                             travellers, trav_counts, gate, (tiles_x + 31) / 32, window ? pyramid : nullptr, d_skip));
    if (nb) HIPCHK(launch_rowscan(v->stream, table, world, nb, totals, d_list_n, 256u * m->pack_rounds, d_skip));
    else HIPCHK(launch_zero_words(v->stream, totals, 64, nullptr, 0));
    return GSX_OK;
}

// pack_write: shade the travellers the lazy projection skipped, slot headers, the records into their slots
gsx_status pack_write(gsx_viewer* v, Model* m, uint32_t world, void* d_send, const SlotSpans& slots, uint32_t round, const uint32_t* d_skip = nullptr) {
    const uint32_t n = (uint32_t)m->n;
    const uint32_t nb = (uint32_t)pack_blocks(n, m->pack_rounds);
    Counters* dc = m->counters.as<Counters>();
    uint32_t* table = m->pack_table.as<uint32_t>();
    uint32_t* totals = table + (size_t)64 * std::max(nb, 1u);
    if (m->pack_travellers && nb) {
        HIPCHK(m->adm_pairs.ensure(8 * std::max<size_t>(n, 1)));
        // (the scatter sums the raw per-workgroup counts itself and leaves the total: no scan launch in between)
        HIPCHK(launch_admit_scatter(v->stream, m->proj_rec().key, n, m->trav_ballots.as<unsigned long long>(), m->trav_counts.as<uint32_t>(),
                                    m->adm_pairs.as<uint2>(), d_skip, &dc->n_sorted2, m->pack_rounds));
        gsx_status sst = shade_admitted(v, m, LateProjection{m->adm_pairs.as<uint2>(), &dc->n_sorted2, m->adm_ballots.as<unsigned long long>(), m->rect8_active});
        if (sst) return sst;
        m->cand_valid = false;
        m->pack_travellers = false;
    }
    HIPCHK(launch_pack_headers(v->stream, totals, world, slots, d_send, dc, round, d_skip));
    uint64_t send_records = 0;
    for (uint32_t p = 0; p < world; ++p) send_records = std::max<uint64_t>(send_records, (uint64_t)slots.off[p] + 1u + slots.cap[p]);
    HIPCHK(launch_pack_scatter(v->stream, m->proj_rec(), n, world, m->pack_masks.as<unsigned long long>(), table, totals, d_send,
                               send_records, m->pack_list ? m->adm_pairs.as<uint2>() : nullptr,
                               m->pack_list ? &dc->n_candidates : nullptr, &slots, d_skip));
    m->stats_pending = true;
    return GSX_OK;
}

// the repair round's count: windows [limit, inf) on the tiles that need it, nothing elsewhere; decided by the min-pyramid of
// the window starts (KEY_ALL where no tile needs anything), built here — only frames that repair pay for it
gsx_status repair_pack_count(gsx_viewer* v, Model* m, uint32_t world, const uint32_t* d_skip = nullptr) {
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->shard_pyr2.ensure(4 * window_pyramid_words(tiles_x, tiles_y)));
    HIPCHK(launch_window_pyramid(v->stream, m->shard_win2.as<uint2>(), tiles_x, tiles_y, m->shard_pyr2.as<uint32_t>(), true, d_skip));
    WindowPyramid pyr2 = window_pyramid_layout(tiles_x, tiles_y, m->shard_pyr2.as<uint32_t>());
    pyr2.min_of_starts = 1;
    return pack_count(v, m, world, m->shard_win2.as<uint2>(), nullptr, &pyr2, d_skip);
}

gsx_status ensure_verdict(gsx_viewer* v) {
    if (!v->h_shard_verdict) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&v->h_shard_verdict), 4 * (size_t)kVerdictWords, hipHostMallocDefault));
        memset(v->h_shard_verdict, 0, 4 * (size_t)kVerdictWords);
    }
    return GSX_OK;
}

}  // namespace

extern "C" {

gsx_status gsx_shard_frame_begin(gsx_viewer* v, const char* key, uint32_t world, uint32_t rank, uint32_t speculate, const uint32_t* d_limit_override) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_frame_begin: no model '%s'", key ? key : "(null)");
    if (world == 0 || world > 64 || rank >= world) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_frame_begin: bad world/rank %u/%u", world, rank);
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE, n_tiles = tiles_x * tiles_y;
    if (d_limit_override) {  // tests / a caller with its own policy: these limits instead of the ones the last frame left
        HIPCHK(m->shard_limit.ensure(4 * (size_t)n_tiles));
        HIPCHK(gsx::op::MemcpyAsync(m->shard_limit.p, d_limit_override, 4 * (size_t)n_tiles, hipMemcpyDeviceToDevice, v->stream));
        m->shard_limit_valid = true;
        m->shard_limit_tx = tiles_x;
        m->shard_limit_ty = tiles_y;
        m->slot_hint = 0;  // nothing is known about what THESE limits let through: the safe slot size
        m->slot_hint_known = false;
        if (v->parent)
            if (Model* om = find_model(v->parent, key)) om->slot_hint_known = false;
    }
    m->shard_frame_limited = speculate && m->shard_limit_valid && m->shard_limit_tx == tiles_x && m->shard_limit_ty == tiles_y;
    m->shard_win_set = false;
    if (d_limit_override) m->shard_win_current = false;
    if (m->shard_frame_limited) {
        HIPCHK(m->shard_win.ensure(window_bytes(v)));
        // (the kernel that computed these limits wrote their windows too — shard_next_windows_post — unless they came from outside)
        if (!m->shard_win_current || m->shard_win_tiles != n_tiles)
            HIPCHK(launch_limits_to_windows(v->stream, m->shard_limit.as<uint32_t>(), n_tiles, m->shard_win.as<uint2>()));
        m->shard_win_current = true;
        m->shard_win_tiles = n_tiles;
        // [max-pyramid of the window ends: admission in the projection kernel | min-pyramid: "every tile takes it" in the binning]
        const size_t pw = window_pyramid_words(tiles_x, tiles_y);
        HIPCHK(m->shard_pyr.ensure(8 * pw));
        HIPCHK(launch_window_pyramid(v->stream, m->shard_win.as<uint2>(), tiles_x, tiles_y, m->shard_pyr.as<uint32_t>(), false, nullptr,
                                     m->shard_pyr.as<uint32_t>() + pw));
        m->shard_tiles_x = tiles_x;
        m->shard_tiles_y = tiles_y;
        m->shard_win_set = true;
    }
    m->repair_counted = false;
    return do_preprocess(v, m);
}

gsx_status gsx_shard_slot_records(gsx_viewer* v, const char* key, uint32_t world, uint32_t shard_records_max, uint32_t* out_records) {
    Model* m = find_model(v, key);
    if (!m || !out_records) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_slot_records: no model '%s' / bad argument", key ? key : "(null)");
    if (shard_records_max < m->n) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_slot_records: shard_records_max %u < this shard's %llu records",
                                              shard_records_max, (unsigned long long)m->n);
    // EVERY rank must arrive at the same number (the all-to-all moves equal slots), so nothing rank-local enters it:
    // shard_records_max is the largest shard of the model (the caller's partition), slot_hint is the global figure of the last
    // verdict.  Without windows every visible record travels: a destination can be sent at most a whole shard, which
    // always fits.  With windows: twice what the busiest pair wanted last frame; should that still be too small, the
    // verdict says so and round 0 is redone with the safe size.
    const uint32_t n = std::max<uint32_t>(shard_records_max, 1u);
    uint32_t t = n;
    // (the figures of the last verdict live with the OWNER's model: whichever lane rendered that frame, whichever renders this one)
    const Model* hm = m;
    if (v->parent)
        if (const Model* om = find_model(v->parent, key)) hm = om;
    if (m->shard_frame_limited && hm->slot_hint_known && hm->slot_hint_limited) t = std::min<uint32_t>(n, std::max<uint32_t>(2u * hm->slot_hint + 4096u, 8192u));
    if (hm->slot_force) t = std::min<uint32_t>(n, hm->slot_force);  // gsx_shard_set_slot_records (the same on every rank, by contract)
    *out_records = t;
    (void)world;
    return GSX_OK;
}

gsx_status gsx_shard_pack_slots(gsx_viewer* v, const char* key, uint32_t world, uint32_t round, void* d_send, uint32_t slot_records) {
    if (world == 0 || world > 64 || slot_records == 0) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack_slots: bad argument");
    return shard_pack_slots(v, key, world, round, d_send, uniform_slots(world, slot_records));
}

}  // extern "C"

// slots: where the records for every destination go (sizes that both ends of every pair agree on: gsx_shard_frame.cpp)
gsx_status gsx::shard_pack_slots(gsx_viewer* v, const char* key, uint32_t world, uint32_t round, void* d_send, const SlotSpans& slots, bool gated) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_pack_slots: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack_slots: model '%s' has no projection this frame (gsx_shard_frame_begin first)", key);
    if (world == 0 || world > 64 || round > 1 || !d_send) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack_slots: bad argument");
    if ((st = check_bands(v, world, "gsx_shard_pack_slots"))) return st;
    if (round == 1) {
        if (m->shard_win2.bytes < window_bytes(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_pack_slots: round 1 before gsx_shard_verify");
        // the repair round is counted once (gsx_shard_repair_count sized it); packing without it counts here
        // gated: the round was enqueued without asking whether any tile needs it (gsx_shard_frame.cpp); its kernels look at the
        // verification's count on the device and fall through when it is 0
        const uint32_t* d_skip = gated ? &m->counters.as<Counters>()->shard_need : nullptr;
        if (!m->repair_counted && (st = repair_pack_count(v, m, world, d_skip))) return st;
        m->repair_counted = false;
        return pack_write(v, m, world, d_send, slots, round, d_skip);
    } else {
        // the senders decide by the max-pyramid of the window ends, per destination band (a conservative superset; the
        // receiver bins by the exact windows)
        WindowPyramid pyr = window_pyramid_layout(m->shard_tiles_x, m->shard_tiles_y, m->shard_pyr.as<uint32_t>());
        if ((st = pack_count(v, m, world, nullptr, nullptr, m->shard_win_set ? &pyr : nullptr))) return st;
    }
    return pack_write(v, m, world, d_send, slots, round);
}

extern "C" {

gsx_status gsx_shard_repair_count(gsx_viewer* v, const char* key, uint32_t world, void* d_out4) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_out4) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_repair_count: no model '%s' / null output", key ? key : "(null)");
    if (!m->preprocessed || m->shard_win2.bytes < window_bytes(v)) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_repair_count: before gsx_shard_verify");
    if ((st = repair_pack_count(v, m, world))) return st;
    m->repair_counted = true;
    const uint32_t* totals = m->pack_table.as<uint32_t>() + (size_t)64 * std::max<uint32_t>((uint32_t)pack_blocks(m->n, m->pack_rounds), 1u);
    HIPCHK(launch_shard_max_count(v->stream, totals, world, static_cast<uint32_t*>(d_out4)));
    return GSX_OK;
}

gsx_status gsx_shard_post_counts(gsx_viewer* v, uint32_t world, const void* d_counts_all, uint32_t* out_seq) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!d_counts_all || !out_seq || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_post_counts: bad argument");
    if ((st = ensure_verdict(v))) return st;
    *out_seq = ++v->shard_seq;
    HIPCHK(launch_shard_post_counts(v->stream, static_cast<const uint32_t*>(d_counts_all), world, v->h_shard_verdict, *out_seq));
    return GSX_OK;
}

gsx_status gsx_shard_import_slots(gsx_viewer* v, const char* key, const void* d_recv, uint32_t world, uint32_t rank, uint32_t round_flags,
                                  uint32_t slot_records) {
    if (world == 0 || world > 64 || slot_records == 0) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import_slots: bad argument");
    return shard_import_slots(v, key, d_recv, world, rank, round_flags, uniform_slots(world, slot_records));
}

}  // extern "C"

// slots: where every source's records lie in d_recv
gsx_status gsx::shard_import_slots(gsx_viewer* v, const char* key, const void* d_recv, uint32_t world, uint32_t rank, uint32_t round_flags,
                                   const SlotSpans& slots) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_import_slots: no model '%s'", key ? key : "(null)");
    if (!m->preprocessed) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import_slots: model '%s' has no frame constants (gsx_shard_frame_begin first)", key);
    const uint32_t round = round_flags & 1u;
    const bool behind = (round_flags & GSX_SHARD_BEHIND) != 0;  // a layered frame: nearer models are in the framebuffer already
    if (world == 0 || world > 64 || rank >= world || round_flags > 3u || !d_recv)
        return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import_slots: bad argument");
    if ((st = check_bands(v, world, "gsx_shard_import_slots"))) return st;
    uint64_t cap = 0;
    for (uint32_t p = 0; p < world; ++p) cap += slots.cap[p];
    if (cap >= 0xFFFFFFF0ull) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import_slots: too many records");
    if ((st = ensure_import_capacity(m, std::max<uint64_t>(cap, 1)))) return st;
    HIPCHK(launch_import_slots(v->stream, d_recv, world, slots, m->imp_rec(), m->counters.as<Counters>()));
    m->stats_pending = true;
    m->rec_n = cap;  // an upper bound: the count is on the device (Counters::n_sorted)
    m->use_imported = true;
    {
        const BandEdges b = bands_of(v, world);
        m->row_lo = b.e[rank];
        m->row_hi = b.e[rank + 1];
        m->rows_nominal = (((v->height + GSX_TILE - 1) / GSX_TILE) + world - 1u) / world;
    }
    // the receiving side of the pack predicate: a tile bins exactly the records its window admits
    const uint2* win = round == 0 ? (m->shard_frame_limited ? m->shard_win.as<uint2>() : nullptr) : m->shard_win2.as<uint2>();
    m->has_window = win != nullptr;
    m->import_min_ends = nullptr;
    m->window_ptr = win;  // the model's own maps (shard_win / shard_win2) stay put until the next frame: no copy
    if (win) {
        if (round == 0) m->import_min_ends = m->shard_pyr.as<uint32_t>() + window_pyramid_words((v->width + GSX_TILE - 1) / GSX_TILE, (v->height + GSX_TILE - 1) / GSX_TILE);
    }
    m->sorted = m->counters_valid = m->binned = false;
    if ((st = do_sort(v, m))) return st;
    if (round == 0) {
        m->shard_behind = behind;
        if (behind) {  // which tiles the nearer models had saturated: this model's feedback must not take their depths for its own
            const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
            const size_t bm = 4 * (size_t)((tiles_x + 31) / 32) * tiles_y;
            if (v->done_bits.bytes < 4 + bm) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_import_slots: GSX_SHARD_BEHIND without a nearer model in this frame");
            HIPCHK(m->spec_done_before.ensure(bm));
            HIPCHK(gsx::op::MemcpyAsync(m->spec_done_before.p, v->done_bits.as<uint32_t>() + 1, bm, hipMemcpyDeviceToDevice, v->stream));
        }
    }
    const char* keys[1] = {m->key.c_str()};
    return do_render(v, keys, 1, round == 1 || behind);
}

extern "C" {

gsx_status gsx_shard_verify(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, uint32_t* out_seq) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_sat_all || !out_seq) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_verify: no model '%s' / null argument", key ? key : "(null)");
    if (world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_verify: world must be 1..64");
    if ((st = check_bands(v, world, "gsx_shard_verify"))) return st;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->shard_win2.ensure(window_bytes(v)));
    HIPCHK(m->shard_need_bits.ensure(4 * (size_t)((tiles_x + 31) / 32) * tiles_y));
    if ((st = ensure_verdict(v))) return st;
    Counters* dc = m->counters.as<Counters>();
    HIPCHK(launch_zero_words(v->stream, &dc->shard_need, 2, m->shard_need_bits.as<uint32_t>(), ((tiles_x + 31) / 32) * tiles_y));  // shard_need + shard_ticket | bitmap
    *out_seq = ++v->shard_seq;
    const gsx_viewer* o = v->parent ? v->parent : v;
    HIPCHK(launch_shard_verify(v->stream, m->shard_frame_limited ? m->shard_limit.as<uint32_t>() : nullptr, static_cast<const uint32_t*>(d_sat_all),
                               tiles_x, tiles_y, bands_of(v, world), m->shard_win2.as<uint2>(), &dc->shard_need, &dc->shard_ticket,
                               v->h_shard_verdict, *out_seq, m->shard_need_bits.as<uint32_t>(), o->shard_balance ? 1u : 0u));
    m->repair_counted = false;
    m->stats_pending = true;
    return GSX_OK;
}

}  // extern "C"

// ---- verdicts that are read late: the device-decided frame (gsx_shard_frame.cpp) ----
// verdicts_outstanding: how many verdicts may be posted before the oldest is read (models per frame x frames that stay unretired)
gsx_status gsx::shard_ensure_ring(gsx_viewer* v, uint32_t verdicts_outstanding) {
    const uint32_t want = std::max<uint32_t>(8u, verdicts_outstanding + 2u);
    if (v->h_verdict_ring && v->ring_slots >= want) return GSX_OK;
    HIPCHK(gsx::op::StreamSynchronize(v->stream));  // (nothing in flight posts into the old ring any more)
    uint32_t* ring = nullptr;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&ring), 4 * (size_t)kVerdictWords * want, hipHostMallocDefault));
    memset(ring, 0, 4 * (size_t)kVerdictWords * want);
    // verdicts already posted and not read yet keep their slots' contents only if the slot index survives: it does not when the ring
    // grows, so the caller grows it only between frames whose verdicts have all been read (frame_front: before anything is enqueued)
    if (v->h_verdict_ring) (void)hipHostFree(v->h_verdict_ring);
    v->h_verdict_ring = ring;
    v->ring_slots = want;
    HIPCHK(v->verdict_stage.ensure(4 * (size_t)kVerdictWords));
    return GSX_OK;
}

gsx_status gsx::shard_verify_staged(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_sat_all) return fail(GSX_ERR_NOT_FOUND, "shard_verify_staged: no model '%s' / null argument", key ? key : "(null)");
    if ((st = check_bands(v, world, "gsx_shard_render_frame"))) return st;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->shard_win2.ensure(window_bytes(v)));
    HIPCHK(m->shard_need_bits.ensure(4 * (size_t)((tiles_x + 31) / 32) * tiles_y));
    HIPCHK(v->verdict_stage.ensure(4 * (size_t)kVerdictWords));
    Counters* dc = m->counters.as<Counters>();
    if (!m->verify_state_zeroed)  // (the feedback kernel in front of the gather zeroed them on its way: shard_feedback)
        HIPCHK(launch_zero_words(v->stream, &dc->shard_need, 2, m->shard_need_bits.as<uint32_t>(), ((tiles_x + 31) / 32) * tiles_y));
    m->verify_state_zeroed = false;
    const gsx_viewer* o = v->parent ? v->parent : v;
    // the verdict block goes to DEVICE memory: nobody waits for it; k_shard_post_verdict posts it behind the repair round
    HIPCHK(launch_shard_verify(v->stream, m->shard_frame_limited ? m->shard_limit.as<uint32_t>() : nullptr, static_cast<const uint32_t*>(d_sat_all),
                               tiles_x, tiles_y, bands_of(v, world), m->shard_win2.as<uint2>(), &dc->shard_need, &dc->shard_ticket,
                               v->verdict_stage.as<unsigned long long>(), 0u, m->shard_need_bits.as<uint32_t>(), o->shard_balance ? 1u : 0u));
    m->repair_counted = false;
    m->stats_pending = true;
    return GSX_OK;
}

gsx_status gsx::shard_post_verdict(gsx_viewer* v, uint32_t world, const void* d_sat_after_repair, uint32_t* out_seq) {
    if (!v->h_verdict_ring || !v->ring_slots) return fail(GSX_ERR_INVALID_ARG, "shard_post_verdict: no verdict ring");
    *out_seq = ++v->ring_seq;
    if (*out_seq == 0) *out_seq = ++v->ring_seq;  // (0 = "nothing posted yet" in a fresh slot)
    uint32_t* block = v->h_verdict_ring + (size_t)(*out_seq % v->ring_slots) * kVerdictWords;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(launch_shard_post_verdict(v->stream, v->verdict_stage.as<uint32_t>(), static_cast<const uint32_t*>(d_sat_after_repair), world,
                                     feedback_stride(bands_of(v, world), tiles_x), block, *out_seq));
    return GSX_OK;
}

// Spins on slot seq % ring of the pinned ring until the kernel that posts `seq` has run (normally it has: the frame was enqueued a
// call or two ago).  Bounded like gsx_shard_wait_verdict.  *block: the whole verdict block (kVerdict* layout).
gsx_status gsx::shard_wait_ring(gsx_viewer* v, uint32_t seq, gsx_shard_verdict* out, const uint32_t** block) {
    if (!v || !out || !v->h_verdict_ring) return fail(GSX_ERR_INVALID_ARG, "shard_wait_ring: nothing was posted");
    const uint32_t* b = v->h_verdict_ring + (size_t)(seq % v->ring_slots) * kVerdictWords;
    const unsigned long long* w64 = reinterpret_cast<const unsigned long long*>(b);
    const auto t_start = std::chrono::steady_clock::now();
    trace_flush();
    for (uint64_t spin = 1;; ++spin) {
        const unsigned long long w = __atomic_load_n(&w64[0], __ATOMIC_ACQUIRE);
        if ((uint32_t)(w >> 32) == seq) {
            const unsigned long long d = __atomic_load_n(&w64[1], __ATOMIC_RELAXED);
            out->need_tiles = (uint32_t)w;
            out->overflow = (uint32_t)(d & 1ull);
            out->max_records = (uint32_t)(d >> 32);
            if (block) *block = b;
            return GSX_OK;
        }
        if ((spin & 0xFFFu) == 0) {
            const hipError_t e = gsx::op::StreamQuery(v->stream);
            if (e == hipSuccess) {
                const unsigned long long w2 = __atomic_load_n(&w64[0], __ATOMIC_ACQUIRE);
                if ((uint32_t)(w2 >> 32) == seq) continue;
                return fail(GSX_ERR_HIP, "sharded frame: verdict %u never arrived (stream idle)", seq);
            }
            if (e != hipErrorNotReady) return fail(GSX_ERR_HIP, "stream failed while waiting for a sharded frame's verdict: %s", hipGetErrorString(e));
            if ((spin & 0xFFFFFu) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                return fail(GSX_ERR_RCCL, "sharded frame: verdict %u did not arrive within 60 s (a collective is stuck)", seq);
        }
        // (pure spinning: with one frame in flight this wait IS the frame, and a sleep's wake-up — tens of microseconds of timer slack —
        //  would be a bubble on the device every frame: 1307 against 1445 fps on cfg4 at world 1).  A wait that has outlasted any
        //  frame yields its core between looks: ranks that share a host with fewer cores than spinning threads (RCCL's proxy threads
        //  spin too) must not starve the thread that would complete the collective.
        if (spin > 200000u) sched_yield();
        else __builtin_ia32_pause();
    }
}

extern "C" {

// Spins on the pinned verdict words until the kernel that posts `seq` has run.  Bounded by the stream itself: if the stream
// drains (or fails) and the words still are not there, that is reported instead of spinning forever.
gsx_status gsx_shard_wait_verdict(gsx_viewer* v, const char* key, uint32_t seq, gsx_shard_verdict* out) {
    if (!v || !out || !v->h_shard_verdict) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_wait_verdict: nothing was posted");
    // a verdict that does not come within a minute is a collective that cannot complete (a rank died, or the ranks disagree
    // about what to exchange): an error the caller can report, not a process that spins for ever
    const auto t_start = std::chrono::steady_clock::now();
    trace_flush();  // the kernel that posts the words may still be in a recorded segment
    for (uint64_t spin = 1;; ++spin) {
        const unsigned long long w = __atomic_load_n(&v->h_shard_verdict[0], __ATOMIC_ACQUIRE);
        if ((uint32_t)(w >> 32) == seq) {
            const unsigned long long d = __atomic_load_n(&v->h_shard_verdict[1], __ATOMIC_RELAXED);
            out->need_tiles = (uint32_t)w;
            out->overflow = (uint32_t)(d & 1ull);
            out->max_records = (uint32_t)(d >> 32);
            if (Model* m = find_model(v, key)) {  // next frame's round-0 slots (global: same on every rank)
                Model* hm = m;
                if (v->parent)
                    if (Model* om = find_model(v->parent, key)) hm = om;
                hm->slot_hint = out->max_records;
                hm->slot_hint_known = true;
                hm->slot_hint_limited = m->shard_frame_limited;
            }
            return GSX_OK;
        }
        if ((spin & 0xFFFu) == 0) {
            const hipError_t e = gsx::op::StreamQuery(v->stream);
            if (e == hipSuccess) {
                const unsigned long long w2 = __atomic_load_n(&v->h_shard_verdict[0], __ATOMIC_ACQUIRE);
                if ((uint32_t)(w2 >> 32) == seq) continue;
                return fail(GSX_ERR_HIP, "gsx_shard_wait_verdict: verdict %u never arrived (stream idle)", seq);
            }
            if (e != hipErrorNotReady) return fail(GSX_ERR_HIP, "stream failed while waiting for the exchange verdict: %s", hipGetErrorString(e));
            if ((spin & 0xFFFFFu) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(60))
                return fail(GSX_ERR_RCCL, "gsx_shard_wait_verdict: verdict %u did not arrive within 60 s (a collective is stuck)", seq);
        }
        __builtin_ia32_pause();
    }
}

gsx_status gsx_shard_next_windows(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, float margin, uint32_t radius) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_sat_all) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_next_windows: no model '%s' / null map", key ? key : "(null)");
    if (!(margin >= 0.0f) || radius > 16 || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_next_windows: margin >= 0, radius <= 16, world 1..64");
    if ((st = check_bands(v, world, "gsx_shard_next_windows"))) return st;
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    // the limits of the frame in flight must survive until its verdict is in (a repair round reads them): double buffer
    DevBuf& next = m->shard_limit_next;
    HIPCHK(next.ensure(4 * (size_t)tiles_x * tiles_y));
    HIPCHK(launch_shard_next_limits(v->stream, static_cast<const uint32_t*>(d_sat_all), tiles_x, tiles_y, margin, radius, next.as<uint32_t>(),
                                    bands_of(v, world)));
    m->shard_next_valid = true;
    m->shard_win_next_valid = false;  // (these limits come without their windows: whatever shard_win_next holds belongs to older ones)
    m->shard_limit_tx = tiles_x;
    m->shard_limit_ty = tiles_y;
    return GSX_OK;
}

}  // extern "C"

// gsx_shard_next_windows for the frames of gsx_shard_frame.cpp: the same limits, plus — in the same launch — the windows [0, limit) of the
// model's next frame and the frame's verdict (what shard_post_verdict does as a launch of its own)
gsx_status gsx::shard_next_windows_post(gsx_viewer* v, const char* key, uint32_t world, const void* d_sat_all, float margin, uint32_t radius,
                                        const void* d_sat_after_repair, uint32_t* out_seq) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !d_sat_all || !out_seq) return fail(GSX_ERR_NOT_FOUND, "shard_next_windows_post: no model '%s' / null argument", key ? key : "(null)");
    if (!(margin >= 0.0f) || radius > 16 || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_render_frame: margin >= 0, radius <= 16, world 1..64");
    if ((st = check_bands(v, world, "gsx_shard_render_frame"))) return st;
    if (!v->h_verdict_ring || !v->ring_slots) return fail(GSX_ERR_INVALID_ARG, "shard_next_windows_post: no verdict ring");
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    HIPCHK(m->shard_limit_next.ensure(4 * (size_t)tiles_x * tiles_y));
    HIPCHK(m->shard_win_next.ensure(window_bytes(v)));
    *out_seq = ++v->ring_seq;
    if (*out_seq == 0) *out_seq = ++v->ring_seq;
    uint32_t* block = v->h_verdict_ring + (size_t)(*out_seq % v->ring_slots) * kVerdictWords;
    HIPCHK(launch_shard_next_limits(v->stream, static_cast<const uint32_t*>(d_sat_all), tiles_x, tiles_y, margin, radius, m->shard_limit_next.as<uint32_t>(),
                                    bands_of(v, world), m->shard_win_next.as<uint2>(), v->verdict_stage.as<uint32_t>(),
                                    static_cast<const uint32_t*>(d_sat_after_repair), block, *out_seq));
    m->shard_next_valid = true;
    m->shard_win_next_valid = true;
    m->shard_limit_tx = tiles_x;
    m->shard_limit_ty = tiles_y;
    return GSX_OK;
}

extern "C" {

gsx_status gsx_shard_frame_end(gsx_viewer* v, const char* key) {
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_frame_end: no model '%s'", key ? key : "(null)");
    if (m->shard_next_valid) {  // the limits gsx_shard_next_windows computed become the next frame's
        std::swap(m->shard_limit.p, m->shard_limit_next.p);
        std::swap(m->shard_limit.bytes, m->shard_limit_next.bytes);
        m->shard_limit_valid = true;
        m->shard_next_valid = false;
        // ... and their windows, where the same kernel wrote them
        m->shard_win_current = false;
        if (m->shard_win_next_valid) {
            std::swap(m->shard_win.p, m->shard_win_next.p);
            std::swap(m->shard_win.bytes, m->shard_win_next.bytes);
            m->shard_win_current = true;
            m->shard_win_next_valid = false;
        }
    }
    return GSX_OK;
}

gsx_status gsx_shard_set_limits(gsx_viewer* v, const char* key, const uint32_t* limits) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !limits) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_set_limits: no model '%s' / null limits", key ? key : "(null)");
    const uint32_t tiles_x = (v->width + GSX_TILE - 1) / GSX_TILE, tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    const size_t bytes = 4 * (size_t)tiles_x * tiles_y;
    // kept with the model until its next sharded frame picks it up — on whichever lane that frame runs (gsx_shard_frame.cpp)
    HIPCHK(m->shard_limit_override.ensure(bytes));
    hipPointerAttribute_t attr{};
    const bool on_device = hipPointerGetAttributes(&attr, limits) == hipSuccess && attr.type == hipMemoryTypeDevice;
    if (on_device) {
        HIPCHK(gsx::op::MemcpyAsync(m->shard_limit_override.p, limits, bytes, hipMemcpyDeviceToDevice, v->stream));
    } else {
        (void)hipGetLastError();
        HIPCHK(gsx::op::StreamSynchronize(v->stream));
        HIPCHK(gsx::op::Memcpy(m->shard_limit_override.p, limits, bytes, hipMemcpyHostToDevice));
    }
    m->shard_override_tiles = tiles_x * tiles_y;
    return GSX_OK;
}

gsx_status gsx_shard_set_slot_records(gsx_viewer* v, const char* key, uint32_t records) {
    Model* m = find_model(v, key);
    if (!m) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_set_slot_records: no model '%s'", key ? key : "(null)");
    m->slot_force = records;
    for (gsx_viewer* l : v->lanes)
        if (Model* sm = find_model(l, key)) sm->slot_force = records;
    return GSX_OK;
}

gsx_status gsx_shard_get_stats(gsx_viewer* v, gsx_shard_stats* out, uint32_t reset) {
    if (!v || !out) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_get_stats: null argument");
    *out = v->shard_stats;
    if (reset) v->shard_stats = gsx_shard_stats{};
    return GSX_OK;
}

gsx_status gsx_shard_set_gather_root(gsx_viewer* v, int32_t root) {
    gsx_status st = viewer_bind(v);  // (finishes the sharded frames in flight: they were promised the old destination)
    if (st) return st;
    if (v->parent) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_gather_root: called on a lane");
    if (root < -1 || root >= 64 || (v->comm_world && root >= (int32_t)v->comm_world))
        return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_gather_root: root %d (-1 = every rank, else a rank of the %u)", (int)root, v->comm_world);
    v->shard_gather_root = root;
    v->shard_root_confirmed = false;  // the next frame's verdict tells whether every rank named the same root; it gathers only then
    return GSX_OK;
}

gsx_status gsx_shard_set_band_edges(gsx_viewer* v, uint32_t world, const uint32_t* edges) {
    gsx_status st = viewer_bind(v);  // (frames in flight keep the layout they were enqueued with)
    if (st) return st;
    if (v->parent) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_band_edges: called on a lane");
    if (!edges) {
        v->band_edges_forced.clear();
        v->band_edges.clear();
        return GSX_OK;
    }
    if (world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_band_edges: world must be 1..64");
    const uint32_t tiles_y = (v->height + GSX_TILE - 1) / GSX_TILE;
    if (edges[0] != 0 || edges[world] < tiles_y || edges[world] > 0xFFFFu)
        return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_band_edges: edges[0] must be 0 and edges[world] = %u must cover the %u tile rows", edges[world], tiles_y);
    for (uint32_t g = 0; g < world; ++g)
        if (edges[g] > edges[g + 1]) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_band_edges: edges must not decrease (edge %u)", g + 1);
    v->band_edges_forced.assign(edges, edges + world + 1);
    v->band_edges = v->band_edges_forced;  // the stage calls read the viewer's layout
    return GSX_OK;
}

gsx_status gsx_shard_get_band_edges(gsx_viewer* v, uint32_t world, uint32_t* out_edges) {
    if (!v || !out_edges || world == 0 || world > 64) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_get_band_edges: bad argument");
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (v->last_edges.size() == (size_t)world + 1u) {
        for (uint32_t g = 0; g <= world; ++g) out_edges[g] = v->last_edges[g];
    } else {
        const BandEdges b = bands_of(v, world);
        for (uint32_t g = 0; g <= world; ++g) out_edges[g] = b.e[g];
    }
    return GSX_OK;
}

gsx_status gsx_shard_set_balance(gsx_viewer* v, uint32_t enabled) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (v->parent) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_set_balance: called on a lane");
    v->shard_balance = enabled != 0;
    return GSX_OK;
}

gsx_status gsx_shard_download_limits(gsx_viewer* v, const char* key, uint32_t* limits, uint64_t n_words) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    Model* m = find_model(v, key);
    if (!m || !limits) return fail(GSX_ERR_NOT_FOUND, "gsx_shard_download_limits: no model '%s'", key ? key : "(null)");
    const uint64_t n_tiles = (uint64_t)m->shard_limit_tx * m->shard_limit_ty;
    if (!m->shard_limit_valid || n_words < n_tiles) return fail(GSX_ERR_INVALID_ARG, "gsx_shard_download_limits: no limits yet, or buffer too small");
    HIPCHK(gsx::op::MemcpyAsync(limits, m->shard_limit.p, 4 * n_tiles, hipMemcpyDeviceToHost, v->stream));
    HIPCHK(gsx::op::StreamSynchronize(v->stream));
    return GSX_OK;
}

}  // extern "C"
