// gsx_api_shard.cpp — C ABI for multi-GPU rendering: band layout, external framebuffer, screen bands, and the stage split of
// the index-sharded exchange (pack / import / feedback / second round).  No reference counterpart (src/main.rs:85-98).
#include "gsx_state.h"

using namespace gsx;

extern "C" {

// ---- multi-GPU stage split ----------------------------------------------------------------------

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


gsx_status gsx_resolve_rgba8_device(gsx_viewer* v, const float bg[3], uint32_t y0, uint32_t y1, void* d_rgba) {
    gsx_status st = viewer_bind(v);
    if (st) return st;
    if (!bg || !d_rgba || y1 < y0) return fail(GSX_ERR_INVALID_ARG, "gsx_resolve_rgba8_device: null argument or y1 < y0");
    if ((st = ensure_fb(v))) return st;
    const uint64_t rows_avail = v->ext_fb ? v->ext_fb_bytes / (sizeof(float4) * (uint64_t)v->width) : v->height;
    if (y1 > rows_avail) return fail(GSX_ERR_INVALID_ARG, "gsx_resolve_rgba8_device: rows [%u, %u) of a %llu-row framebuffer", y0, y1, (unsigned long long)rows_avail);
    const uint64_t npx = (uint64_t)(y1 - y0) * v->width;
    HIPCHK(launch_resolve_rgba8(v->stream, fb_ptr(v) + (size_t)y0 * v->width, (uint32_t)npx, bg[0], bg[1], bg[2], static_cast<uint32_t*>(d_rgba)));
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

}  // extern "C"
