// kernels_bin.hip — tile binning for gfx950: per-splat tile counts in depth order, exclusive scan,
// duplicate emission of (tile id, Gaussian index) pairs, and per-tile ranges of the tile-sorted list.
//
// The reference has no screen tiles (it draws one instanced quad per surviving Gaussian and lets the
// ROP blend, src/tab/scene.rs:2306-2313); this stage is the build's replacement for that rasteriser
// front-end.  Pairs are emitted in front-to-back depth order, so a STABLE sort by tile id alone
// (kernels_sort.hip, 2 passes for <= 65536 tiles) leaves every tile's list depth-ordered.
// Integer-only; must be bit-exact against oracle/gsx_oracle.c:gsxo_tile_lists.
// Algorithmic bytes: N_vis*44 + D*12.
#include <atomic>
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

constexpr int kBinThreads = 256;  // one splat per lane
constexpr uint32_t kCoopThreshold = 24;  // rectangles with more tiles are expanded by the whole wave
constexpr uint32_t kBinGrid = 256 * 24;  // workgroups of the chunk-striding kernels: 24 per CU keeps the gathers in flight

size_t scan_blocks(uint64_t n) { return (size_t)((n + kBinThreads - 1) / kBinThreads); }

// Multi-GPU: a rank bins only the band of tile rows it owns, [row_lo, row_hi); one GPU owns [0, tiles_y).

__device__ inline uint32_t rect_area(float4 a, uint32_t row_lo, uint32_t row_hi) {
    uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
    uint32_t first = max(ry & 0xFFFFu, row_lo), last = min(ry >> 16, row_hi);
    uint32_t rows = first < last ? last - first : 0u;
    return ((rx >> 16) - (rx & 0xFFFFu)) * rows;
}

// Progressive slabs: tiles whose every pixel is already saturated (T < t_epsilon) are flagged in a bitmap
// (one bit per tile, `row_words` 32-bit words per tile row) and receive no further entries.
// Number of NOT-done tiles of row `ty` in [x0, x1).
__device__ inline uint32_t live_tiles_in_row(const uint32_t* done, uint32_t row_words, uint32_t ty, uint32_t x0, uint32_t x1) {
    uint32_t n = 0;
    const uint32_t* row = done + ty * row_words;
    for (uint32_t w = x0 >> 5; w <= ((x1 - 1u) >> 5); ++w) {
        uint32_t lo = w == (x0 >> 5) ? (x0 & 31u) : 0u;
        uint32_t hi = w == ((x1 - 1u) >> 5) ? ((x1 - 1u) & 31u) : 31u;
        uint32_t mask = (hi == 31u ? 0xFFFFFFFFu : ((1u << (hi + 1u)) - 1u)) & ~((1u << lo) - 1u);
        n += __popc(~row[w] & mask);
    }
    return n;
}

__device__ inline uint32_t rect_live_area(float4 a, uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words) {
    uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
    uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16;
    uint32_t n = 0;
    for (uint32_t ty = max(ry & 0xFFFFu, row_lo), last = min(ry >> 16, row_hi); ty < last; ++ty)
        n += live_tiles_in_row(done, row_words, ty, x0, x1);
    return n;
}

// Multi-GPU speculation (kernels_shard.hip): every tile has a depth-key window [lo, hi); a tile takes a record only
// if the record's key lies inside it (and the tile is not saturated).  window == nullptr on one GPU.
struct TileWindow {
    const uint2* win;             // [tiles_y * tiles_x] or nullptr
    const uint32_t* sorted_keys;  // key of depth-order position j
    uint32_t tiles_x;
    WindowPyramid min_ends;       // .data nullable: min-pyramid of the window ends (every window starts at 0)
};

// srect.x bit 31 (set by k_tile_counts, read by k_tile_emit): every tile of the rectangle takes the splat — no per-tile test
constexpr uint32_t kAllTake = 0x80000000u;

// true if EVERY tile of the (band-clipped, non-empty) rectangle [x0, xb] x [y0, yb] has a window end above `key`: the
// rectangle lies under at most 2x2 cells of the level its extent selects, and a cell holds the minimum over its tiles
__device__ inline bool pyramid_all_take(const WindowPyramid& p, uint32_t key, uint32_t x0, uint32_t xb, uint32_t y0, uint32_t yb) {
    const uint32_t ext = max(xb - x0, yb - y0);  // extent - 1
    const uint32_t l = ext ? 32u - (uint32_t)__clz((int)ext) : 0u;
    if (l >= p.levels) return false;
    const uint32_t* L = p.data + p.off[l];
    const uint32_t wx = p.wx[l];
    const uint32_t cx0 = x0 >> l, cx1 = min(xb >> l, wx - 1u), cy0 = y0 >> l, cy1 = min(yb >> l, p.wy[l] - 1u);
    return key < min(min(L[cy0 * wx + cx0], L[cy0 * wx + cx1]), min(L[cy1 * wx + cx0], L[cy1 * wx + cx1]));
}

__device__ inline bool tile_takes(const uint32_t* done, uint32_t row_words, const uint2* win, uint32_t tiles_x, uint32_t tx,
                                  uint32_t ty, uint32_t key) {
    if (done && ((done[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u)) return false;
    const uint2 w = win[ty * tiles_x + tx];
    return key >= w.x && key < w.y;
}

__device__ inline uint32_t window_tiles_in_row(const uint32_t* done, uint32_t row_words, const uint2* win, uint32_t tiles_x,
                                               uint32_t ty, uint32_t x0, uint32_t x1, uint32_t key) {
    uint32_t n = 0;
    for (uint32_t tx = x0; tx < x1; ++tx) n += tile_takes(done, row_words, win, tiles_x, tx, ty, key) ? 1u : 0u;
    return n;
}

__device__ inline uint32_t block_reduce_sum(uint32_t v, uint32_t* smem4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63u) == 0) smem4[threadIdx.x >> 6] = v;
    __syncthreads();
    return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// Slab splat j in [j0, min(j1, N_vis)): gather its tile rectangle once (-> srect[j - j0], reused by the
// emit kernel), count its live tiles (-> cnt[j - j0]) and reduce per 256-splat chunk (-> block_sums[chunk]).
// One splat per lane: the gather rec_a[sorted_idx[j]] is a dependent random access, so the pass lives on
// memory-level parallelism.  Workgroups stride over the chunks that exist ON THE DEVICE (N_vis is only known
// there), so a slab bound far above N_vis — the host plans slabs from an upper bound — costs nothing.
// done == nullptr: every tile is live.
__global__ __launch_bounds__(kBinThreads) void k_tile_counts(const uint32_t* __restrict__ d_n_vis, uint32_t j0,
                                                              uint32_t j1, const uint32_t* __restrict__ sorted_idx,
                                                              const float4* __restrict__ rec_a,
                                                              uint2* __restrict__ srect, uint32_t* __restrict__ cnt,
                                                              uint32_t* __restrict__ block_sums, uint32_t row_lo,
                                                              uint32_t row_hi, const uint32_t* __restrict__ done,
                                                              uint32_t row_words,
                                                              const uint32_t* __restrict__ d_done_count,
                                                              uint32_t owned_tiles, TileWindow tw) {
    __shared__ uint32_t red[4];
    const uint32_t n_vis = min(*d_n_vis, j1);
    const uint32_t chunks = n_vis > j0 ? (n_vis - j0 + kBinThreads - 1) / kBinThreads : 0u;
    // every tile this rank composites is saturated: whatever is left is hidden, skip the gather
    const bool all_done = d_done_count && *d_done_count >= owned_tiles;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        if (all_done) {
            if (threadIdx.x == 0) block_sums[chunk] = 0;
            continue;
        }
        const uint32_t j = j0 + chunk * kBinThreads + threadIdx.x;
        uint32_t c = 0, rx = 0, ry = 0, key = 0, area = 0;
        if (j < n_vis) {
            const float4 a = rec_a[sorted_idx[j]];
            rx = __float_as_uint(a.z);
            ry = __float_as_uint(a.w);
            if (tw.win) {
                key = tw.sorted_keys[j];
                area = rect_area(a, row_lo, row_hi);
                // most splats of a speculated round lie in front of every window end under them: four pyramid loads
                // instead of one dependent load per tile (the per-tile tests are bound by the L1's line rate: 64 lanes,
                // 64 different lines per instruction)
                if (area && !done && tw.min_ends.data &&
                    pyramid_all_take(tw.min_ends, key, rx & 0xFFFFu, (rx >> 16) - 1u, max(ry & 0xFFFFu, row_lo), min(ry >> 16, row_hi) - 1u)) {
                    c = area;
                    area = 0;  // nothing left to test
                    rx |= kAllTake;
                } else if (area <= kCoopThreshold)
                    for (uint32_t ty = max(ry & 0xFFFFu, row_lo), last = min(ry >> 16, row_hi); ty < last; ++ty)
                        c += window_tiles_in_row(done, row_words, tw.win, tw.tiles_x, ty, rx & 0xFFFFu, rx >> 16, key);
            } else {
                c = done ? rect_live_area(a, row_lo, row_hi, done, row_words) : rect_area(a, row_lo, row_hi);
            }
            srect[j - j0] = make_uint2(rx, ry);
        }
        if (tw.win) {
            // large rectangles: the whole wave evaluates the window predicate, lane l takes tiles l, l + 64, ...
            unsigned long long big = __ballot(area > kCoopThreshold);
            const uint32_t lane = threadIdx.x & 63u;
            while (big) {
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1;
                const uint32_t brx = __shfl(rx, src, 64), bry = __shfl(ry, src, 64), bkey = __shfl(key, src, 64);
                const uint32_t total = __shfl(area, src, 64);
                const uint32_t x0 = brx & 0xFFFFu, w = (brx >> 16) - x0, first = max(bry & 0xFFFFu, row_lo);
                uint32_t n = 0;
                for (uint32_t k = lane; k < total; k += 64)
                    n += tile_takes(done, row_words, tw.win, tw.tiles_x, x0 + k % w, first + k / w, bkey) ? 1u : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
                if ((int)lane == src) c = n;
            }
        }
        if (j < n_vis) cnt[j - j0] = c;
        const uint32_t tot = block_reduce_sum(c, red);
        if (threadIdx.x == 0) block_sums[chunk] = tot;
        __syncthreads();  // red[] is reused by the next chunk
    }
}

// single workgroup: exclusive scan of block_sums in place; slab total D -> stats->n_entries (what fits the pair buffers:
// see the cut below)
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ sums, uint32_t j0, uint32_t j1,
                                                           const uint32_t* __restrict__ d_n_vis,
                                                           SlabStats* __restrict__ stats, uint32_t capacity,
                                                           const uint32_t* __restrict__ d_done_count,
                                                           uint32_t owned_tiles, uint32_t slab_index) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_vis = min(*d_n_vis, j1);
    const uint32_t nblocks = n_vis > j0 ? (n_vis - j0 + kBinThreads - 1) / kBinThreads : 0u;  // chunks that exist
    if (tid == 0) carry_s = 0;
    __syncthreads();
    // eight consecutive sums per thread and trip (a 4 M-splat slab has 16 K chunks: two trips instead of sixteen — the kernel is
    // one workgroup on the critical path of every slab, 11 -> 4 us)
    constexpr uint32_t kPer = 8;
    for (uint32_t base = 0; base < nblocks; base += 1024 * kPer) {
        const uint32_t i0 = base + tid * kPer;
        uint32_t v[kPer], x = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            v[k] = i0 + k < nblocks ? sums[i0 + k] : 0u;
            x += v[k];
        }
        const uint32_t mine = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wave; ++w) woff += wsum[w];
        const uint32_t carry = carry_s;
        uint32_t run = carry + woff + x - mine;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            if (i0 + k < nblocks) sums[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    // More entries than the pair buffers hold: the slab is CUT after the last 256-splat chunk whose entries still fit.  The
    // chunks before the cut are binned, tile-sorted and composited as usual; k_composite_spill then composites the splats
    // behind the cut straight from the depth order (slow, pair-free, same per-pixel operation sequence), so the frame is
    // complete whatever the capacity — the host only learns (lazily) that it should grow the buffers.
    __shared__ uint32_t fit_s;
    const uint32_t total = carry_s;
    if (tid == 0) fit_s = 0;
    __syncthreads();
    if (total > capacity) {
        for (uint32_t i = tid; i < nblocks; i += 1024) {
            const uint32_t incl = i + 1 < nblocks ? sums[i + 1] : total;  // inclusive prefix of chunk i (monotone in i)
            if (incl <= capacity) atomicMax(&fit_s, i + 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const bool over = total > capacity;
        const uint32_t fit = over ? fit_s : nblocks;
        stats->n_entries = over ? (fit < nblocks ? sums[fit] : total) : total;
        stats->slab_cut = over ? min(j0 + fit * (uint32_t)kBinThreads, n_vis) : max(n_vis, j0);
        stats->n_entries_total += total;
        stats->max_needed = max(stats->max_needed, total);
        // the host sizes the next frame's slab plan from this (read lazily, never waited for)
        if (nblocks && !(d_done_count && *d_done_count >= owned_tiles)) stats->slabs_used = max(stats->slabs_used, slab_index + 1u);
        if (over) {
            stats->overflow = 1;
            stats->overflow_events += 1;
            stats->max_needed_ever = max(stats->max_needed_ever, total);
        }
    }
}

// emit interleaved {tile id, Gaussian index} pairs of slab splat j at offset = block_offs[workgroup] +
// exclusive scan of cnt inside the workgroup.  Reads only sequential arrays (srect, cnt, sorted_idx).
// Splats touching up to kCoopThreshold live tiles are written by their own lane; the few large ones are
// expanded cooperatively by the whole wave (lane l writes tiles l, l+64, ...), which removes the long
// divergent tail a single lane would otherwise serialise.
__device__ inline void emit_rect(uint2* __restrict__ tpairs, uint32_t o, uint32_t capacity, uint32_t idx, uint2 r,
                                 uint32_t tiles_x, uint32_t row_lo, uint32_t row_hi, const uint32_t* __restrict__ done,
                                 uint32_t row_words, const uint2* __restrict__ win, uint32_t key) {
    const uint32_t x0 = r.x & 0xFFFFu, x1 = r.x >> 16, y0 = r.y & 0xFFFFu, y1 = r.y >> 16;
    for (uint32_t ty = max(y0, row_lo), last = min(y1, row_hi); ty < last; ++ty)
        for (uint32_t tx = x0; tx < x1; ++tx) {
            if (win) {
                if (!tile_takes(done, row_words, win, tiles_x, tx, ty, key)) continue;
            } else if (done && ((done[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u)) {
                continue;
            }
            if (o < capacity) tpairs[o] = make_uint2(ty * tiles_x + tx, idx);
            ++o;
        }
}

__global__ __launch_bounds__(kBinThreads) void k_tile_emit(uint32_t jbase, uint32_t n_vis /* upper bound j1 */,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint2* __restrict__ srect,
                                                            const uint32_t* __restrict__ cnt,
                                                            const uint32_t* __restrict__ block_offs, uint32_t tiles_x,
                                                            uint2* __restrict__ tpairs, uint32_t row_lo, uint32_t row_hi,
                                                            const uint32_t* __restrict__ done, uint32_t row_words,
                                                            const uint32_t* __restrict__ d_n_vis,
                                                            const uint32_t* __restrict__ d_entries, uint32_t capacity,
                                                            TileWindow tw, const uint32_t* __restrict__ d_cut) {
    __shared__ uint32_t wsum[4];
    if (*d_entries == 0) return;  // empty slab (also: every tile already saturated)
    n_vis = min(min(n_vis, *d_n_vis), *d_cut);  // an overflowing slab is binned up to its cut (k_scan_block_sums)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunks = n_vis > jbase ? (n_vis - jbase + kBinThreads - 1) / kBinThreads : 0u;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    const uint32_t j = jbase + chunk * kBinThreads + tid;
    const uint32_t mine = j < n_vis ? cnt[j - jbase] : 0u;
    uint32_t x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t o = block_offs[chunk] + x - mine;
    for (uint32_t w = 0; w < wave; ++w) o += wsum[w];
    uint32_t idx = 0;
    uint2 r = make_uint2(0, 0);
    uint32_t key = 0;
    bool all = false;  // k_tile_counts found every tile of the rectangle taking it (kAllTake; implies no saturation bitmap)
    if (mine) {
        idx = sorted_idx[j];
        r = srect[j - jbase];
        all = (r.x & kAllTake) != 0u;
        r.x &= ~kAllTake;
        if (tw.win && !all) key = tw.sorted_keys[j];
    }
    // a lane walks its own rectangle only if that is short; large rectangles (even with few takers left) go to the wave
    uint32_t area = 0;
    if (mine) {
        const uint32_t ya = max(r.y & 0xFFFFu, row_lo), yb = min(r.y >> 16, row_hi);
        area = ((r.x >> 16) - (r.x & 0xFFFFu)) * (yb > ya ? yb - ya : 0u);
    }
    if (mine && area <= kCoopThreshold)
        emit_rect(tpairs, o, capacity, idx, r, tiles_x, row_lo, row_hi, done, row_words, all ? nullptr : tw.win, key);
    // large splats: one at a time, all 64 lanes
    unsigned long long big = __ballot(area > kCoopThreshold);
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t bo = __shfl(o, src, 64), bidx = __shfl(idx, src, 64);
        const uint32_t rx = __shfl(r.x, src, 64), ry = __shfl(r.y, src, 64);
        const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
        const uint32_t w = x1 - x0;
        const uint32_t bkey = __shfl(key, src, 64);
        const bool ball = __shfl((int)all, src, 64) != 0;
        if (!done && (!tw.win || ball)) {
            // no saturated tiles: the k-th entry is tile (first + k / w, x0 + k % w)
            const uint32_t first = max(y0, row_lo);
            const uint32_t total = __shfl(mine, src, 64);
            for (uint32_t k = lane; k < total; k += 64) {
                const uint32_t ty = first + k / w, tx = x0 + k % w;
                if (bo + k < capacity) tpairs[bo + k] = make_uint2(ty * tiles_x + tx, bidx);
            }
        } else {
            // with saturated tiles / windows the slots are not a closed form: lanes stride over the rectangle's tiles,
            // a ballot gives every taker its slot (the order of one splat's entries is irrelevant: the tile sort keys
            // on the tile id and each (tile, splat) pair is unique)
            const uint32_t first = max(y0, row_lo), rows = min(y1, row_hi) - first, total = w * rows;
            const unsigned long long lt = (1ull << lane) - 1ull;
            uint32_t base = bo;
            for (uint32_t k0 = 0; k0 < total; k0 += 64) {
                const uint32_t k = k0 + lane;
                bool take = false;
                uint32_t tx = 0, ty = 0;
                if (k < total) {
                    ty = first + k / w;
                    tx = x0 + k % w;
                    take = tw.win ? tile_takes(done, row_words, tw.win, tiles_x, tx, ty, bkey)
                                  : !((done[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u);
                }
                const unsigned long long bal = __ballot(take);
                if (take) {
                    const uint32_t oo = base + (uint32_t)__popcll(bal & lt);
                    if (oo < capacity) tpairs[oo] = make_uint2(ty * tiles_x + tx, bidx);
                }
                base += (uint32_t)__popcll(bal);
            }
        }
    }
    __syncthreads();  // wsum[] is reused by the next chunk
    }
}

// ranges[t] = [first, last+1) of tile t in the tile-sorted pair list (ranges pre-zeroed)
__global__ __launch_bounds__(256) void k_tile_ranges(const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ tkey,
                                                      uint2* __restrict__ ranges) {
    const uint32_t D = *d_n;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < D; e += gridDim.x * 256u) {
        uint32_t t = tkey[e];
        if (e == 0 || tkey[e - 1] != t) ranges[t].x = e;
        if (e == D - 1 || tkey[e + 1] != t) ranges[t].y = e + 1;
    }
}

// ---- block lists: binning by BLOCKS of tiles (progressive frames) ----
// The screen's tiles are grouped into at most 256 blocks of 2^bsx x 2^bsy tiles (1920x1080: 15 x 17 blocks of 8 x 4 tiles).
// A slab's records are binned by block — a record makes one entry per block its rectangle touches, a few instead of one per
// tile — the entries are sorted by block id in ONE 8-bit radix pass, and the compositor of a tile walks its block's list and
// keeps the records whose rectangle and depth-key window take the tile (k_composite_blocks).  The per-tile predicate thus
// costs one register compare where the list is consumed instead of one dependent load per (record, tile) where it is
// built, and the sort moves 1/4 of the entries once instead of all of them twice.  Block-level tests here are
// CONSERVATIVE (a block takes a record unless all its tiles are saturated / outside the band, or no window of its tiles
// can contain the key); the exact decision is the compositor's.  Entry value = the record's position in the slab;
// brec[position] = {rect x, rect y, depth key, record index}.
// (BlockGrid, wave_block_table_entry: gsx_internal.h — kernels_spec.hip builds the repair round's table in its verification kernel)
// table[b] = {min window start, max window end (of the non-empty windows of the block's live tiles), live}; also zeroes the
// block's range (the block sort fills in the blocks that have entries).  One wave per block, a lane per tile
// (wave_block_table_entry, gsx_internal.h).  za / zb: words to zero on the way — the frame's saturation state and counters, when
// this is the first slab of the frame (one launch less: k_zero_words).
__global__ __launch_bounds__(256) void k_block_table(BlockGrid g, uint32_t tiles_x, uint32_t tiles_y, uint32_t row_lo, uint32_t row_hi,
                                                     const uint32_t* __restrict__ done, uint32_t row_words,
                                                     const uint2* __restrict__ win, uint4* __restrict__ table,
                                                     uint2* __restrict__ ranges, uint32_t* __restrict__ za, uint32_t nza,
                                                     uint32_t* __restrict__ zb, uint32_t nzb, uint32_t* __restrict__ live_cells,
                                                     const uint32_t* __restrict__ copy_src, uint32_t* __restrict__ copy_dst, uint32_t n_copy) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < max(max(nza, nzb), n_copy); i += gridDim.x * 256u) {
        if (i < nza) za[i] = 0u;
        if (i < nzb) zb[i] = 0u;
        if (i < n_copy) copy_dst[i] = copy_src[i];   // (ZeroJob: the saturation bitmap as the models in front left it)
    }
    const uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (b >= g.blocks_x * g.blocks_y) return;
    wave_block_table_entry(g, b, tiles_x, tiles_y, row_lo, row_hi, done, row_words, win, table, ranges, live_cells);
}

__device__ inline bool block_takes(const uint4* tab, uint32_t b, uint32_t key, bool keyed) {
    const uint4 t = tab[b];
    return t.z != 0u && (!keyed || (key >= t.x && key < t.y));
}

// blocks touched by the band-clipped tile rectangle: [bx0, bx1) x [by0, by1); false if none
__device__ inline bool block_rect(const BlockGrid& g, uint32_t rx, uint32_t ry, uint32_t row_lo, uint32_t row_hi, uint32_t& bx0,
                                  uint32_t& bx1, uint32_t& by0, uint32_t& by1) {
    const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = max(ry & 0xFFFFu, row_lo), y1 = min(ry >> 16, row_hi);
    if (x0 >= x1 || y0 >= y1) return false;
    bx0 = x0 >> g.bsx;
    bx1 = ((x1 - 1u) >> g.bsx) + 1u;
    by0 = (y0 - row_lo) >> g.bsy;  // (block rows count from the first row this viewer composites)
    by1 = ((y1 - 1u - row_lo) >> g.bsy) + 1u;
    return true;
}

__global__ __launch_bounds__(kBinThreads) void k_block_counts(const uint32_t* __restrict__ d_n_vis, uint32_t j0, uint32_t j1,
                                                               const uint32_t* __restrict__ sorted_idx,
                                                               const float4* __restrict__ rec_a,
                                                               const uint32_t* __restrict__ sorted_keys, uint4* __restrict__ brec,
                                                               uint32_t* __restrict__ cnt, uint32_t* __restrict__ block_sums,
                                                               uint32_t row_lo, uint32_t row_hi,
                                                               const uint32_t* __restrict__ d_done_count, uint32_t owned_tiles,
                                                               BlockGrid g, const uint4* __restrict__ table, int keyed,
                                                               uint32_t* __restrict__ order_buf, uint32_t order_tiles, uint32_t* __restrict__ walk_max_out) {
    __shared__ uint32_t red[4];
    __shared__ uint4 tab[1024];
    // the launch's one extra workgroup (the first, so that it starts at once): the block compositor's dispatch order, made while the
    // others count (tile_order_job, gsx_internal.h — ~8 us of one workgroup's latency chain, hidden here; as a launch of its own: 11)
    const uint32_t extra = order_buf ? 1u : 0u, workers = gridDim.x - extra, worker = blockIdx.x - extra;
    if (extra && blockIdx.x == 0u) {
        tile_order_job<kBinThreads>(order_buf, order_tiles, reinterpret_cast<uint32_t*>(tab), walk_max_out);
        return;
    }
    const uint32_t n_vis = min(*d_n_vis, j1);
    const uint32_t chunks = n_vis > j0 ? (n_vis - j0 + kBinThreads - 1) / kBinThreads : 0u;
    const bool all_done = d_done_count && *d_done_count >= owned_tiles;
    if (!all_done && worker < chunks) {
        for (uint32_t b = threadIdx.x; b < g.blocks_x * g.blocks_y; b += kBinThreads) tab[b] = table[b];
        __syncthreads();
    }
    for (uint32_t chunk = worker; chunk < chunks; chunk += workers) {
        if (all_done) {
            if (threadIdx.x == 0) block_sums[chunk] = 0;
            continue;
        }
        const uint32_t j = j0 + chunk * kBinThreads + threadIdx.x;
        uint32_t c = 0;
        if (j < n_vis) {
            const uint32_t idx = sorted_idx[j];
            const float4 a = rec_a[idx];
            const uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
            const uint32_t key = sorted_keys[j];
            uint32_t bx0, bx1, by0, by1;
            if (block_rect(g, rx, ry, row_lo, row_hi, bx0, bx1, by0, by1))
                for (uint32_t by = by0; by < by1; ++by)
                    for (uint32_t bx = bx0; bx < bx1; ++bx) c += block_takes(tab, by * g.blocks_x + bx, key, keyed != 0) ? 1u : 0u;
            if (c)  // (only records that make an entry are ever looked up: later slabs mostly hit saturated blocks)
                brec[j - j0] = make_uint4(rx, ry, key, idx);
            cnt[j - j0] = c;
        }
        const uint32_t tot = block_reduce_sum(c, red);
        if (threadIdx.x == 0) block_sums[chunk] = tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBinThreads) void k_block_emit(uint32_t jbase, uint32_t j1, const uint4* __restrict__ brec,
                                                             const uint32_t* __restrict__ cnt,
                                                             const uint32_t* __restrict__ block_offs, uint2* __restrict__ pairs,
                                                             uint32_t row_lo, uint32_t row_hi, const uint32_t* __restrict__ d_n_vis,
                                                             const uint32_t* __restrict__ d_entries, uint32_t capacity,
                                                             const uint32_t* __restrict__ d_cut, BlockGrid g,
                                                             const uint4* __restrict__ table, int keyed) {
    __shared__ uint32_t wsum[4];
    __shared__ uint4 tab[1024];
    if (*d_entries == 0) return;
    const uint32_t n_vis = min(min(j1, *d_n_vis), *d_cut);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunks = n_vis > jbase ? (n_vis - jbase + kBinThreads - 1) / kBinThreads : 0u;
    if (blockIdx.x < chunks)
        for (uint32_t b = tid; b < g.blocks_x * g.blocks_y; b += kBinThreads) tab[b] = table[b];
    __syncthreads();
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t j = jbase + chunk * kBinThreads + tid;
        const uint32_t mine = j < n_vis ? cnt[j - jbase] : 0u;
        uint32_t x = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= (uint32_t)o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t o = block_offs[chunk] + x - mine;
        for (uint32_t w = 0; w < wave; ++w) o += wsum[w];
        if (mine) {
            const uint4 r = brec[j - jbase];
            uint32_t bx0, bx1, by0, by1;
            if (block_rect(g, r.x, r.y, row_lo, row_hi, bx0, bx1, by0, by1))
                for (uint32_t by = by0; by < by1; ++by)
                    for (uint32_t bx = bx0; bx < bx1; ++bx) {
                        const uint32_t b = by * g.blocks_x + bx;
                        if (!block_takes(tab, b, r.z, keyed != 0)) continue;
                        if (o < capacity) pairs[o] = make_uint2(b, j - jbase);
                        ++o;
                    }
        }
        __syncthreads();
    }
}

// ---- k_block_counts + k_scan_block_sums + k_block_emit + the block sort's histogram as ONE launch (round 6) ----
// Persistent 1024-lane workgroups take 2048-record tiles of the slab in ticket order.  A tile: every lane gathers two records
// (both loads in flight), counts the blocks each touches, the tile's total is published as one 64-bit {epoch, flag, count} word and
// wave 0 looks back over the tiles before it (kernels_sort.hip's protocol; 64 predecessors per round trip) — so the entries' slots
// are known without a scan kernel in between — then the lanes emit their (block, slab position) entries and count them into the
// workgroup's LDS histogram of the block sort's digits, flushed once per workgroup (a few hundred workgroups at most: when every one
// of 6144 emit workgroups flushed 256 bins, round 4, the flushes cost more than the histogram launch they replaced).
// The slab's cut (entries that do not fit the pair buffers: SlabStats::slab_cut) falls out of the same prefix: the tile in which the
// capacity is crossed emits up to the last record that fits and says where it stopped; tiles behind it emit nothing.
constexpr int kFuseThreads = 1024;
constexpr int kFuseWaves = kFuseThreads / 64;
constexpr uint32_t kFuseTile = 2u * kFuseThreads;   // records per tile: two per lane (eight on slabs of kFuseBigSlab records and more)
constexpr uint32_t kFuseBigSlab = 1u << 20;
constexpr uint32_t kBigRect = 32;   // blocks: larger rectangles are walked by the whole wave (block_bin_tiles)
static std::atomic<uint32_t> g_big_rect{kBigRect};
void block_bin_set_big_rect(uint32_t blocks) { g_big_rect.store(blocks ? blocks : kBigRect); }
static std::atomic<uint32_t> g_big_slab{kFuseBigSlab};
void block_bin_set_big_slab(uint32_t records) { g_big_slab.store(records ? records : kFuseBigSlab); }
constexpr uint32_t kFuseGrid = 256;
typedef unsigned long long u64b;

size_t bin_workspace_words(uint64_t n_records) { return 8 + 4 * (size_t)((n_records + kFuseTile - 1) / kFuseTile + 1); }  // two 64-bit status words per tile

// The tile loop of k_block_bin with R records per lane (a tile = R x 1024 records in depth order, lane t holding positions
// r * 1024 + t of it: coalesced loads, R wave scans).  A workgroup's tile costs a chain of memory round trips (ticket, index, rectangle,
// look-back) whatever R is, and 1024-lane workgroups sit one to a CU: big slabs take R = 8 (a quarter of the trips), small ones R = 2
// (enough tiles for every CU).
struct BinShared {
    uint4 tab[1024];
    uint32_t hist[2][256];
    uint32_t w[8][kFuseWaves], f[8][kFuseWaves];
    uint32_t tile, before, total, cut, shade_before, shade_total;
    uint32_t live[8];
};

template <int R>
__device__ __forceinline__ void block_bin_tiles(BinShared& sh, uint32_t n, uint32_t n_vis, uint32_t j0, const uint32_t* __restrict__ sorted_idx,
                                                const float4* __restrict__ rec_a, const uint32_t* __restrict__ sorted_keys,
                                                uint4* __restrict__ brec, uint2* __restrict__ pairs, SlabStats* __restrict__ stats,
                                                uint32_t capacity, uint32_t row_lo, uint32_t row_hi, uint32_t slab_index, const BlockGrid& g,
                                                bool keyed, uint32_t* __restrict__ ticket, u64b* __restrict__ status, uint32_t epoch, int passes,
                                                int dbits, const uint32_t* __restrict__ rect8, uint2* __restrict__ shade_pairs,
                                                const uint8_t* __restrict__ sorted_code, bool coarse, uint32_t big_rect) {
    constexpr uint32_t kTile = (uint32_t)R * kFuseThreads;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_tiles = (n + kTile - 1u) / kTile;
    const uint32_t dmask = (1u << dbits) - 1u;
    for (;;) {
        if (tid == 0) sh.tile = atomicAdd(&ticket[0], 1u);
        __syncthreads();
        const uint32_t tile = sh.tile;
        if (tile >= n_tiles) break;
        uint32_t idx[R], key[R], rx[R], ry[R], c[R];
        bool look[R];
        const uint32_t p0 = tile * kTile + tid;   // the lane's records: slab positions p0 + r * 1024
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pos = p0 + (uint32_t)r * kFuseThreads;
            idx[r] = pos < n ? sorted_idx[j0 + pos] : 0u;
            key[r] = pos < n ? sorted_keys[j0 + pos] : 0u;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pos = p0 + (uint32_t)r * kFuseThreads;
            look[r] = pos < n;
            if (coarse && look[r]) look[r] = coarse_hit(sorted_code[j0 + pos], sh.live);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {   // (slab shading: the records are not shaded yet — the packed rectangle of the geometry-only projection)
            rx[r] = ry[r] = 0;
            if (look[r]) rec_rect(rec_a, rect8, idx[r], rx[r], ry[r]);
        }
        // A record asks every block its rectangle touches.  Rectangles of up to kBigRect blocks are walked by their own lane; larger ones
        // (a splat across half the screen touches hundreds of the 1024 blocks of a long-walk scene) are taken one at a time by the whole
        // wave, 64 blocks a step — a lane walking 1024 blocks alone held its wave for 2 x 1024 steps: k_block_bin took 200-2000 us a frame
        // on the surfaces scene (round 6), a third of the frame.
        bool big[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            c[r] = 0;
            uint32_t bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
            const bool has = look[r] && block_rect(g, rx[r], ry[r], row_lo, row_hi, bx0, bx1, by0, by1);
            big[r] = has && (bx1 - bx0) * (by1 - by0) > big_rect;
            if (has && !big[r])
                for (uint32_t by = by0; by < by1; ++by)
                    for (uint32_t bx = bx0; bx < bx1; ++bx) c[r] += block_takes(sh.tab, by * g.blocks_x + bx, key[r], keyed) ? 1u : 0u;
            unsigned long long bm = __ballot(big[r]);
            while (bm) {
                const int L = __ffsll((long long)bm) - 1;
                bm &= bm - 1ull;
                const uint32_t rxL = (uint32_t)__shfl((int)rx[r], L, 64), ryL = (uint32_t)__shfl((int)ry[r], L, 64), keyL = (uint32_t)__shfl((int)key[r], L, 64);
                uint32_t a0, a1, b0, b1;
                (void)block_rect(g, rxL, ryL, row_lo, row_hi, a0, a1, b0, b1);
                const uint32_t w = a1 - a0, total = w * (b1 - b0);
                uint32_t cnt = 0;
                for (uint32_t i0 = 0; i0 < total; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    const bool t = i < total && block_takes(sh.tab, (b0 + i / w) * g.blocks_x + a0 + i % w, keyL, keyed);
                    cnt += (uint32_t)__popcll(__ballot(t));
                }
                if (lane == (uint32_t)L) c[r] = cnt;
            }
            if (c[r]) brec[p0 + (uint32_t)r * kFuseThreads] = make_uint4(rx[r], ry[r], key[r], idx[r]);   // (only records that make an entry are ever looked up)
        }
        // inclusive wave scans of the R stripes; slab shading: the records some block takes, counted by ballots
        uint32_t x[R];
        unsigned long long fb[R];
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = c[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t y = __shfl_up(x[r], o, 64);
                if (lane >= (uint32_t)o) x[r] += y;
            }
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fb[r] = __ballot(c[r] != 0u);
            if (lane == 63) sh.w[r][wave] = x[r];
            if (lane == 0) sh.f[r][wave] = (uint32_t)__popcll(fb[r]);
        }
        __syncthreads();
        // the R x 16 wave totals in tile order (stripe by stripe): wave 0 turns the entry counts into their exclusive prefix and asks the
        // tiles in front for the tile's first slot; wave 1 does the same for the shading list (a second status word per tile)
        if (wave < 2u && (wave == 0u || shade_pairs)) {
            uint32_t* cells = wave == 0u ? &sh.w[0][0] : &sh.f[0][0];
            constexpr int kCells = R * kFuseWaves;
            uint32_t carry = 0;
#pragma unroll
            for (int h = 0; h < (kCells + 63) / 64; ++h) {
                const uint32_t i = (uint32_t)h * 64u + lane;
                const uint32_t v = i < (uint32_t)kCells ? cells[i] : 0u;
                uint32_t inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = __shfl_up(inc, o, 64);
                    if (lane >= (uint32_t)o) inc += y;
                }
                if (i < (uint32_t)kCells) cells[i] = carry + inc - v;
                carry += __shfl(inc, 63, 64);
            }
            const uint32_t excl = tile_scan_publish(status + wave, 2u, tile, epoch, lane, carry);
            if (lane == 0) {
                if (wave == 0u) {
                    sh.before = excl;
                    sh.total = carry;
                    sh.cut = 0xFFFFFFFFu;
                } else {
                    sh.shade_before = excl;
                    sh.shade_total = carry;
                }
            }
        }
        __syncthreads();
        const uint32_t total = sh.total;
        uint32_t e[R];
#pragma unroll
        for (int r = 0; r < R; ++r) e[r] = sh.w[r][wave] + x[r] - c[r];   // first slot of each record's entries inside the tile
        const uint32_t before = sh.before;
        if (shade_pairs) {   // (every record some block takes, whatever the cut: the tile compositor's pair-free tail blends them too)
            const uint32_t sb = sh.shade_before;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (c[r]) shade_pairs[sb + sh.f[r][wave] + (uint32_t)__popcll(fb[r] & lt)] = make_uint2(key[r], idx[r]);
            if (tile == n_tiles - 1u && tid == 0) {
                stats->n_slab_shade = sb + sh.shade_total;
                stats->n_shaded_total += sb + sh.shade_total;
            }
        }
        // the pair buffers hold `capacity` entries: the first record whose last entry would not fit is the slab's cut
        const bool crossing = before <= capacity && before + total > capacity;
        if (crossing) {
            const uint32_t room = capacity - before;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (c[r] && e[r] + c[r] > room) atomicMin(&sh.cut, p0 + (uint32_t)r * kFuseThreads);
            __syncthreads();
        }
        const uint32_t cut = crossing ? sh.cut : (before > capacity ? 0u : 0xFFFFFFFFu);   // (before > capacity: an earlier tile crossed)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t pos = p0 + (uint32_t)r * kFuseThreads;
            const bool emit = c[r] && pos < cut;
            if (emit && !big[r]) {
                uint32_t o = before + e[r];
                uint32_t bx0, bx1, by0, by1;
                if (block_rect(g, rx[r], ry[r], row_lo, row_hi, bx0, bx1, by0, by1))
                    for (uint32_t by = by0; by < by1; ++by)
                        for (uint32_t bx = bx0; bx < bx1; ++bx) {
                            const uint32_t b = by * g.blocks_x + bx;
                            if (!block_takes(sh.tab, b, key[r], keyed)) continue;
                            pairs[o++] = make_uint2(b, pos);
                            atomicAdd(&sh.hist[0][b & dmask], 1u);
                            if (passes > 1) atomicAdd(&sh.hist[1][(b >> dbits) & dmask], 1u);
                        }
            }
            unsigned long long bm = __ballot(emit && big[r]);   // (the wave takes the large rectangles one at a time; a record's entries go to different blocks: their order among themselves is nobody's)
            while (bm) {
                const int L = __ffsll((long long)bm) - 1;
                bm &= bm - 1ull;
                const uint32_t rxL = (uint32_t)__shfl((int)rx[r], L, 64), ryL = (uint32_t)__shfl((int)ry[r], L, 64), keyL = (uint32_t)__shfl((int)key[r], L, 64);
                const uint32_t posL = (uint32_t)__shfl((int)pos, L, 64);
                uint32_t o = (uint32_t)__shfl((int)(before + e[r]), L, 64);
                uint32_t a0, a1, b0, b1;
                (void)block_rect(g, rxL, ryL, row_lo, row_hi, a0, a1, b0, b1);
                const uint32_t w = a1 - a0, total = w * (b1 - b0);
                for (uint32_t i0 = 0; i0 < total; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    const uint32_t b = (b0 + i / w) * g.blocks_x + a0 + i % w;
                    const bool t = i < total && block_takes(sh.tab, b, keyL, keyed);
                    const unsigned long long bal = __ballot(t);
                    if (t) {
                        pairs[o + (uint32_t)__popcll(bal & lt)] = make_uint2(b, posL);
                        atomicAdd(&sh.hist[0][b & dmask], 1u);
                        if (passes > 1) atomicAdd(&sh.hist[1][(b >> dbits) & dmask], 1u);
                    }
                    o += (uint32_t)__popcll(bal);
                }
            }
        }
        if (tid == 0) {
            if (crossing) {   // entries up to the cut: the prefix of the record at the cut
                stats->slab_cut = j0 + cut;
            }
            if (tile == n_tiles - 1u) {
                const uint32_t all = before + total;
                const bool over = all > capacity;
                stats->n_entries_total += all;
                stats->max_needed = max(stats->max_needed, all);
                stats->slabs_used = max(stats->slabs_used, slab_index + 1u);
                if (over) {
                    stats->overflow = 1;
                    stats->overflow_events += 1;
                    stats->max_needed_ever = max(stats->max_needed_ever, all);
                } else {
                    stats->n_entries = all;
                    stats->slab_cut = max(n_vis, j0);
                }
            }
        }
        if (crossing) {   // n_entries = entries in front of the cut: every lane knows its records' slots; the record AT the cut says it
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (c[r] && p0 + (uint32_t)r * kFuseThreads == cut) stats->n_entries = before + e[r];
        }
        __syncthreads();   // sh.tile, sh.w / sh.f, sh.before, sh.cut are reused by the next tile
    }
}

__global__ __launch_bounds__(kFuseThreads) void k_block_bin(const uint32_t* __restrict__ d_n_vis, uint32_t j0, uint32_t j1,
                                                             const uint32_t* __restrict__ sorted_idx, const float4* __restrict__ rec_a,
                                                             const uint32_t* __restrict__ sorted_keys, uint4* __restrict__ brec,
                                                             uint2* __restrict__ pairs, SlabStats* __restrict__ stats, uint32_t capacity,
                                                             uint32_t row_lo, uint32_t row_hi, const uint32_t* __restrict__ d_done_count,
                                                             uint32_t owned_tiles, uint32_t slab_index, BlockGrid g,
                                                             const uint4* __restrict__ table, int keyed, uint32_t* __restrict__ order_buf,
                                                             uint32_t order_tiles, uint32_t* __restrict__ walk_max_out,
                                                             uint32_t* __restrict__ ticket, u64b* __restrict__ status, uint32_t epoch,
                                                             uint32_t* __restrict__ ghist, int passes, int dbits,
                                                             const uint32_t* __restrict__ rect8, uint2* __restrict__ shade_pairs,
                                                             const uint8_t* __restrict__ sorted_code, const uint32_t* __restrict__ live_cells, uint32_t big_slab, uint32_t big_rect) {
    __shared__ BinShared sh;
    const uint32_t extra = order_buf ? 1u : 0u, workers = gridDim.x - extra, worker = blockIdx.x - extra;
    if (extra && blockIdx.x == 0u) {
        tile_order_job<kFuseThreads>(order_buf, order_tiles, reinterpret_cast<uint32_t*>(sh.tab), walk_max_out);
        return;
    }
    const uint32_t tid = threadIdx.x;
    const uint32_t n_vis = min(*d_n_vis, j1);
    const uint32_t n = n_vis > j0 ? n_vis - j0 : 0u;
    const bool big = n >= big_slab;
    const uint32_t tile_records = (big ? 8u : 2u) * kFuseThreads;
    const uint32_t n_tiles = (n + tile_records - 1u) / tile_records;
    const bool all_done = d_done_count && *d_done_count >= owned_tiles;   // every tile this rank composites is saturated
    if (n_tiles == 0u || all_done) {
        if (worker == 0u && tid == 0u) {
            stats->n_entries = 0u;
            stats->slab_cut = max(n_vis, j0);
            if (shade_pairs) stats->n_slab_shade = 0u;
        }
        return;
    }
    const uint32_t participants = min(workers, n_tiles);
    if (worker >= participants) return;
    const uint32_t n_blocks = g.blocks_x * g.blocks_y;
    for (uint32_t b = tid; b < n_blocks; b += kFuseThreads) sh.tab[b] = table[b];
    if (tid < 512u) (&sh.hist[0][0])[tid] = 0u;
    // later slabs: the coarse cells that still hold an open tile; a record none of whose cells does is refused before its rectangle —
    // a 64-byte sector by depth order — is looked up (its cells arrived in depth order with the sort: one coalesced byte)
    if (live_cells && tid < 8u) sh.live[tid] = live_cells[tid];
    __syncthreads();
    if (big)
        block_bin_tiles<8>(sh, n, n_vis, j0, sorted_idx, rec_a, sorted_keys, brec, pairs, stats, capacity, row_lo, row_hi, slab_index, g, keyed != 0,
                           ticket, status, epoch, passes, dbits, rect8, shade_pairs, sorted_code, live_cells != nullptr, big_rect);
    else
        block_bin_tiles<2>(sh, n, n_vis, j0, sorted_idx, rec_a, sorted_keys, brec, pairs, stats, capacity, row_lo, row_hi, slab_index, g, keyed != 0,
                           ticket, status, epoch, passes, dbits, rect8, shade_pairs, sorted_code, live_cells != nullptr, big_rect);
    for (uint32_t i = tid; i < 512u; i += kFuseThreads) {
        const uint32_t v = (&sh.hist[0][0])[i];
        if (v && (int)(i >> 8) < passes) atomicAdd(&ghist[i], v);
    }
    if (tid == 0) {   // the last workgroup to leave re-arms the ticket
        const uint32_t fin = atomicAdd(&ticket[1], 1u);
        if (fin == participants - 1u) {
            ticket[1] = 0;
            __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

hipError_t launch_block_bin_fused(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                                  const Records& rec, const uint32_t* sorted_keys, uint4* brec, SlabStats* stats, uint32_t capacity,
                                  uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words, const uint32_t* d_done_count,
                                  uint32_t owned_tiles, uint32_t slab_index, const uint2* window, uint32_t tiles_x, uint32_t tiles_y,
                                  uint32_t bsx, uint32_t bsy, uint4* table, uint2* pairs, uint2* ranges, const ZeroJob& zero, bool table_ready,
                                  uint32_t* bin_ws, uint32_t* sort_ghist, int block_bits, uint2* shade_pairs, const uint8_t* sorted_code) {
    const BlockGrid g = block_grid(bsx, bsy, tiles_x, row_lo, row_hi);
    // later slabs with the records' coarse cells in depth order: the table kernel also says which coarse cells still hold an open tile
    // (zero.a is NOT this slab's word: the frame's zero job ran with the first slab, whose word nobody reads)
    // (and not the slab whose table kernel carries the frame's zero job: that job zeroes these very words)
    uint32_t* live = (sorted_code && done && !table_ready && zero.na == 0u && zero.nb == 0u) ? &stats->live_cells[std::min<uint32_t>(slab_index, 7u)][0] : nullptr;
    if (!table_ready)
        GSX_LAUNCH(k_block_table, dim3((g.blocks_x * g.blocks_y + 3u) / 4u), dim3(256), 0, s, g, tiles_x, tiles_y, row_lo, row_hi, done,
                   row_words, window, table, ranges, zero.a, zero.na, zero.b, zero.nb, live, zero.copy_src, zero.copy_dst, zero.n_copy);
    const uint64_t tiles = ((uint64_t)(j1 > j0 ? j1 - j0 : 0) + kFuseTile - 1) / kFuseTile;
    const uint32_t nb = (uint32_t)std::min<uint64_t>(tiles, kFuseGrid);
    const int passes = (block_bits + 7) / 8, dbits = (block_bits + passes - 1) / passes;
    if (nb)
        GSX_LAUNCH(k_block_bin, dim3(nb + (zero.order_buf ? 1u : 0u)), dim3(kFuseThreads), 0, s, d_n_vis, j0, j1, sorted_idx, rec.a, sorted_keys, brec, pairs,
                   stats, capacity, row_lo, row_hi, d_done_count, owned_tiles, slab_index, g, table, window ? 1 : 0, zero.order_buf, zero.order_tiles,
                   zero.order_buf ? &stats->walk_max : nullptr, bin_ws, reinterpret_cast<u64b*>(bin_ws + 8), next_sort_epoch(), sort_ghist, passes, dbits,
                   shade_pairs ? rec.rect8 : nullptr, shade_pairs, live ? sorted_code : nullptr, live, g_big_slab.load(), g_big_rect.load());
    return hipGetLastError();
}

hipError_t launch_block_bin(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                            const Records& rec, const uint32_t* sorted_keys, uint4* brec, uint32_t* cnt, uint32_t* block_sums,
                            SlabStats* stats, uint32_t capacity, uint32_t row_lo, uint32_t row_hi, const uint32_t* done,
                            uint32_t row_words, const uint32_t* d_done_count, uint32_t owned_tiles, uint32_t slab_index,
                            const uint2* window, uint32_t tiles_x, uint32_t tiles_y, uint32_t bsx, uint32_t bsy, uint4* table,
                            uint2* pairs, uint2* ranges, const ZeroJob& zero, bool table_ready) {
    const uint32_t nb = std::min<uint32_t>((uint32_t)scan_blocks(j1 > j0 ? j1 - j0 : 0), kBinGrid);
    const BlockGrid g = block_grid(bsx, bsy, tiles_x, row_lo, row_hi);
    if (!table_ready)  // (the repair round of a speculated frame: k_spec_verify_fused has built the table and zeroed the ranges)
        GSX_LAUNCH(k_block_table, dim3((g.blocks_x * g.blocks_y + 3u) / 4u), dim3(256), 0, s, g, tiles_x, tiles_y, row_lo, row_hi, done,
                   row_words, window, table, ranges, zero.a, zero.na, zero.b, zero.nb, (uint32_t*)nullptr, zero.copy_src, zero.copy_dst, zero.n_copy);
    if (nb)
        GSX_LAUNCH(k_block_counts, dim3(nb + (zero.order_buf ? 1u : 0u)), dim3(kBinThreads), 0, s, d_n_vis, j0, j1, sorted_idx, rec.a, sorted_keys,
                   brec, cnt, block_sums, row_lo, row_hi, d_done_count, owned_tiles, g, table, window ? 1 : 0, zero.order_buf, zero.order_tiles,
                   zero.order_buf ? &stats->walk_max : nullptr);
    GSX_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, j0, j1, d_n_vis, stats, capacity, d_done_count,
                       owned_tiles, slab_index);
    if (nb)
        GSX_LAUNCH(k_block_emit, dim3(nb), dim3(kBinThreads), 0, s, j0, j1, brec, cnt, block_sums, pairs, row_lo, row_hi,
                           d_n_vis, &stats->n_entries, capacity, &stats->slab_cut, g, table, window ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_tile_counts(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* d_n_vis, const uint32_t* sorted_idx,
                              const Records& rec, uint2* srect, uint32_t* cnt, uint32_t* block_sums, SlabStats* stats,
                              uint32_t capacity, uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words,
                              const uint32_t* d_done_count, uint32_t owned_tiles, uint32_t slab_index,
                              const uint2* window, const uint32_t* sorted_keys, uint32_t tiles_x, const WindowPyramid* min_ends) {
    const uint32_t nb = std::min<uint32_t>((uint32_t)scan_blocks(j1 > j0 ? j1 - j0 : 0), kBinGrid);
    TileWindow tw{window, sorted_keys, tiles_x, min_ends ? *min_ends : WindowPyramid{}};
    if (nb)
        GSX_LAUNCH(k_tile_counts, dim3(nb), dim3(kBinThreads), 0, s, d_n_vis, j0, j1, sorted_idx, rec.a, srect, cnt,
                           block_sums, row_lo, row_hi, done, row_words, d_done_count, owned_tiles, tw);
    GSX_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, j0, j1, d_n_vis, stats, capacity, d_done_count,
                       owned_tiles, slab_index);
    return hipGetLastError();
}

hipError_t launch_tile_emit(hipStream_t s, uint32_t j0, uint32_t j1, const uint32_t* sorted_idx, const uint2* srect,
                            const uint32_t* cnt, const uint32_t* block_sums, uint32_t tiles_x, uint2* tpairs,
                            uint32_t row_lo, uint32_t row_hi, const uint32_t* done, uint32_t row_words,
                            const uint32_t* d_n_vis, const uint32_t* d_entries, uint32_t capacity,
                            const uint2* window, const uint32_t* sorted_keys, const uint32_t* d_cut) {
    const uint32_t nb = std::min<uint32_t>((uint32_t)scan_blocks(j1 > j0 ? j1 - j0 : 0), kBinGrid);
    if (!nb) return hipSuccess;
    TileWindow tw{window, sorted_keys, tiles_x, WindowPyramid{}};
    GSX_LAUNCH(k_tile_emit, dim3(nb), dim3(kBinThreads), 0, s, j0, j1, sorted_idx, srect, cnt, block_sums,
                       tiles_x, tpairs, row_lo, row_hi, done, row_words, d_n_vis, d_entries, capacity, tw, d_cut);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(hipStream_t s, uint32_t capacity, const uint32_t* d_n, const uint32_t* tkey_sorted,
                              uint32_t table_tiles, uint2* ranges, bool ranges_clean) {
    if (!ranges_clean) {
        // the WHOLE allocation, not this frame's tile count: "clean" is a statement about every entry a later, larger
        // viewport may read (a table allocated with slack and zeroed up to a smaller frame's tile count handed stale
        // ranges to the compositor after a resize — found by the API fuzz test)
        hipError_t e = gsx::op::MemsetAsync(ranges, 0, sizeof(uint2) * (size_t)table_tiles, s);
        if (e != hipSuccess) return e;
    }
    if (capacity == 0) return hipSuccess;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)capacity + 255) / 256, 4096);
    GSX_LAUNCH(k_tile_ranges, dim3(grid), dim3(256), 0, s, d_n, tkey_sorted, ranges);
    return hipGetLastError();
}

}  // namespace gsx
