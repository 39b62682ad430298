// kernels_spec.hip — temporal occlusion speculation on one GPU (gfx950): the device-side policy kernels.
//
// No reference counterpart (the reference sorts and draws every surviving Gaussian, src/tab/scene.rs:865-869,
// 2302-2314).  An opaque scene hides most visible splats, and which ones barely changes between two frames.  Every
// 16x16 tile therefore carries a depth-key window [0, hi): hi = (1 + margin) x the deepest depth at which the tile's
// (2 radius + 1)^2 neighbourhood saturated in the model's previous frame, unbounded where a neighbour stayed open.  Only
// records some tile admits enter the depth sort and the binning (kernels_admit.hip, kernels_bin.hip).  After
// compositing, k_spec_verify finds the tiles that had a bounded window and are still open; a second round gives
// exactly those tiles the records they were refused, [hi, inf), composited behind — so every tile always blends a
// gap-free depth prefix and the frame is bit-identical to the unspeculated one.  Everything stays on the device: no
// host round trip, the second round's kernels fall through when nothing needs repair.
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

constexpr uint32_t kKeyAll = 0xFFFFFFFFu;

// replaces hipMemsetAsync for small, oddly sized ranges (the runtime splits those into several fill kernels)
__global__ __launch_bounds__(256) void k_zero_words(uint32_t* __restrict__ a, uint32_t na, uint32_t* __restrict__ b, uint32_t nb) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < na) a[i] = 0u;
    if (i < nb) b[i] = 0u;
}

hipError_t launch_zero_words(hipStream_t s, uint32_t* a, uint32_t na, uint32_t* b, uint32_t nb) {
    const uint32_t n = na > nb ? na : nb;
    if (n) GSX_LAUNCH(k_zero_words, dim3((n + 255) / 256), dim3(256), 0, s, a, na, b, nb);
    return hipGetLastError();
}

// Occupies a stream for `ticks` of the 100 MHz wall clock (one wave, no memory traffic): gsx_api.cpp uses it once per lane to
// find out whether two HIP streams share a hardware queue.
__global__ __launch_bounds__(64) void k_spin(unsigned long long ticks, uint32_t* __restrict__ sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long now = t0;
    while (now - t0 < ticks) now = wall_clock64();
    if (sink && threadIdx.x == 0 && now == 0ull) *sink = 1u;  // never true: keeps the loop from being optimised away
}

hipError_t launch_spin(hipStream_t s, uint32_t microseconds) {
    GSX_LAUNCH(k_spin, dim3(1), dim3(64), 0, s, 100ull * microseconds, static_cast<uint32_t*>(nullptr));
    return hipGetLastError();
}

// Debug aid (GSX_VALIDATE=1, gsx_frame.cpp): everything the compositor is about to dereference, checked before it runs.
// report[0] = first error code (1 range outside the sorted entries, 2 list index outside the records), [1] tile, [2..3] detail.
__global__ __launch_bounds__(256) void k_validate_tiles(const uint2* __restrict__ ranges, uint32_t n_tiles,
                                                         const uint32_t* __restrict__ list, const uint32_t* __restrict__ d_entries,
                                                         uint32_t capacity, uint32_t n_records, uint32_t* __restrict__ report) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t D = min(*d_entries, capacity);
    const uint2 r = ranges[t];
    if (r.x > r.y || r.y > D) {
        if (atomicCAS(&report[0], 0u, 1u) == 0u) { report[1] = t; report[2] = r.x; report[3] = r.y; report[4] = D; }
        return;
    }
    for (uint32_t e = r.x; e < r.y; ++e) {
        const uint32_t idx = list[e];
        if (idx >= n_records) {
            if (atomicCAS(&report[0], 0u, 2u) == 0u) { report[1] = t; report[2] = e; report[3] = idx; report[4] = n_records; }
            return;
        }
    }
}

hipError_t launch_validate_tiles(hipStream_t s, const uint2* ranges, uint32_t n_tiles, const uint32_t* list, const uint32_t* d_entries,
                                 uint32_t capacity, uint32_t n_records, uint32_t* report) {
    GSX_LAUNCH(k_validate_tiles, dim3((n_tiles + 255) / 256), dim3(256), 0, s, ranges, n_tiles, list, d_entries, capacity, n_records, report);
    return hipGetLastError();
}

// one workgroup builds every level (11 k words at 1080p): level 0 = the window ends, level l = 2x2 max of level l-1.
// min_ends (nullable): a second pyramid, the MIN of the window ends — "every tile under this rectangle takes this key"
// for windows that start at 0 (the binning's fast path, kernels_bin.hip).
// Levels that fit kPyrLds words are kept in LDS as well: the level above is built from there instead of from global words this
// workgroup has just written (a round trip through L2 per level, 7 of them at 1080p: 12 us for 11 k words).  Level 1 comes
// straight from the windows, so that it does not wait for level 0 either.
constexpr uint32_t kPyrLds = 2304;  // 1920x1080: level 1 = 60 x 34 = 2040 cells
struct PyramidLds {
    unsigned long long cells;
    uint32_t a[2][kPyrLds], m[2][kPyrLds];
};
// every thread of ONE workgroup (NT threads); `window` may have been written by this workgroup (barrier before the call)
template <uint32_t NT>
__device__ inline void build_window_pyramid(const uint2* __restrict__ window, const WindowPyramid& p, uint32_t* __restrict__ data,
                                            uint32_t* __restrict__ min_ends, PyramidLds& lds) {
    if (threadIdx.x == 0) lds.cells = 0ull;
    __syncthreads();
    const uint32_t mos = p.min_of_starts;
    auto leaf = [&](const uint2 w) -> uint32_t { return mos ? (w.y > w.x ? w.x : 0xFFFFFFFFu) : w.y; };
    auto leaf_min = [](const uint2 w) -> uint32_t { return w.x == 0u ? w.y : 0u; };  // a window that does not start at 0 promises nothing
    auto join = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) -> uint32_t { return mos ? min(min(a, b), min(c, d)) : max(max(a, b), max(c, d)); };
    unsigned long long cells = 0ull;
    // (kPyrBatch cells per trip, their loads issued together: one window per trip made level 0 a chain of 32 memory round trips per thread at
    //  3840x2160 — 30 us of one workgroup, four times a cfg5 frame)
    constexpr uint32_t kPyrBatch = 8;
    const uint32_t n0 = p.wx[0] * p.wy[0], px0 = p.wx[0];
    for (uint32_t base = threadIdx.x; base < n0; base += NT * kPyrBatch) {
        uint2 w[kPyrBatch];
#pragma unroll
        for (uint32_t u = 0; u < kPyrBatch; ++u) {
            const uint32_t i = base + u * NT;
            w[u] = i < n0 ? window[i] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (uint32_t u = 0; u < kPyrBatch; ++u) {
            const uint32_t i = base + u * NT;
            if (i >= n0) continue;
            data[i] = leaf(w[u]);
            if (min_ends) min_ends[i] = leaf_min(w[u]);
            if (mos && w[u].y > w[u].x) cells |= 1ull << ((((i / px0) >> p.cell_sy) << 3) | ((i % px0) >> p.cell_sx));
        }
    }
    if (mos) {
        if (cells) atomicOr(&lds.cells, cells);
        __syncthreads();
        if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(data + p.cells_off) = lds.cells;
    }
    uint32_t l = 1, cur = 0;
    bool prev_lds = false;
    if (p.levels > 1) {  // level 1 from the windows themselves (it does not wait for level 0), four cells per trip
        const uint32_t wx = p.wx[1], px = p.wx[0], py = p.wy[0], n1 = wx * p.wy[1];
        const bool keep1 = n1 <= kPyrLds;
        constexpr uint32_t kB1 = 4;
        for (uint32_t base = threadIdx.x; base < n1; base += NT * kB1) {
            uint2 a[kB1], b[kB1], c[kB1], d[kB1];
#pragma unroll
            for (uint32_t u = 0; u < kB1; ++u) {
                const uint32_t i = min(base + u * NT, n1 - 1u);
                const uint32_t x = 2u * (i % wx), y = 2u * (i / wx), x1 = min(x + 1u, px - 1u), y1 = min(y + 1u, py - 1u);
                a[u] = window[y * px + x];
                b[u] = window[y * px + x1];
                c[u] = window[y1 * px + x];
                d[u] = window[y1 * px + x1];
            }
#pragma unroll
            for (uint32_t u = 0; u < kB1; ++u) {
                const uint32_t i = base + u * NT;
                if (i >= n1) continue;
                const uint32_t v = join(leaf(a[u]), leaf(b[u]), leaf(c[u]), leaf(d[u]));
                data[p.off[1] + i] = v;
                if (keep1) lds.a[0][i] = v;
                if (min_ends) {
                    const uint32_t mv = min(min(leaf_min(a[u]), leaf_min(b[u])), min(leaf_min(c[u]), leaf_min(d[u])));
                    min_ends[p.off[1] + i] = mv;
                    if (keep1) lds.m[0][i] = mv;
                }
            }
        }
        prev_lds = keep1;
        l = 2;
    }
    for (; l < p.levels; ++l) {
        __syncthreads();
        const uint32_t wx = p.wx[l], wy = p.wy[l], px = p.wx[l - 1], py = p.wy[l - 1];
        const uint32_t* src = prev_lds ? lds.a[cur] : data + p.off[l - 1];
        const uint32_t* msrc = prev_lds ? lds.m[cur] : (min_ends ? min_ends + p.off[l - 1] : nullptr);
        uint32_t* dst = data + p.off[l];
        const bool keep = wx * wy <= kPyrLds;
        const uint32_t wbuf = prev_lds ? cur ^ 1u : 0u;  // (never the buffer this level is read from)
        for (uint32_t i = threadIdx.x; i < wx * wy; i += NT) {
            const uint32_t x = 2u * (i % wx), y = 2u * (i / wx), x1 = min(x + 1u, px - 1u), y1 = min(y + 1u, py - 1u);
            const uint32_t v = join(src[y * px + x], src[y * px + x1], src[y1 * px + x], src[y1 * px + x1]);
            dst[i] = v;
            if (keep) lds.a[wbuf][i] = v;
            if (min_ends) {
                const uint32_t mv = min(min(msrc[y * px + x], msrc[y * px + x1]), min(msrc[y1 * px + x], msrc[y1 * px + x1]));
                min_ends[p.off[l] + i] = mv;
                if (keep) lds.m[wbuf][i] = mv;
            }
        }
        if (keep) cur = wbuf;
        prev_lds = keep;
    }
}

__global__ __launch_bounds__(1024) void k_window_pyramid(const uint2* __restrict__ window, WindowPyramid p, uint32_t* __restrict__ data,
                                                         const uint32_t* __restrict__ d_skip, uint32_t* __restrict__ min_ends) {
    if (d_skip && *d_skip == 0) return;  // repair round with nothing to repair
    __shared__ PyramidLds lds;
    build_window_pyramid<1024>(window, p, data, min_ends, lds);
}

// Verification of a speculated round, ONE workgroup (8 k tiles at 1080p: eight a thread):
//   need[t] = bounded window && still open  ->  win2[t] = [hi, inf) for those tiles, [0, 0) for the rest; *d_need = count;
//   the verdict to pinned host memory (host_verify);
//   and, when something needs repair, what the repair round's first kernels would otherwise be launched for: the min-pyramid
//   of the repair windows' starts (the conservative admission test of k_admit_count) and — block lists — the repair slab's block
//   table {min window start, max window end, live} with its ranges zeroed (k_block_table).  Two launches less per frame, and
//   nothing at all behind the count when nothing needs repair.
struct VerifyTables {
    WindowPyramid pyr2;          // of win2's starts (min_of_starts = 1); data == nullptr: none
    uint32_t* pyr2_data;
    BlockGrid grid;              // block lists of the repair slab; table == nullptr: none (per-tile lists)
    uint4* table;
    uint2* ranges;
};
__global__ __launch_bounds__(1024) void k_spec_verify_fused(const uint2* __restrict__ win1, const uint32_t* __restrict__ done,
                                                             uint32_t row_words, uint32_t tiles_x, uint32_t n_tiles,
                                                             uint2* __restrict__ win2, uint32_t* __restrict__ need_bits,
                                                             uint32_t* __restrict__ d_need, uint32_t band_lo, uint32_t band_hi,
                                                             unsigned long long* __restrict__ host_verdict, uint32_t seq, VerifyTables vt) {
    __shared__ uint32_t s_need;
    __shared__ PyramidLds lds;
    if (threadIdx.x == 0) s_need = 0;
    __syncthreads();
    const uint32_t tiles_y = n_tiles / tiles_x;
    // One pass, a wave per 64-tile piece of a tile row, four pieces in flight per wave: need = bounded window && still open (inside the
    // band); the repair window; the need bitmap straight from the ballot (no clear, no atomics).  (Until round 4 the bitmap was a loop
    // of its own, a thread per word reading 32 windows 8 bytes apart one after the other: 40 us of this kernel at 3840 x 2160.)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t pieces = (tiles_x + 63u) / 64u, items = tiles_y * pieces;
    uint32_t mine = 0;
    for (uint32_t base = wave; base < items; base += 4u * 16u) {
        uint2 w[4];
        uint32_t dw[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t item = base + 16u * k, ty = item / pieces, tx = (item % pieces) * 64u + lane;
            const bool in = item < items && tx < tiles_x;
            w[k] = in ? win1[ty * tiles_x + tx] : make_uint2(0u, kKeyAll);
            dw[k] = in ? done[ty * row_words + (tx >> 5)] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t item = base + 16u * k, ty = item / pieces, tx = (item % pieces) * 64u + lane;
            const bool in = item < items && tx < tiles_x;
            const bool need = in && ty >= band_lo && ty < band_hi && w[k].y != kKeyAll && !((dw[k] >> (tx & 31u)) & 1u);
            if (in) win2[ty * tiles_x + tx] = need ? make_uint2(w[k].y, kKeyAll) : make_uint2(0u, 0u);
            const unsigned long long nb = __ballot(need);
            if (item < items && (lane & 31u) == 0u && (tx >> 5) < row_words) need_bits[ty * row_words + (tx >> 5)] = (uint32_t)(nb >> lane);
            mine += need ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(&s_need, mine);
    __syncthreads();  // (also: win2 is complete and visible to this workgroup)
    const uint32_t total = s_need;
    if (threadIdx.x == 0) {
        *d_need = total;
        // {seq, tiles that need the repair round} to pinned host memory: one system-scope 64-bit store, polled by gsx_render
        if (host_verdict) __hip_atomic_store(host_verdict, ((unsigned long long)seq << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (total == 0) {  // the repair round's kernels fall through on *d_need == 0
        // (nothing will rebuild the repair slab's block table; its ranges must not keep the main round's — every tile's repair window
        //  is empty, so nothing would be read through them, but GSX_VALIDATE checks what is there)
        if (vt.table)
            for (uint32_t b = threadIdx.x; b < vt.grid.blocks_x * vt.grid.blocks_y; b += 1024u) vt.ranges[b] = make_uint2(0u, 0u);
        return;
    }
    if (vt.pyr2_data) build_window_pyramid<1024>(win2, vt.pyr2, vt.pyr2_data, nullptr, lds);
    if (vt.table) {
        const uint32_t n_blocks = vt.grid.blocks_x * vt.grid.blocks_y;
        for (uint32_t b = threadIdx.x >> 6; b < n_blocks; b += 16u)
            wave_block_table_entry(vt.grid, b, tiles_x, tiles_y, band_lo, band_hi, done, row_words, win2, vt.table, vt.ranges);
    }
}

// the model's windows for its next frame.  A tile that was saturated before this model was composited (done_before,
// nearer models) says nothing about this model's depths; a tile still open afterwards makes its neighbourhood unbounded.
// One workgroup per 16x16 block of tiles: the block's neighbourhood (radius <= 16) is staged in LDS once and the
// (2r+1)^2 maximum is taken separably — rows, then columns — instead of (2r+1)^2 global loads per tile (169 at the radius
// the frame-parallel mode uses: 40 us; 14 us at radius 3).
constexpr int kNextBlock = 16, kNextMaxR = 16, kNextSpan = kNextBlock + 2 * kNextMaxR;
__global__ __launch_bounds__(256) void k_spec_next(const uint32_t* __restrict__ tile_sat, const uint32_t* __restrict__ done,
                                                    const uint32_t* __restrict__ done_before, uint32_t row_words,
                                                    uint32_t tiles_x, uint32_t tiles_y, float gain, int radius,
                                                    uint2* __restrict__ win_next, int band_lo, int band_hi) {
    // per staged tile: depth (0 = contributes nothing) and an "open" flag; outside the band / image: neither
    __shared__ float s_deep[kNextSpan][kNextSpan + 1];
    __shared__ unsigned char s_open[kNextSpan][kNextSpan + 1];
    __shared__ float s_hdeep[kNextSpan][kNextBlock + 1];
    __shared__ unsigned char s_hopen[kNextSpan][kNextBlock + 1];
    const int bx = (int)blockIdx.x * kNextBlock, by = (int)blockIdx.y * kNextBlock;
    const int span = kNextBlock + 2 * radius;
    for (int i = (int)threadIdx.x; i < span * span; i += 256) {
        const int ly = i / span, lx = i - ly * span;
        const int x = bx - radius + lx, y = by - radius + ly;
        float deep = 0.0f;
        unsigned char open = 0;
        if (x >= 0 && x < (int)tiles_x && y >= band_lo && y < band_hi) {
            const uint32_t w = (uint32_t)y * row_words + ((uint32_t)x >> 5), bit = (uint32_t)x & 31u;
            if (!((done[w] >> bit) & 1u)) {
                open = 1;
            } else if (!(done_before && ((done_before[w] >> bit) & 1u))) {
                deep = __uint_as_float(tile_sat[(uint32_t)y * tiles_x + (uint32_t)x]);
            }
        }
        s_deep[ly][lx] = deep;
        s_open[ly][lx] = open;
    }
    __syncthreads();
    // rows: for every staged row and every column of the block, max / any over [x - r, x + r]
    for (int i = (int)threadIdx.x; i < span * kNextBlock; i += 256) {
        const int ly = i / kNextBlock, cx = i - ly * kNextBlock;
        float deep = 0.0f;
        unsigned char open = 0;
        for (int d = 0; d <= 2 * radius; ++d) {
            deep = fmaxf(deep, s_deep[ly][cx + d]);
            open |= s_open[ly][cx + d];
        }
        s_hdeep[ly][cx] = deep;
        s_hopen[ly][cx] = open;
    }
    __syncthreads();
    const int cx = (int)threadIdx.x & 15, cy = (int)threadIdx.x >> 4;
    const int tx = bx + cx, ty = by + cy;
    if (tx >= (int)tiles_x || ty >= (int)tiles_y) return;
    const uint32_t t = (uint32_t)ty * tiles_x + (uint32_t)tx;
    if (ty < band_lo || ty >= band_hi) {  // not this viewer's band: takes nothing
        win_next[t] = make_uint2(0u, 0u);
        return;
    }
    float deepest = 0.0f;
    unsigned char open = 0;
    for (int d = 0; d <= 2 * radius; ++d) {
        deepest = fmaxf(deepest, s_hdeep[cy + d][cx]);
        open |= s_hopen[cy + d][cx];
    }
    uint32_t hi = kKeyAll;
    if (!open) {
        const float lim = deepest * gain;
        hi = (lim < 3.0e38f) ? max(__float_as_uint(lim), 1u) : kKeyAll;
    }
    win_next[t] = make_uint2(0u, hi);
}

WindowPyramid window_pyramid_layout(uint32_t tiles_x, uint32_t tiles_y, const uint32_t* data) {
    WindowPyramid p{};
    p.data = data;
    uint32_t wx = tiles_x, wy = tiles_y, off = 0, l = 0;
    for (;;) {
        p.off[l] = off;
        p.wx[l] = wx;
        p.wy[l] = wy;
        off += wx * wy;
        ++l;
        if ((wx == 1 && wy == 1) || l == 9) break;
        wx = (wx + 1) / 2;
        wy = (wy + 1) / 2;
    }
    p.levels = l;
    p.cells_off = (off + 1u) & ~1u;
    auto shift = [](uint32_t tiles) { uint32_t s = 0; while (((tiles - 1u) >> s) > 7u) ++s; return s; };  // (tiles - 1) >> s <= 7
    p.cell_sx = shift(std::max(tiles_x, 1u));
    p.cell_sy = shift(std::max(tiles_y, 1u));
    return p;
}

size_t window_pyramid_words(uint32_t tiles_x, uint32_t tiles_y) {
    const WindowPyramid p = window_pyramid_layout(tiles_x, tiles_y, nullptr);
    return (size_t)p.cells_off + 2;  // the levels, then the 64-bit cell word of a min-of-starts pyramid
}

hipError_t launch_window_pyramid(hipStream_t s, const uint2* window, uint32_t tiles_x, uint32_t tiles_y, uint32_t* data,
                                 bool min_of_starts, const uint32_t* d_skip, uint32_t* min_ends) {
    WindowPyramid p = window_pyramid_layout(tiles_x, tiles_y, data);
    p.min_of_starts = min_of_starts ? 1u : 0u;
    GSX_LAUNCH(k_window_pyramid, dim3(1), dim3(1024), 0, s, window, p, data, d_skip, min_ends);
    return hipGetLastError();
}

hipError_t launch_spec_verify(hipStream_t s, const uint2* win1, const uint32_t* done, uint32_t row_words, uint32_t tiles_x,
                              uint32_t tiles_y, uint2* win2, uint32_t* need_bits, uint32_t* d_need, uint32_t band_lo, uint32_t band_hi,
                              unsigned long long* host_verdict, uint32_t seq, uint32_t* pyr2_data, const BlockGrid* grid, uint4* table,
                              uint2* ranges) {
    const uint32_t n_tiles = tiles_x * tiles_y;
    VerifyTables vt{};
    vt.pyr2_data = pyr2_data;
    if (pyr2_data) {
        vt.pyr2 = window_pyramid_layout(tiles_x, tiles_y, pyr2_data);
        vt.pyr2.min_of_starts = 1u;
    }
    if (grid && table) {
        vt.grid = *grid;
        vt.table = table;
        vt.ranges = ranges;
    }
    GSX_LAUNCH(k_spec_verify_fused, dim3(1), dim3(1024), 0, s, win1, done, row_words, tiles_x, n_tiles, win2, need_bits, d_need,
               std::min(band_lo, tiles_y), std::min(band_hi, tiles_y), host_verdict, seq, vt);
    return hipGetLastError();
}

hipError_t launch_spec_next(hipStream_t s, const uint32_t* tile_sat, const uint32_t* done, const uint32_t* done_before,
                            uint32_t row_words, uint32_t tiles_x, uint32_t tiles_y, float margin, uint32_t radius, uint2* win_next,
                            uint32_t band_lo, uint32_t band_hi) {
    GSX_LAUNCH(k_spec_next, dim3((tiles_x + kNextBlock - 1) / kNextBlock, (tiles_y + kNextBlock - 1) / kNextBlock), dim3(256), 0, s,
                       tile_sat, done, done_before, row_words, tiles_x, tiles_y, 1.0f + margin, (int)std::min<uint32_t>(radius, kNextMaxR), win_next,
                       (int)std::min(band_lo, tiles_y), (int)std::min(band_hi, tiles_y));
    return hipGetLastError();
}

}  // namespace gsx
