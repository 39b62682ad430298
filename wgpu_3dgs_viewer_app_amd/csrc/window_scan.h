// window_scan.h — which tiles' depth-key windows admit a projected record (shared by the multi-GPU pack,
// kernels_shard.hip, and the single-GPU admission pass, kernels_admit.hip).
#pragma once
#include "gsx_internal.h"

namespace gsx {

constexpr uint32_t kWindowCoop = 32;  // rectangles with more tiles are tested against the windows by the whole wave

// the whole screen as one band (the single-GPU admission passes ask "does ANY tile admit it")
__device__ inline BandEdges one_band() {
    BandEdges b;
    b.world = 1u;
    b.e[0] = 0u;
    b.e[1] = 0x10000u;
    return b;
}

// destinations of a record (key, tile rect): bit g set iff band g (tile rows [bands.e[g], bands.e[g + 1])) holds a tile of the
// rectangle whose window contains the key (window == nullptr: every touched band)
__device__ inline unsigned long long dest_mask(const uint2* __restrict__ window, uint32_t tiles_x, uint32_t key,
                                               uint32_t rx, uint32_t ry, const BandEdges& bands) {
    const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
    unsigned long long m = 0;
    if (y0 >= y1 || x0 >= x1) return 0;
    const uint32_t g0 = band_of(bands, y0), g1 = band_of(bands, y1 - 1u);
    for (uint32_t g = g0; g <= g1; ++g) {
        bool hit = window == nullptr;
        const uint32_t ya = max(y0, bands.e[g]), yb = g == g1 ? y1 : min(y1, bands.e[g + 1u]);
        if (ya >= yb) continue;  // an empty band between two others
        if (!hit) {
            for (uint32_t ty = ya; ty < yb && !hit; ++ty)
                for (uint32_t tx = x0; tx < x1; ++tx) {
                    const uint2 w = window[ty * tiles_x + tx];
                    if (key >= w.x && key < w.y) {
                        hit = true;
                        break;
                    }
                }
        }
        if (hit) m |= 1ull << g;
    }
    return m;
}

// does the rectangle hold a tile whose bit is set in `bits` (row_words u32 per tile row)?
__device__ inline bool rect_hits_bitmap(const uint32_t* __restrict__ bits, uint32_t row_words, uint32_t rx, uint32_t ry) {
    const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
    for (uint32_t ty = y0; ty < y1; ++ty)
        for (uint32_t w = x0 >> 5; w <= ((x1 - 1u) >> 5); ++w) {
            const uint32_t lo = w == (x0 >> 5) ? (x0 & 31u) : 0u, hi = w == ((x1 - 1u) >> 5) ? ((x1 - 1u) & 31u) : 31u;
            const uint32_t mask = (hi == 31u ? 0xFFFFFFFFu : ((1u << (hi + 1u)) - 1u)) & ~((1u << lo) - 1u);
            if (bits[ty * row_words + w] & mask) return true;
        }
    return false;
}

// Wave-cooperative form: every lane passes its record (key == kCulledKey: none).  Rectangles of at most 2x2 tiles —
// nine records in ten — are decided by four independent loads; rectangles of more than kWindowCoop tiles, mostly
// hidden background splats that no tile admits, are scanned by all 64 lanes.  gate (nullable): a tile bitmap; a record
// whose rectangle holds no gated tile is refused without looking at the windows (repair round: the tiles in need).
// Must be called by the whole wave.
__device__ inline unsigned long long wave_dest_mask(const uint2* __restrict__ window, uint32_t tiles_x, uint32_t kk,
                                                    uint32_t rx, uint32_t ry, const BandEdges& bands,
                                                    const uint32_t* __restrict__ gate = nullptr, uint32_t row_words = 0) {
    const uint32_t world = bands.world;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long m = 0;
    uint32_t area = 0;
    if (kk != kCulledKey && gate && !rect_hits_bitmap(gate, row_words, rx, ry)) kk = kCulledKey;
    if (kk != kCulledKey) {
        const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
        area = (x1 - x0) * (y1 - y0);
        if (!window) {
            m = dest_mask(window, tiles_x, kk, rx, ry, bands);
        } else if (x1 - x0 <= 2u && y1 - y0 <= 2u) {
            const uint32_t xb = x1 - 1u, yb = y1 - 1u;  // == x0 / y0 for a one-tile extent: the duplicates cost nothing
            const uint2 w00 = window[y0 * tiles_x + x0], w01 = window[y0 * tiles_x + xb];
            const uint2 w10 = window[yb * tiles_x + x0], w11 = window[yb * tiles_x + xb];
            const uint32_t g0 = band_of(bands, y0), g1 = band_of(bands, yb);
            if ((kk >= w00.x && kk < w00.y) || (kk >= w01.x && kk < w01.y)) m |= 1ull << g0;
            if ((kk >= w10.x && kk < w10.y) || (kk >= w11.x && kk < w11.y)) m |= 1ull << g1;
            area = 0;
        } else if (area <= kWindowCoop) {
            m = dest_mask(window, tiles_x, kk, rx, ry, bands);
        }
    }
    if (window) {
        unsigned long long big = __ballot(area > kWindowCoop);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const uint32_t brx = __shfl(rx, src, 64), bry = __shfl(ry, src, 64), bkey = __shfl(kk, src, 64);
            const uint32_t total = __shfl(area, src, 64);
            const uint32_t x0 = brx & 0xFFFFu, w = (brx >> 16) - x0, y0 = bry & 0xFFFFu;
            uint32_t lo = 0, hi = 0;  // destination bits 0..31 / 32..63
            for (uint32_t k0 = 0; k0 < total; k0 += 64) {
                const uint32_t k = k0 + lane;
                if (k < total) {
                    const uint32_t ty = y0 + k / w;
                    const uint2 ww = window[ty * tiles_x + x0 + k % w];
                    if (bkey >= ww.x && bkey < ww.y) {
                        const uint32_t g = band_of(bands, ty);
                        if (g < 32u) lo |= 1u << g; else hi |= 1u << (g - 32u);
                    }
                }
                if (world == 1u && __ballot(lo != 0u)) break;  // one destination: the first admitting tile settles it
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo |= __shfl_xor(lo, o, 64);
                hi |= __shfl_xor(hi, o, 64);
            }
            if ((int)lane == src) m = ((unsigned long long)hi << 32) | lo;
        }
    }
    return m;
}

// conservative admission against the max-pyramid of the window ends: true if SOME tile under the rectangle may admit
__device__ inline bool pyramid_admits(const WindowPyramid& p, uint32_t key, uint32_t rx, uint32_t ry) {
    const uint32_t x0 = rx & 0xFFFFu, xb = (rx >> 16) - 1u, y0 = ry & 0xFFFFu, yb = (ry >> 16) - 1u;
    if (p.min_of_starts) {
        // no tile with a window anywhere near the rectangle (an 8 x 8 grid of screen cells, one bit each): every cell the walk
        // below would load holds KEY_ALL — the same answer without a load
        const unsigned long long need = *reinterpret_cast<const unsigned long long*>(p.data + p.cells_off);
        const uint32_t cx0 = x0 >> p.cell_sx, cx1 = min(xb >> p.cell_sx, 7u), cy0 = y0 >> p.cell_sy, cy1 = min(yb >> p.cell_sy, 7u);
        const unsigned long long row = ((2ull << (cx1 - cx0)) - 1ull) << cx0;
        const uint32_t rows = cy1 - cy0 + 1u;
        const unsigned long long sel = (rows >= 8u ? ~0ull : ((1ull << (8u * rows)) - 1ull)) << (8u * cy0);
        if (((row * 0x0101010101010101ull) & sel & need) == 0ull) return false;
    }
    const uint32_t ext = max(xb - x0, yb - y0);  // extent - 1
    const uint32_t l = ext ? 32u - (uint32_t)__clz((int)ext) : 0u;
    if (l >= p.levels) return true;  // wider than the pyramid's top cell (viewports beyond 4096 px): take it
    const uint32_t* L = p.data + p.off[l];
    const uint32_t wx = p.wx[l];
    const uint32_t cx0 = x0 >> l, cx1 = min(xb >> l, wx - 1u), cy0 = y0 >> l, cy1 = min(yb >> l, p.wy[l] - 1u);
    const uint32_t a = L[cy0 * wx + cx0], b = L[cy0 * wx + cx1], c = L[cy1 * wx + cx0], d = L[cy1 * wx + cx1];
    if (p.min_of_starts) return key >= min(min(a, b), min(c, d));
    return key < max(max(a, b), max(c, d));
}

// Destinations of a record by the pyramid alone: bit g set iff the part of the rectangle inside band g MAY hold a tile that
// admits the key (four loads per band the rectangle touches, no walk over tiles).  A conservative superset of dest_mask —
// allowed on the sending side of the exchange: the receiver bins every record by the exact per-tile windows, so a record
// that travels in vain costs link bytes, never a pixel.  (The exact walk was the slowest kernel of a sharded frame: large
// rectangles are scanned by the whole wave, one after the other.)
__device__ inline unsigned long long dest_mask_pyramid(const WindowPyramid& p, uint32_t key, uint32_t rx, uint32_t ry, const BandEdges& bands) {
    const uint32_t x0 = rx & 0xFFFFu, x1 = rx >> 16, y0 = ry & 0xFFFFu, y1 = ry >> 16;
    if (key == kCulledKey || y0 >= y1 || x0 >= x1) return 0ull;
    if (bands.world == 1u) return pyramid_admits(p, key, rx, ry) ? 1ull : 0ull;  // one band: nothing to look up (96 -> 40 us over 10 M records)
    unsigned long long m = 0;
    const uint32_t g0 = band_of(bands, y0), g1 = band_of(bands, y1 - 1u);
    for (uint32_t g = g0; g <= g1; ++g) {
        const uint32_t ya = max(y0, bands.e[g]), yb = g == g1 ? y1 : min(y1, bands.e[g + 1u]);
        if (ya < yb && pyramid_admits(p, key, rx, ya | (yb << 16))) m |= 1ull << g;
    }
    return m;
}

}  // namespace gsx
