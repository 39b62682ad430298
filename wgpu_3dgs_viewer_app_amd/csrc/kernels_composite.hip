// kernels_composite.hip — tile compositor and resolve for gfx950.
//
// Replaces the raster half of the reference's K3 (`renderer.render_with_pass`, instanced quads with
// fixed-function "over" blending in back-to-front order, src/tab/scene.rs:2302-2314).  Here one
// 128-lane workgroup owns one 16x16 px tile (two pixels per lane) and walks the tile's depth-ordered splat list FRONT to
// back:  C += T*alpha*c ; T *= 1-alpha   — algebraically the same premultiplied "over" result.
// The list is staged through LDS 128 records at a time (one gather per lane, in flight while the previous batch is
// blended; every lane then reads all records as LDS broadcasts); waves vote (`__syncthreads_and`) to stop once every
// pixel of the tile has T < t_epsilon.  Models are layered by carrying (C,T) in the framebuffer: the host walks
// the reference's far->near key list (scene.rs:533-558) in reverse.
// VALU / LDS bound, not HBM bound; algorithmic bytes D*40 + W*H*16.
//
// The support decision uses exactly the oracle's operation order (spec §6): explicit fmaf, no contraction.
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

constexpr int kBatch = 128;                  // splat records staged through LDS per barrier pair (256 measured: no gain)
constexpr int kGroupTiles = 4;               // splats blended between two wave-level exit checks, per-tile lists (2 / 3 / 4 measured alike, round 3)
constexpr int kGroupBlocks = 4;              // ... block lists
constexpr int kPerLane = kBatch / 128;

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f splat2(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f32

// Blends the first `cnt` records of the LDS batch (slots up to the next multiple of kGroup must hold records no pixel
// supports) into this lane's two pixels.  Groups of kGroup splats, straight-line, one skip branch per splat; a wave whose
// pixels are all saturated leaves the batch at the next group.
// CLAMP: alpha_max < 1 or alpha_min > 0.  With the default constants (alpha_max = 1, alpha_min = 0) min(alpha_max, opacity x weight)
// and the alpha_min test change nothing — opacity <= 1 and exp(-q / 2) <= 1, so their product is <= 1 and >= 0 — and the loop is
// bound by vector issue: six instructions of the ~26 a hit costs are left out (same values, bit for bit; the INRIA-constants
// fixture and every test that sets gsx_spec_params run the clamping instantiation).
// Measured on cfg4 (a counting build, round 4): 60 % of the (record, wave) evaluations hit a pixel — the rest leave after the ten
// instructions of q and the two compares — and a hit covers 65 % of the wave's 128 pixels; ~144 evaluations and ~87 hits per wave
// and tile, 80 % of the kernel's vector instructions.
template <int MODE, int kGroup, bool CLAMP>
__device__ __forceinline__ void blend_batch(const FrameConsts& f, const uint32_t cnt, const float2* s_mean, const float4* s_conic,
                                            const float4* s_rgb, const float pxf, const v2f pyf, uint32_t& lim0, uint32_t& lim1,
                                            v2f& T, v2f& C0, v2f& C1, v2f& C2, uint32_t& stop_key) {
    for (uint32_t j0 = 0; j0 < cnt; j0 += kGroup) {
        if (!__ballot((lim0 | lim1) != 0u)) break;
        // the group's records first, so the LDS round trips overlap instead of each splat waiting for its own
        float2 gm[kGroup];
        float4 gc[kGroup];
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) {
            gm[u] = s_mean[j0 + u];
            gc[u] = s_conic[j0 + u];
        }
#pragma unroll
        for (uint32_t u = 0; u < kGroup; ++u) {
            const uint32_t j = j0 + u;
            const float2 m = gm[u];
            const float4 co = gc[u];
            const float dx = pxf - m.x;
            const v2f dy = pyf - splat2(m.y);
            // q = fma(a*dx, dx, fma(c*dy, dy, ((2b)*dx)*dy))
            const v2f q = fma2(splat2(co.x * dx), splat2(dx), fma2(splat2(co.z) * dy, dy, splat2(co.y * dx) * dy));
            const bool h0 = __float_as_uint(q.x) < lim0, h1 = __float_as_uint(q.y) < lim1;
            if (h0 || h1) {
                v2f alpha;
                if (MODE == 0) {
                    // exp(-q / 2) as ONE packed multiply and two v_exp_f32: 2^(q x (-log2(e) / 2)) (__expf(-0.5f * q) is a multiply by
                    // -0.5, a multiply by log2(e) and the same v_exp_f32 per pixel; the argument differs by at most an ulp)
                    const v2f e = splat2(-0.72134752044448170368f) * q;
                    alpha = splat2(co.w) * v2f{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                } else {
                    alpha = splat2(co.w * 1.0f);
                }
                if (CLAMP) {
                    alpha.x = fminf(f.alpha_max, alpha.x);
                    alpha.y = fminf(f.alpha_max, alpha.y);
                    alpha.x = (h0 && !(alpha.x < f.alpha_min)) ? alpha.x : 0.0f;
                    alpha.y = (h1 && !(alpha.y < f.alpha_min)) ? alpha.y : 0.0f;
                } else {
                    alpha.x = h0 ? alpha.x : 0.0f;
                    alpha.y = h1 ? alpha.y : 0.0f;
                }
                const float4 c = s_rgb[j];
                const v2f wgt = T * alpha;
                C0 = fma2(wgt, splat2(c.x), C0);
                C1 = fma2(wgt, splat2(c.y), C1);
                C2 = fma2(wgt, splat2(c.z), C2);
                T = T * (splat2(1.0f) - alpha);
                // only a blend lowers T, so T < t_eps here means saturated now or before
                lim0 = T.x < f.t_eps ? 0u : lim0;
                lim1 = T.y < f.t_eps ? 0u : lim1;
                stop_key = __float_as_uint(c.w);
            }
        }
    }
}

// One 128-lane workgroup (two waves) per 16x16 tile; every lane owns the two vertically adjacent pixels
// (x, 2r) and (x, 2r+1), so wave w covers rows 8w..8w+7.  Two pixels per lane because the loop is bound by VALU
// issue and by the LDS return path (every splat record is broadcast to all lanes): gfx950's packed fp32
// instructions (v_pk_fma/mul/add_f32) blend both pixels in one issue slot, dx and the dx-only products are shared,
// and each LDS broadcast now feeds two pixels.  Per pixel the operation sequence is exactly the oracle's (spec §6;
// packed ops round like their scalar forms), a pixel that is not hit takes alpha = 0, which leaves (C, T) unchanged
// bit for bit for finite colour records.
template <int MODE /* 0 splat (gaussian falloff), 1 constant alpha inside the cutoff */, bool CLAMP /* blend_batch */>
__global__ __launch_bounds__(128) void k_composite(const FrameConsts f, uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ list,
                                                    const float4* __restrict__ rec_a, const float4* __restrict__ rec_b,
                                                    const float4* __restrict__ rec_c, float4* __restrict__ fb,
                                                    const int carry, uint32_t* __restrict__ done_bits,
                                                    const uint32_t row_words, uint32_t* __restrict__ done_count,
                                                    const int clear_ranges, uint32_t* __restrict__ tile_sat,
                                                    uint32_t* __restrict__ row_work) {
    __shared__ float2 s_mean[kBatch];
    __shared__ uint32_t s_sat;
    __shared__ float4 s_conic[kBatch];  // (a, 2b, c, opacity)
    __shared__ float4 s_rgb[kBatch];

    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % f.tiles_x, ty = tile / f.tiles_x;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + 2u * (tid >> 4);
    const bool in0 = px < f.w_px && py < f.h_px, in1 = px < f.w_px && py + 1u < f.h_px;
    const float pxf = (float)px + 0.5f;
    const v2f pyf = v2f{(float)py + 0.5f, (float)(py + 1u) + 0.5f};
    const uint2 range = ranges[tile];
    const size_t fbo = (size_t)py * f.w_px + px;
    // Progressive mode leaves the range table clean for the next slab (saves a memset per slab).  EVERY lane reads
    // ranges[tile] itself, so the entry may only be zeroed once both waves have read it: on the early-return path
    // that is harmless (a wave that reads the zeroed entry returns as well), on the main path it happens after the
    // list loop, whose barriers every wave has passed by then.  (Zeroing it up front lost whole waves of pixels whenever
    // wave 0 ran a memory round trip ahead of the others — seen on the first frame of a 24 M-Gaussian 4K scene.)
    const bool had_entries = range.y > range.x;

    // Later depth slabs / models behind continue from the (C, T) the framebuffer already holds; a tile
    // with nothing new to blend, or already saturated, leaves it untouched.
    if (carry && (range.x >= range.y || (done_bits && ((done_bits[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u)))) {
        if (clear_ranges && tid == 0 && had_entries) ranges[tile] = make_uint2(0u, 0u);
        return;
    }
    v2f T = splat2(1.0f), C0 = splat2(0.0f), C1 = C0, C2 = C0;
    if (carry) {
        if (in0) {
            const float4 p = fb[fbo];
            C0.x = p.x; C1.x = p.y; C2.x = p.z; T.x = p.w;
        }
        if (in1) {
            const float4 p = fb[fbo + f.w_px];
            C0.y = p.x; C1.y = p.y; C2.y = p.z; T.y = p.w;
        }
    }
    // The support test `!done && 0 <= q <= k2` as ONE unsigned compare per pixel: non-negative floats order like their bit
    // patterns, negative values and NaN have larger patterns than any finite k2, so it is `bits(q) < lim` with
    // lim = bits(k2) + 1 while the pixel is live and 0 once it is saturated (or outside the image).  This keeps the
    // predicate logic off the scalar unit, which was busier than the vector ALU in this loop (110 M vs 77 M instructions).
    const uint32_t live = f.k2 > 0.0f ? __float_as_uint(f.k2) + 1u : (f.k2 == 0.0f ? 1u : 0u);
    uint32_t lim0 = (in0 && !(T.x < f.t_eps)) ? live : 0u, lim1 = (in1 && !(T.y < f.t_eps)) ? live : 0u;
    uint32_t stop_key = 0;  // depth key of this lane's last hit: once both its pixels are saturated, the key of the splat
                            // that saturated the later one (keys ascend along the list)
    if (tid == 0) s_sat = 0;

    // software pipeline: the gather of batch b+1 (list -> three record planes, dependent random loads; two records per
    // lane) is in flight while batch b is blended out of LDS.  Slots past the end of the list hold a record no pixel
    // supports (q = +inf), so the blend loop can run in fixed groups of four.
    const float4 pad_a = make_float4(3.0e38f, 3.0e38f, 0.0f, 0.0f), pad_b = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    float4 pa[kPerLane], pb[kPerLane], pc4[kPerLane];
#pragma unroll
    for (int k = 0; k < kPerLane; ++k) {
        pa[k] = pad_a;
        pb[k] = pad_b;
        pc4[k] = make_float4(0, 0, 0, 0);
        const uint32_t at = range.x + tid + 128u * k;
        if (at < range.y) {
            const uint32_t idx = list[at];
            pa[k] = rec_a[idx];
            pb[k] = rec_b[idx];
            pc4[k] = rec_c[idx];
        }
    }
    uint32_t base = range.x;
    for (; base < range.y; base += kBatch) {
        // vote + barrier: also protects the LDS batch of the previous iteration
        if (__syncthreads_and((lim0 | lim1) == 0u)) break;
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) {
            s_mean[tid + 128u * k] = make_float2(pa[k].x, pa[k].y);
            s_conic[tid + 128u * k] = make_float4(pb[k].x, 2.0f * pb[k].y, pb[k].z, pb[k].w);
            s_rgb[tid + 128u * k] = pc4[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPerLane; ++k) {
            const uint32_t nxt = base + kBatch + tid + 128u * k;
            pa[k] = pad_a;
            pb[k] = pad_b;
            if (nxt < range.y) {
                const uint32_t idx = list[nxt];
                pa[k] = rec_a[idx];
                pb[k] = rec_b[idx];
                pc4[k] = rec_c[idx];
            }
        }
        const uint32_t cnt = min((uint32_t)kBatch, range.y - base);
        blend_batch<MODE, kGroupTiles, CLAMP>(f, cnt, s_mean, s_conic, s_rgb, pxf, pyf, lim0, lim1, T, C0, C1, C2, stop_key);
    }
    if (in0) fb[fbo] = make_float4(C0.x, C1.x, C2.x, T.x);
    if (in1) fb[fbo + f.w_px] = make_float4(C0.y, C1.y, C2.y, T.y);
    if (clear_ranges && tid == 0 && had_entries) ranges[tile] = make_uint2(0u, 0u);
    // multi-GPU: what this tile cost (tile_work, gsx_internal.h: here every entry walked is blended); the next frame's bands are
    // balanced by the rows' sums (gsx_shard_frame.cpp, k_shard_verify)
    if (row_work && tid == 0) atomicAdd(&row_work[ty], tile_work(min(base, range.y) - range.x, min(base, range.y) - range.x, range.y - range.x, 1u));
    if (done_bits && __syncthreads_and((lim0 | lim1) == 0u)) {
        // the tile saturated in this launch: its last pixels stopped here, behind everything blended earlier
        if (tile_sat) {
            if (stop_key) atomicMax(&s_sat, stop_key);
            __syncthreads();
        }
        if (tid == 0) {
            atomicOr(&done_bits[ty * row_words + (tx >> 5)], 1u << (tx & 31u));
            atomicAdd(done_count, 1u);  // at most one per tile per frame
            if (tile_sat) tile_sat[tile] = max(s_sat, 1u);  // saturation depth key (multi-GPU speculation); 0 = open
        }
    }
}

// Block lists (kernels_bin.hip): the tile's BLOCK has one depth-ordered list of slab positions; brec[position] = {rect x, rect y,
// depth key, record index}.  The workgroup walks it 128 candidates at a time and keeps, in order, those whose rectangle
// covers this tile and whose key lies in the tile's window — the exact per-tile decision, a handful of register compares.
// Software pipeline: while the takers of chunk c are blended out of LDS, the candidate loads of chunk c + 1 (brec) and c + 2
// (list) are in flight.  The takers' 48-byte records are gathered when their LDS slots are written, NOT held across the blend:
// twelve registers less is one more wave per SIMD (84 -> 74 VGPRs, occupancy 5 -> 6), worth more than the overlap (+1.2 %).
// What a tile cost, in the units the dispatch order is decided by (tile_order_job): measured on cfg4 (tools/tile_profile.py), a tile's time under a full chip is
// ~16 us + 0.95 us per list chunk walked + 0.11 us per taker blended; one unit = 0.44 us.
// what a tile cost, as the compositor leaves it in tile_cost: chunks walked << 16 | candidates taken (tile_order_job weighs the two —
// (9 chunks + takers) / 4, fitted to the tile profile — and keeps the longest walk for the host: gsx_internal.h)
__device__ __forceinline__ uint32_t tile_cost_units(uint32_t chunks, uint32_t taken) { return (min(chunks, 0x7FFFu) << 16) | min(taken, 0xFFFFu); }

constexpr int kCand = 1;                         // candidates per lane and iteration (1 / 2 / 3 measured alike, round 2)
constexpr uint32_t kChunk = 128u * kCand;

template <int MODE, bool CLAMP, bool SORTED /* brec is in list order (below) */>
__global__ __launch_bounds__(128) void k_composite_blocks(const FrameConsts f, const uint2* __restrict__ ranges,
                                                           const uint32_t* __restrict__ list, const uint4* __restrict__ brec,
                                                           const float4* __restrict__ rec_a, const float4* __restrict__ rec_b,
                                                           const float4* __restrict__ rec_c, float4* __restrict__ fb,
                                                           const int carry, uint32_t* __restrict__ done_bits,
                                                           const uint32_t row_words, uint32_t* __restrict__ done_count,
                                                           uint32_t* __restrict__ tile_sat, const uint2* __restrict__ window,
                                                           const uint32_t row_lo, const uint32_t row_hi, const uint32_t bsx,
                                                           const uint32_t bsy, const uint32_t blocks_x, uint32_t* __restrict__ row_work,
                                                           const SlabStats* __restrict__ stats, const uint32_t j1,
                                                           const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ sorted_idx,
                                                           const uint32_t* __restrict__ sorted_keys, uint4* __restrict__ tile_prof,
                                                           const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ tile_cost,
                                                           const uint32_t* __restrict__ rect8 /* slab shading: rectangles of records nobody shaded */) {
    const unsigned long long t_start = tile_prof ? wall_clock64() : 0ull;  // (development: gsx_debug_tile_profile)
    __shared__ float2 s_mean[kChunk + kGroupBlocks];
    __shared__ float4 s_conic[kChunk + kGroupBlocks];
    __shared__ float4 s_rgb[kChunk + kGroupBlocks];
    __shared__ uint32_t s_sat, s_w[2][kCand];

    // The launch is as slow as its tail: the dispatcher hands out workgroups in index order, twelve a CU, and a tile takes 15 to
    // 100 us — measured (tools/tile_profile.py, cfg4): the last 55 of 187 us ran with under a sixth of the slots occupied, and
    // list scheduling of the same durations with the expensive half first ends 35 us earlier (finer classes gain nothing: what a
    // tile cost a frame ago predicts its time only roughly).  So workgroup i composites tile_order[i] — the tiles that cost more
    // than the average in the model's frame before, then the others, each class in index order (tile_order_job, gsx_internal.h:
    // one more workgroup of the frame's first block-table kernel).  A schedule, not data: any permutation renders the same pixels.
    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
    const uint32_t tx = tile % f.tiles_x, ty = tile / f.tiles_x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + 2u * (tid >> 4);
    const bool in0 = px < f.w_px && py < f.h_px, in1 = px < f.w_px && py + 1u < f.h_px;
    const float pxf = (float)px + 0.5f;
    const v2f pyf = v2f{(float)py + 0.5f, (float)(py + 1u) + 0.5f};
    const size_t fbo = (size_t)py * f.w_px + px;
    const bool owned = ty >= row_lo && ty < row_hi;
    // (another rank's tile: nothing to composite and nothing to clear — the band gather writes it, or nobody looks at it; clearing them was
    //  7/8 of a 3840x2160 framebuffer written by every rank of 8 for every frame's first model)
    if (!owned) return;
    uint2 range = ranges[((ty - row_lo) >> bsy) * blocks_x + (tx >> bsx)];  // (BlockGrid: block rows count from row_lo)
    const uint2 win = window ? window[tile] : make_uint2(0u, 0xFFFFFFFFu);
    if (win.x >= win.y) range = make_uint2(0u, 0u);  // takes nothing
    // The slab's entries did not fit the pair buffers (k_scan_block_sums cut it: rare, the host grows the buffers when it learns of
    // it): the splats [cut, end) of the depth order never reached the lists.  The tile composites them itself, BEHIND its list,
    // pair-free — it scans them 128 at a time and keeps those whose rectangle and window take it; the per-pixel operation sequence
    // is that of a frame with larger buffers.  (Until round 4 a launch of its own behind every compositor launch: k_composite_spill,
    // which the per-tile-list compositor still uses.)
    const uint32_t spill_end = stats ? min(j1, *d_n) : 0u, spill_cut = stats ? stats->slab_cut : 0u;
    const bool tile_done = done_bits && ((done_bits[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u);
    const bool spill = spill_cut < spill_end && owned && !tile_done;
    if (carry && (range.x >= range.y || tile_done) && !spill) return;
    if (tile_done) range = make_uint2(0u, 0u);
    v2f T = splat2(1.0f), C0 = splat2(0.0f), C1 = C0, C2 = C0;
    if (carry) {
        if (in0) {
            const float4 p = fb[fbo];
            C0.x = p.x; C1.x = p.y; C2.x = p.z; T.x = p.w;
        }
        if (in1) {
            const float4 p = fb[fbo + f.w_px];
            C0.y = p.x; C1.y = p.y; C2.y = p.z; T.y = p.w;
        }
    }
    const uint32_t live = f.k2 > 0.0f ? __float_as_uint(f.k2) + 1u : (f.k2 == 0.0f ? 1u : 0u);
    uint32_t lim0 = (in0 && !(T.x < f.t_eps)) ? live : 0u, lim1 = (in1 && !(T.y < f.t_eps)) ? live : 0u;
    uint32_t stop_key = 0;
    if (tid == 0) s_sat = 0;

    // Candidate k of a chunk sits at list position chunk + 128 k + tid: sub-chunk k (128 consecutive candidates) comes before
    // sub-chunk k + 1, inside one the waves in order, inside a wave the lanes — LDS slots are handed out in that order.
    // stage 1 registers: the next chunk's candidates; stage 2 registers: the gathered takers of the chunk before
    // (the list -> brec chain is split over two iterations: a load whose address is another load's result would stall the
    // wave before the blend it is meant to hide under)
    // SORTED: brec is in LIST order — the block sort's write-out carried the records along (kernels_sort.hip), a chunk's candidates
    // are 2 KB side by side, no list -> brec chain.  The frame chooses it for models whose lists are long (gsx_frame.cpp): a scene
    // where nothing saturates walks ~1700 entries per tile, one dependent 64-byte line each (translucent leg: 84 -> 112 fps);
    // on cfg4's ~110 per tile the 16 bytes more per entry the sort then moves cost what the compositor gains.
    uint4 cand[kCand];
    bool cand_ok[kCand];
    uint32_t ent[kCand];   // list entries of the chunk after the candidates'
    bool ent_ok[kCand];
#pragma unroll
    for (int k = 0; k < kCand; ++k) {
        const uint32_t at = range.x + 128u * k + tid;
        cand_ok[k] = at < range.y;
        cand[k] = make_uint4(0u, 0u, 0u, 0u);
        ent_ok[k] = false;
        ent[k] = 0;
        if (SORTED) {
            if (cand_ok[k]) cand[k] = brec[at];
        } else {
            if (cand_ok[k]) cand[k] = brec[list[at]];
            ent_ok[k] = at + kChunk < range.y;
            if (ent_ok[k]) ent[k] = list[at + kChunk];
        }
    }
    bool g_take[kCand];
    uint32_t g_my[kCand], g_cnt[kCand];
    uint32_t g_idx[kCand];   // the takers' records are gathered when their LDS slots are written, not held across the blend
#pragma unroll
    for (int k = 0; k < kCand; ++k) {
        g_take[k] = false;
        g_my[k] = g_cnt[k] = 0;
        g_idx[k] = 0;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // iteration i: write the takers gathered in iteration i - 1 to LDS, filter chunk i and start its gathers, prefetch the
    // candidates of chunk i + 1, blend; one more iteration drains the pipeline
    uint32_t base = range.x, taken = 0;
    for (; base < range.y + kChunk; base += kChunk) {
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < kCand; ++k) s_w[wave][k] = g_cnt[k];
        }
        // vote + barrier: also protects the LDS batch of the previous iteration and publishes s_w
        if (__syncthreads_and((lim0 | lim1) == 0u)) break;
        uint32_t cnt = 0;
#pragma unroll
        for (int k = 0; k < kCand; ++k) {
            const uint32_t w0 = s_w[0][k], w1 = s_w[1][k];
            if (g_take[k]) {
                const uint32_t slot = cnt + (wave ? w0 : 0u) + g_my[k];
                const float4 pa1 = rec_a[g_idx[k]], pb1 = rec_b[g_idx[k]];
                s_rgb[slot] = rec_c[g_idx[k]];
                s_mean[slot] = make_float2(pa1.x, pa1.y);
                s_conic[slot] = make_float4(pb1.x, 2.0f * pb1.y, pb1.z, pb1.w);
            }
            cnt += w0 + w1;
        }
        if (tid < (uint32_t)kGroupBlocks) {  // the blend loop runs in whole groups
            s_mean[cnt + tid] = make_float2(3.0e38f, 3.0e38f);
            s_conic[cnt + tid] = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            s_rgb[cnt + tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        // filter this chunk's candidates (already in registers) and start the gathers of its takers
#pragma unroll
        for (int k = 0; k < kCand; ++k) {
            const uint4 c = cand[k];
            g_take[k] = cand_ok[k] && tx >= (c.x & 0xFFFFu) && tx < (c.x >> 16) && ty >= (c.y & 0xFFFFu) && ty < (c.y >> 16) &&
                        c.z >= win.x && c.z < win.y;
            const unsigned long long bal = __ballot(g_take[k]);
            g_my[k] = (uint32_t)__popcll(bal & lt);
            g_cnt[k] = (uint32_t)__popcll(bal);
            g_idx[k] = c.w;
        }
        // candidates of the chunk after (their list entries arrived an iteration ago), list entries of the chunk after that
#pragma unroll
        for (int k = 0; k < kCand; ++k) {
            if (SORTED) {
                const uint32_t nn = base + kChunk + 128u * k + tid;
                cand_ok[k] = nn < range.y;
                if (cand_ok[k]) cand[k] = brec[nn];
            } else {
                cand_ok[k] = ent_ok[k];
                if (cand_ok[k]) cand[k] = brec[ent[k]];
                const uint32_t nn = base + 2u * kChunk + 128u * k + tid;
                ent_ok[k] = nn < range.y;
                if (ent_ok[k]) ent[k] = list[nn];
            }
        }
        taken += cnt;
        if (cnt) blend_batch<MODE, kGroupBlocks, CLAMP>(f, cnt, s_mean, s_conic, s_rgb, pxf, pyf, lim0, lim1, T, C0, C1, C2, stop_key);
    }
    if (spill) {
        for (uint32_t sb = spill_cut; sb < spill_end; sb += 128u) {
            if (__syncthreads_and((lim0 | lim1) == 0u)) break;  // (also protects the LDS batch of the previous iteration)
            const uint32_t j = sb + tid;
            bool hit = false;
            uint32_t idx = 0;
            float4 a = make_float4(0, 0, 0, 0);
            if (j < spill_end) {
                idx = sorted_idx[j];
                uint32_t rx, ry;
                rec_rect(rec_a, rect8, idx, rx, ry);
                hit = tx >= (rx & 0xFFFFu) && tx < (rx >> 16) && ty >= (ry & 0xFFFFu) && ty < (ry >> 16);
                if (hit && window) {
                    const uint32_t key = sorted_keys[j];
                    hit = key >= win.x && key < win.y;
                }
                if (hit) a = rec_a[idx];   // (a record that hits was taken by the tile's block: it is shaded)
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) s_w[wave][0] = (uint32_t)__popcll(bal);
            __syncthreads();
            const uint32_t cnt = s_w[0][0] + s_w[1][0];
            const uint32_t slot = (wave ? s_w[0][0] : 0u) + (uint32_t)__popcll(bal & lt);
            if (hit) {
                const float4 b = rec_b[idx];
                s_mean[slot] = make_float2(a.x, a.y);
                s_conic[slot] = make_float4(b.x, 2.0f * b.y, b.z, b.w);
                s_rgb[slot] = rec_c[idx];
            }
            if (tid < (uint32_t)kGroupBlocks) {
                s_mean[cnt + tid] = make_float2(3.0e38f, 3.0e38f);
                s_conic[cnt + tid] = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
                s_rgb[cnt + tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            __syncthreads();
            if (cnt) blend_batch<MODE, kGroupBlocks, CLAMP>(f, cnt, s_mean, s_conic, s_rgb, pxf, pyf, lim0, lim1, T, C0, C1, C2, stop_key);
        }
    }
    if (in0) fb[fbo] = make_float4(C0.x, C1.x, C2.x, T.x);
    if (in1) fb[fbo + f.w_px] = make_float4(C0.y, C1.y, C2.y, T.y);
    if (row_work && tid == 0)  // (as k_composite; a block's list is shared by its tiles)
        atomicAdd(&row_work[ty], tile_work(min(base, range.y) - range.x, taken, range.y - range.x, 1u << (bsx + bsy)));
    if (tile_cost && tid == 0) tile_cost[tile] += tile_cost_units((min(base, range.y) - range.x + kChunk - 1u) / kChunk, taken);
    if (tile_prof && tid == 0)  // {start, duration in 10 ns ticks, list entries walked | list length << 16 (in chunks), takers blended}
        tile_prof[tile] = make_uint4((uint32_t)t_start, (uint32_t)(wall_clock64() - t_start),
                                     ((min(base, range.y) - range.x + kChunk - 1u) / kChunk) | (((range.y - range.x + kChunk - 1u) / kChunk) << 16), taken);
    if (done_bits && __syncthreads_and((lim0 | lim1) == 0u)) {
        if (tile_sat) {
            if (stop_key) atomicMax(&s_sat, stop_key);
            __syncthreads();
        }
        if (tid == 0) {
            atomicOr(&done_bits[ty * row_words + (tx >> 5)], 1u << (tx & 31u));
            atomicAdd(done_count, 1u);
            if (tile_sat) tile_sat[tile] = max(s_sat, 1u);
        }
    }
}

// An overflowing slab's tail: the splats [stats->slab_cut, min(j1, *d_n)) of the depth order never reached the tile-pair
// buffers (k_scan_block_sums cut the slab where they were full).  One workgroup per tile that is still open scans them 128
// at a time, keeps — in depth order — those whose tile rectangle (and depth-key window) takes this tile, and blends them
// with the compositor's own code: the per-pixel operation sequence is that of a frame with larger buffers, so the pixels are
// the same.  O(tiles x tail) rectangle tests: a rare path that trades speed for never delivering an incomplete frame; the
// host grows the buffers as soon as it learns of the overflow.  A slab that was not cut: two loads, then return.
template <int MODE, bool CLAMP>
__global__ __launch_bounds__(128) void k_composite_spill(const FrameConsts f, const SlabStats* __restrict__ stats, const uint32_t j1,
                                                          const uint32_t* __restrict__ d_n, const uint32_t* __restrict__ sorted_idx,
                                                          const uint32_t* __restrict__ sorted_keys, const float4* __restrict__ rec_a,
                                                          const float4* __restrict__ rec_b, const float4* __restrict__ rec_c,
                                                          float4* __restrict__ fb, uint32_t* __restrict__ done_bits,
                                                          const uint32_t row_words, uint32_t* __restrict__ done_count,
                                                          uint32_t* __restrict__ tile_sat, const uint32_t row_lo, const uint32_t row_hi,
                                                          const uint2* __restrict__ window) {
    const uint32_t end = min(j1, *d_n), cut = stats->slab_cut;
    if (cut >= end) return;
    __shared__ float2 s_mean[128 + kGroupTiles];
    __shared__ float4 s_conic[128 + kGroupTiles];
    __shared__ float4 s_rgb[128 + kGroupTiles];
    __shared__ uint32_t s_sat, s_w[2];
    // a fixed, small grid strides over the tiles: the launch that finds nothing to do (every frame but the rare one that
    // overflowed) costs a kernel boundary, not the dispatch of one workgroup per tile (4.5 -> ~3 us at 1080p)
    for (uint32_t tile = blockIdx.x; tile < f.tiles_x * f.tiles_y; tile += gridDim.x) {
    __syncthreads();  // the previous tile's LDS batch and s_sat are done with
    const uint32_t tx = tile % f.tiles_x, ty = tile / f.tiles_x;
    if (ty < row_lo || ty >= row_hi) continue;
    if (done_bits && ((done_bits[ty * row_words + (tx >> 5)] >> (tx & 31u)) & 1u)) continue;
    const uint2 win = window ? window[tile] : make_uint2(0u, 0xFFFFFFFFu);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + 2u * (tid >> 4);
    const bool in0 = px < f.w_px && py < f.h_px, in1 = px < f.w_px && py + 1u < f.h_px;
    const float pxf = (float)px + 0.5f;
    const v2f pyf = v2f{(float)py + 0.5f, (float)(py + 1u) + 0.5f};
    const size_t fbo = (size_t)py * f.w_px + px;
    v2f T = splat2(1.0f), C0 = splat2(0.0f), C1 = C0, C2 = C0;
    if (in0) {  // the slab's k_composite has run: the framebuffer holds every pixel
        const float4 p = fb[fbo];
        C0.x = p.x; C1.x = p.y; C2.x = p.z; T.x = p.w;
    }
    if (in1) {
        const float4 p = fb[fbo + f.w_px];
        C0.y = p.x; C1.y = p.y; C2.y = p.z; T.y = p.w;
    }
    const uint32_t live = f.k2 > 0.0f ? __float_as_uint(f.k2) + 1u : (f.k2 == 0.0f ? 1u : 0u);
    uint32_t lim0 = (in0 && !(T.x < f.t_eps)) ? live : 0u, lim1 = (in1 && !(T.y < f.t_eps)) ? live : 0u;
    uint32_t stop_key = 0;
    if (tid == 0) s_sat = 0;
    for (uint32_t base = cut; base < end; base += 128u) {
        if (__syncthreads_and((lim0 | lim1) == 0u)) break;  // also protects the LDS batch of the previous iteration
        const uint32_t j = base + tid;
        bool hit = false;
        uint32_t idx = 0;
        float4 a = make_float4(0, 0, 0, 0);
        if (j < end) {
            idx = sorted_idx[j];
            a = rec_a[idx];
            const uint32_t rx = __float_as_uint(a.z), ry = __float_as_uint(a.w);
            hit = tx >= (rx & 0xFFFFu) && tx < (rx >> 16) && ty >= (ry & 0xFFFFu) && ty < (ry >> 16);
            if (hit && window) {
                const uint32_t key = sorted_keys[j];
                hit = key >= win.x && key < win.y;
            }
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        const uint32_t cnt = s_w[0] + s_w[1];
        const uint32_t slot = (wave ? s_w[0] : 0u) + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (hit) {
            const float4 b = rec_b[idx];
            s_mean[slot] = make_float2(a.x, a.y);
            s_conic[slot] = make_float4(b.x, 2.0f * b.y, b.z, b.w);
            s_rgb[slot] = rec_c[idx];
        }
        if (tid < (uint32_t)kGroupTiles) {  // the blend loop runs in whole groups: records no pixel supports behind the last hit
            s_mean[cnt + tid] = make_float2(3.0e38f, 3.0e38f);
            s_conic[cnt + tid] = make_float4(1.0f, 0.0f, 1.0f, 0.0f);
            s_rgb[cnt + tid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        blend_batch<MODE, kGroupTiles, CLAMP>(f, cnt, s_mean, s_conic, s_rgb, pxf, pyf, lim0, lim1, T, C0, C1, C2, stop_key);
    }
    if (in0) fb[fbo] = make_float4(C0.x, C1.x, C2.x, T.x);
    if (in1) fb[fbo + f.w_px] = make_float4(C0.y, C1.y, C2.y, T.y);
    if (done_bits && __syncthreads_and((lim0 | lim1) == 0u)) {
        if (tile_sat) {
            if (stop_key) atomicMax(&s_sat, stop_key);
            __syncthreads();
        }
        if (tid == 0) {
            atomicOr(&done_bits[ty * row_words + (tx >> 5)], 1u << (tx & 31u));
            atomicAdd(done_count, 1u);
            if (tile_sat) tile_sat[tile] = max(s_sat, 1u);
        }
    }
    }
}

__global__ __launch_bounds__(256) void k_clear_fb(float4* __restrict__ fb, uint32_t n) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) fb[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
}

// (premultiplied rgb, T) over a background colour -> RGBA8 UNORM, what the egui target would hold
__global__ __launch_bounds__(256) void k_resolve_rgba8(const float4* __restrict__ fb, uint32_t n, float br, float bg,
                                                        float bb, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 p = fb[i];
    float r = fminf(fmaxf(fmaf(p.w, br, p.x), 0.0f), 1.0f);
    float g = fminf(fmaxf(fmaf(p.w, bg, p.y), 0.0f), 1.0f);
    float b = fminf(fmaxf(fmaf(p.w, bb, p.z), 0.0f), 1.0f);
    float a = fminf(fmaxf(1.0f - p.w, 0.0f), 1.0f);
    uint32_t R = (uint32_t)floorf(r * 255.0f + 0.5f), G = (uint32_t)floorf(g * 255.0f + 0.5f);
    uint32_t B = (uint32_t)floorf(b * 255.0f + 0.5f), A = (uint32_t)floorf(a * 255.0f + 0.5f);
    out[i] = R | (G << 8) | (B << 16) | (A << 24);
}

hipError_t launch_composite(hipStream_t s, const FrameConsts& f, uint2* ranges, const uint32_t* list,
                            const Records& rec, float4* fb, bool carry, uint32_t* done, uint32_t row_words,
                            uint32_t* d_done_count, bool clear_ranges, uint32_t* tile_sat, uint32_t* row_work) {
    dim3 grid(f.tiles_x * f.tiles_y), block(128);
    const bool clamp = f.alpha_max < 1.0f || f.alpha_min > 0.0f;  // (blend_batch: the default constants need no clamping)
    if (f.display_mode == GSX_DISPLAY_SPLAT) {
        if (clamp) GSX_LAUNCH((k_composite<0, true>), grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat, row_work);
        else GSX_LAUNCH((k_composite<0, false>), grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat, row_work);
    } else {
        if (clamp) GSX_LAUNCH((k_composite<1, true>), grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat, row_work);
        else GSX_LAUNCH((k_composite<1, false>), grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat, row_work);
    }
    return hipGetLastError();
}

hipError_t launch_composite_blocks(hipStream_t s, const FrameConsts& f, const uint2* ranges, const uint32_t* list, const uint4* brec,
                                   const Records& rec, float4* fb, bool carry, uint32_t* done, uint32_t row_words,
                                   uint32_t* d_done_count, uint32_t* tile_sat, const uint2* window, uint32_t row_lo,
                                   uint32_t row_hi, uint32_t bsx, uint32_t bsy, uint32_t* row_work, const SlabStats* stats, uint32_t j1,
                                   const uint32_t* d_n, const uint32_t* sorted_idx, const uint32_t* sorted_keys, uint4* tile_prof,
                                   const uint32_t* tile_order, uint32_t* tile_cost, const uint32_t* rect8) {
    dim3 grid(f.tiles_x * f.tiles_y), block(128);
    const uint32_t blocks_x = (f.tiles_x + (1u << bsx) - 1u) >> bsx;
    const bool clamp = f.alpha_max < 1.0f || f.alpha_min > 0.0f;  // (blend_batch: the default constants need no clamping)
    const bool sorted = list == nullptr;                          // brec in list order (the block sort gathered it)
#define GSX_CB(M, C, S)                                                                                                              \
    GSX_LAUNCH((k_composite_blocks<M, C, S>), grid, block, 0, s, f, ranges, list, brec, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, \
               d_done_count, tile_sat, window, row_lo, row_hi, bsx, bsy, blocks_x, row_work, stats, j1, d_n, sorted_idx, sorted_keys, tile_prof,    \
               tile_order, tile_cost, rect8)
    if (f.display_mode == GSX_DISPLAY_SPLAT) {
        if (clamp) { if (sorted) GSX_CB(0, true, true); else GSX_CB(0, true, false); }
        else { if (sorted) GSX_CB(0, false, true); else GSX_CB(0, false, false); }
    } else {
        if (clamp) { if (sorted) GSX_CB(1, true, true); else GSX_CB(1, true, false); }
        else { if (sorted) GSX_CB(1, false, true); else GSX_CB(1, false, false); }
    }
#undef GSX_CB
    return hipGetLastError();
}

hipError_t launch_composite_spill(hipStream_t s, const FrameConsts& f, const SlabStats* stats, uint32_t j1, const uint32_t* d_n,
                                  const uint32_t* sorted_idx, const uint32_t* sorted_keys, const Records& rec, float4* fb,
                                  uint32_t* done, uint32_t row_words, uint32_t* d_done_count, uint32_t* tile_sat, uint32_t row_lo,
                                  uint32_t row_hi, const uint2* window) {
    dim3 grid(std::min<uint32_t>(f.tiles_x * f.tiles_y, 2048u)), block(128);
    const bool clamp = f.alpha_max < 1.0f || f.alpha_min > 0.0f;  // (blend_batch: the default constants need no clamping)
    if (f.display_mode == GSX_DISPLAY_SPLAT) {
        if (clamp) GSX_LAUNCH((k_composite_spill<0, true>), grid, block, 0, s, f, stats, j1, d_n, sorted_idx, sorted_keys, rec.a, rec.b, rec.c, fb, done,
                           row_words, d_done_count, tile_sat, row_lo, row_hi, window);
        else GSX_LAUNCH((k_composite_spill<0, false>), grid, block, 0, s, f, stats, j1, d_n, sorted_idx, sorted_keys, rec.a, rec.b, rec.c, fb, done,
                           row_words, d_done_count, tile_sat, row_lo, row_hi, window);
    } else {
        if (clamp) GSX_LAUNCH((k_composite_spill<1, true>), grid, block, 0, s, f, stats, j1, d_n, sorted_idx, sorted_keys, rec.a, rec.b, rec.c, fb, done,
                           row_words, d_done_count, tile_sat, row_lo, row_hi, window);
        else GSX_LAUNCH((k_composite_spill<1, false>), grid, block, 0, s, f, stats, j1, d_n, sorted_idx, sorted_keys, rec.a, rec.b, rec.c, fb, done,
                           row_words, d_done_count, tile_sat, row_lo, row_hi, window);
    }
    return hipGetLastError();
}

hipError_t launch_clear_fb(hipStream_t s, float4* fb, uint32_t n_px) {
    if (!n_px) return hipSuccess;
    GSX_LAUNCH(k_clear_fb, dim3((n_px + 255) / 256), dim3(256), 0, s, fb, n_px);
    return hipGetLastError();
}

hipError_t launch_resolve_rgba8(hipStream_t s, const float4* fb, uint32_t n_px, float bg_r, float bg_g, float bg_b,
                                uint32_t* out_rgba8) {
    if (!n_px) return hipSuccess;
    GSX_LAUNCH(k_resolve_rgba8, dim3((n_px + 255) / 256), dim3(256), 0, s, fb, n_px, bg_r, bg_g, bg_b, out_rgba8);
    return hipGetLastError();
}

}  // namespace gsx
