// kernels_composite.hip — tile compositor and resolve for gfx950.
//
// Replaces the raster half of the reference's K3 (`renderer.render_with_pass`, instanced quads with
// fixed-function "over" blending in back-to-front order, src/tab/scene.rs:2302-2314).  Here one
// 256-lane workgroup owns one 16x16 px tile and walks the tile's depth-ordered splat list FRONT to
// back:  C += T*alpha*c ; T *= 1-alpha   — algebraically the same premultiplied "over" result.
// The list is staged through LDS 256 records at a time (one gather per lane, then every lane reads
// all 256 records as LDS broadcasts); waves vote (`__syncthreads_and`) to stop once every pixel of the
// tile has T < t_epsilon.  Models are layered by carrying (C,T) in the framebuffer: the host walks
// the reference's far->near key list (scene.rs:533-558) in reverse.
// LDS/latency bound, not HBM bound; algorithmic bytes D*40 + W*H*16.
//
// The support decision uses exactly the oracle's operation order (spec §6): explicit fmaf, no contraction.
#include "gsx_internal.h"

namespace gsx {

constexpr int kBatch = 256;

template <int MODE /* 0 splat (gaussian falloff), 1 constant alpha inside the cutoff */>
__global__ __launch_bounds__(256) void k_composite(const FrameConsts f, uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ list,
                                                    const float4* __restrict__ rec_a, const float4* __restrict__ rec_b,
                                                    const float4* __restrict__ rec_c, float4* __restrict__ fb,
                                                    const int carry, uint32_t* __restrict__ done_bits,
                                                    const uint32_t row_words, uint32_t* __restrict__ done_count,
                                                    const int clear_ranges, uint32_t* __restrict__ tile_sat) {
    __shared__ float2 s_mean[kBatch];
    __shared__ uint32_t s_sat;
    __shared__ float4 s_conic[kBatch];
    __shared__ float4 s_rgb[kBatch];

    const uint32_t tile = blockIdx.x;
    const uint32_t tx = tile % f.tiles_x, ty = tile / f.tiles_x;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tx * kTile + (tid & 15u), py = ty * kTile + (tid >> 4);
    const bool inside = px < f.w_px && py < f.h_px;
    const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
    const uint2 range = ranges[tile];
    const size_t fbo = (size_t)py * f.w_px + px;
    // Progressive mode leaves the range table clean for the next slab (saves a memset per slab).  EVERY lane reads
    // ranges[tile] itself, so the entry may only be zeroed once all four waves have read it: on the early-return path
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
    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    if (carry && inside) {
        const float4 p = fb[fbo];
        C0 = p.x; C1 = p.y; C2 = p.z; T = p.w;
    }
    bool done = !inside || T < f.t_eps;
    uint32_t stop_key = 0;  // depth key of the splat that saturated this pixel (in this launch)
    if (tid == 0) s_sat = 0;

    // software pipeline: the gather of batch b+1 (list -> three record planes, dependent random loads) is
    // in flight while batch b is blended out of LDS
    float4 pa = make_float4(0, 0, 0, 0), pb = pa, pc4 = pa;
    if (range.x + tid < range.y) {
        const uint32_t idx = list[range.x + tid];
        pa = rec_a[idx];
        pb = rec_b[idx];
        pc4 = rec_c[idx];
    }
    for (uint32_t base = range.x; base < range.y; base += kBatch) {
        // vote + barrier: also protects the LDS batch of the previous iteration
        if (__syncthreads_and(done)) break;
        s_mean[tid] = make_float2(pa.x, pa.y);
        s_conic[tid] = pb;
        s_rgb[tid] = pc4;
        __syncthreads();
        const uint32_t nxt = base + kBatch + tid;
        if (nxt < range.y) {
            const uint32_t idx = list[nxt];
            pa = rec_a[idx];
            pb = rec_b[idx];
            pc4 = rec_c[idx];
        }
        const uint32_t cnt = min((uint32_t)kBatch, range.y - base);
        // Straight-line body, one predicated region per splat: a saturated pixel simply stops hitting (`done` is part of
        // the support predicate) instead of leaving the loop, which keeps the wave's control flow to one skip branch per
        // splat; a wave whose pixels are all saturated leaves the batch (checked every 4 splats).
        for (uint32_t j0 = 0; j0 < cnt; j0 += 4) {
            if (!__ballot(!done)) break;
            const uint32_t j1 = min(j0 + 4u, cnt);
            for (uint32_t j = j0; j < j1; ++j) {
                const float2 m = s_mean[j];
                const float4 co = s_conic[j];
                const float dx = pxf - m.x, dy = pyf - m.y;
                const float q = fmaf(co.x * dx, dx, fmaf(co.z * dy, dy, ((2.0f * co.y) * dx) * dy));
                if (!done && q <= f.k2 && q >= 0.0f) {
                    const float w = MODE == 0 ? __expf(-0.5f * q) : 1.0f;
                    const float alpha = fminf(f.alpha_max, co.w * w);
                    if (!(alpha < f.alpha_min)) {
                        const float4 c = s_rgb[j];
                        const float wgt = T * alpha;
                        C0 = fmaf(wgt, c.x, C0);
                        C1 = fmaf(wgt, c.y, C1);
                        C2 = fmaf(wgt, c.z, C2);
                        T = T * (1.0f - alpha);
                        if (T < f.t_eps) {
                            done = true;
                            stop_key = __float_as_uint(c.w);
                        }
                    }
                }
            }
        }
    }
    if (inside) fb[fbo] = make_float4(C0, C1, C2, T);
    if (clear_ranges && tid == 0 && had_entries) ranges[tile] = make_uint2(0u, 0u);
    if (done_bits && __syncthreads_and(done)) {
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
                            uint32_t* d_done_count, bool clear_ranges, uint32_t* tile_sat) {
    dim3 grid(f.tiles_x * f.tiles_y), block(256);
    if (f.display_mode == GSX_DISPLAY_SPLAT)
        hipLaunchKernelGGL(k_composite<0>, grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat);
    else
        hipLaunchKernelGGL(k_composite<1>, grid, block, 0, s, f, ranges, list, rec.a, rec.b, rec.c, fb, carry ? 1 : 0, done, row_words, d_done_count,
                           clear_ranges ? 1 : 0, tile_sat);
    return hipGetLastError();
}

hipError_t launch_clear_fb(hipStream_t s, float4* fb, uint32_t n_px) {
    if (!n_px) return hipSuccess;
    hipLaunchKernelGGL(k_clear_fb, dim3((n_px + 255) / 256), dim3(256), 0, s, fb, n_px);
    return hipGetLastError();
}

hipError_t launch_resolve_rgba8(hipStream_t s, const float4* fb, uint32_t n_px, float bg_r, float bg_g, float bg_b,
                                uint32_t* out_rgba8) {
    if (!n_px) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_rgba8, dim3((n_px + 255) / 256), dim3(256), 0, s, fb, n_px, bg_r, bg_g, bg_b, out_rgba8);
    return hipGetLastError();
}

}  // namespace gsx
