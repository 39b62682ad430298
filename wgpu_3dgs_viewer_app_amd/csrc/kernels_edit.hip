// kernels_edit.hip — selection, per-Gaussian edits and queries around the projection pass (gfx950).
//
// Reference call sites: viewer.update_query / update_selection_highlight[_with_pod] / update_selection_edit_with_pod
// (src/tab/scene.rs:785-835), postprocessor.postprocess (scene.rs:601-611), gs::query::download (scene.rs:651-657),
// the per-Gaussian edit buffer (scene.rs:1816-1830, app.rs:769-816).  The arithmetic is the build's
// (spec/RENDER_SPEC.md §7, edit_math.h).  These passes run only while a selection, an edit or a query exists, so the
// projection kernel itself stays untouched: HIDDEN edits are folded into the keep-bitset the projection pass
// already honours (the mask), colour ops and the highlight rewrite the projected records afterwards.
// All bitsets are one bit per Gaussian; a 64-lane wave owns exactly two words, so no atomics are needed.
#include "edit_math.h"
#include <algorithm>

#include "gsx_internal.h"

namespace gsx {

__device__ inline gsx_gaussian_edit load_edit(const float4* ea, const float4* eb, uint32_t i) {
    const float4 a = ea[i], b = eb[i];
    gsx_gaussian_edit e;
    e.flag = __float_as_uint(a.x);
    e.color[0] = a.y; e.color[1] = a.z; e.color[2] = a.w;
    e.contrast = b.x; e.exposure = b.y; e.gamma = b.z; e.alpha = b.w;
    return e;
}

__device__ inline void store_wave_bits(uint32_t* words, uint32_t i, uint32_t n, bool bit) {
    const unsigned long long bal = __ballot(bit);
    const uint32_t lane = threadIdx.x & 63u;
    if ((lane & 31u) == 0 && i < n) words[i >> 5] = (uint32_t)(bal >> lane);
}

// K1, before the projection: persist the selection edit into the selected Gaussians' edit records, and derive the
// keep-bitset = mask & ~hidden the projection pass culls with.
__global__ __launch_bounds__(256) void k_edit_prepare(uint32_t n, const uint32_t* __restrict__ selection,
                                                       uint32_t* __restrict__ edited, float4* __restrict__ edit_a,
                                                       float4* __restrict__ edit_b, gsx_gaussian_edit sel_edit,
                                                       const uint32_t* __restrict__ mask, uint32_t* __restrict__ keep) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool in = i < n;
    const bool sel = in && selection && ((selection[i >> 5] >> (i & 31u)) & 1u);
    bool has = in && ((edited[i >> 5] >> (i & 31u)) & 1u);
    uint32_t flag = 0;
    if (sel && (sel_edit.flag & GSX_EDIT_ENABLED)) {
        edit_a[i] = make_float4(__uint_as_float(sel_edit.flag), sel_edit.color[0], sel_edit.color[1], sel_edit.color[2]);
        edit_b[i] = make_float4(sel_edit.contrast, sel_edit.exposure, sel_edit.gamma, sel_edit.alpha);
        flag = sel_edit.flag;
        has = true;
    } else if (has) {
        flag = __float_as_uint(edit_a[i].x);
    }
    const bool hidden = (flag & GSX_EDIT_ENABLED) && (flag & GSX_EDIT_HIDDEN);
    const bool kept = in && !hidden && (!mask || ((mask[i >> 5] >> (i & 31u)) & 1u));
    __syncthreads();  // every read of `edited` above precedes the rewrite below
    store_wave_bits(edited, i, n, has);
    store_wave_bits(keep, i, n, kept);
}

// K3 colour ops, after the projection: edit the colour / opacity of the surviving Gaussians that carry an edit, then
// blend the selection highlight over the selected ones.
__global__ __launch_bounds__(256) void k_edit_apply(uint32_t n, const uint32_t* __restrict__ key, float4* __restrict__ rec_b,
                                                     float4* __restrict__ rec_c, const uint32_t* __restrict__ selection,
                                                     const uint32_t* __restrict__ edited, const float4* __restrict__ edit_a,
                                                     const float4* __restrict__ edit_b, float4 highlight) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || key[i] == kCulledKey) return;
    const bool sel = selection && highlight.w > 0.0f && ((selection[i >> 5] >> (i & 31u)) & 1u);
    const bool has = edited && ((edited[i >> 5] >> (i & 31u)) & 1u);
    if (!sel && !has) return;
    float4 c = rec_c[i];
    if (has) {
        const gsx_gaussian_edit e = load_edit(edit_a, edit_b, i);
        if (e.flag & GSX_EDIT_ENABLED) {
            float4 b = rec_b[i];
            em_apply_edit(e, c.x, c.y, c.z, b.w);
            rec_b[i] = b;
        }
    }
    if (sel) {
        c.x = c.x + (highlight.x - c.x) * highlight.w;
        c.y = c.y + (highlight.y - c.y) * highlight.w;
        c.z = c.z + (highlight.z - c.z) * highlight.w;
    }
    rec_c[i] = c;
}

// The same ops over a list of (key, index) pairs: the records k_shade has just written on a lazily projected frame (skip = the
// ballots of the records an earlier round shaded — and edited — already: the same bits k_shade skips).
__global__ __launch_bounds__(256) void k_edit_apply_list(const uint2* __restrict__ pairs, const uint32_t* __restrict__ d_n,
                                                          const unsigned long long* __restrict__ skip, float4* __restrict__ rec_b,
                                                          float4* __restrict__ rec_c, const uint32_t* __restrict__ selection,
                                                          const uint32_t* __restrict__ edited, const float4* __restrict__ edit_a,
                                                          const float4* __restrict__ edit_b, float4 highlight) {
    const uint32_t count = *d_n;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < count; j += gridDim.x * 256u) {
        const uint32_t i = pairs[j].y;
        if (skip && ((skip[i >> 6] >> (i & 63u)) & 1ull)) continue;
        const bool sel = selection && highlight.w > 0.0f && ((selection[i >> 5] >> (i & 31u)) & 1u);
        const bool has = edited && ((edited[i >> 5] >> (i & 31u)) & 1u);
        if (!sel && !has) continue;
        float4 c = rec_c[i];
        if (has) {
            const gsx_gaussian_edit e = load_edit(edit_a, edit_b, i);
            if (e.flag & GSX_EDIT_ENABLED) {
                float4 b = rec_b[i];
                em_apply_edit(e, c.x, c.y, c.z, b.w);
                rec_b[i] = b;
            }
        }
        if (sel) {
            c.x = c.x + (highlight.x - c.x) * highlight.w;
            c.y = c.y + (highlight.y - c.y) * highlight.w;
            c.z = c.z + (highlight.z - c.z) * highlight.w;
        }
        rec_c[i] = c;
    }
}

// K1 query: Rect / Brush / Texture set one flag bit per Gaussian; Hit appends (index, depth, alpha) results.
__global__ __launch_bounds__(256) void k_query(uint32_t n, const uint32_t* __restrict__ key, const float4* __restrict__ rec_a,
                                                const float4* __restrict__ rec_b, const float4* __restrict__ rec_c,
                                                gsx_query q, const uint8_t* __restrict__ texture, uint32_t tex_w,
                                                uint32_t tex_h, float k2, float alpha_max, uint32_t display_mode,
                                                uint32_t* __restrict__ flags, gsx_query_hit* __restrict__ hits,
                                                uint32_t* __restrict__ hit_count, uint32_t hit_capacity) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const bool vis = i < n && key[i] != kCulledKey;
    bool flag = false;
    if (vis) {
        const float4 a = rec_a[i];
        if (q.kind == GSX_QUERY_RECT) {
            flag = em_in_rect(a.x, a.y, q);
        } else if (q.kind == GSX_QUERY_BRUSH) {
            flag = em_in_brush(a.x, a.y, q);
        } else if (q.kind == GSX_QUERY_TEXTURE) {
            const float fx = floorf(a.x), fy = floorf(a.y);
            if (texture && fx >= 0.0f && fy >= 0.0f && fx < (float)tex_w && fy < (float)tex_h)
                flag = texture[(size_t)fy * tex_w + (size_t)fx] != 0;
        } else if (q.kind == GSX_QUERY_HIT) {
            const float4 co = rec_b[i];
            const float dx = q.p0[0] - a.x, dy = q.p0[1] - a.y;
            const float qq = fmaf(co.x * dx, dx, fmaf(co.z * dy, dy, ((2.0f * co.y) * dx) * dy));
            if (qq <= k2 && qq >= 0.0f) {
                const float w = display_mode == GSX_DISPLAY_SPLAT ? __expf(-0.5f * qq) : 1.0f;
                const float alpha = fminf(alpha_max, co.w * w);
                if (alpha >= (1.0f / 255.0f)) {
                    const uint32_t slot = atomicAdd(hit_count, 1u);
                    if (slot < hit_capacity) hits[slot] = gsx_query_hit{i, rec_c[i].w, alpha, 0u};
                }
            }
        }
    }
    if (q.kind != GSX_QUERY_HIT) store_wave_bits(flags, i, n, flag);
}

// K4: selection = op(selection, flags), one word per lane
__global__ __launch_bounds__(256) void k_selection_op(uint32_t n_words, uint32_t op, const uint32_t* __restrict__ flags,
                                                       uint32_t* __restrict__ selection) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t f = flags[w], s = selection[w];
    selection[w] = op == GSX_SELECTION_SET ? f : (op == GSX_SELECTION_ADD ? (s | f) : (s & ~f));
}

hipError_t launch_edit_prepare(hipStream_t s, uint32_t n, const uint32_t* selection, uint32_t* edited, float4* edit_a,
                               float4* edit_b, const gsx_gaussian_edit& sel_edit, const uint32_t* mask, uint32_t* keep) {
    if (n) GSX_LAUNCH(k_edit_prepare, dim3((n + 255) / 256), dim3(256), 0, s, n, selection, edited, edit_a, edit_b, sel_edit, mask, keep);
    return hipGetLastError();
}

hipError_t launch_edit_apply(hipStream_t s, uint32_t n, const Records& rec, const uint32_t* selection, const uint32_t* edited,
                             const float4* edit_a, const float4* edit_b, const float highlight[4]) {
    if (n)
        GSX_LAUNCH(k_edit_apply, dim3((n + 255) / 256), dim3(256), 0, s, n, rec.key, rec.b, rec.c, selection, edited, edit_a,
                           edit_b, make_float4(highlight[0], highlight[1], highlight[2], highlight[3]));
    return hipGetLastError();
}

hipError_t launch_edit_apply_list(hipStream_t s, uint32_t n, const Records& rec, const uint2* pairs, const uint32_t* d_n,
                                  const unsigned long long* skip, const uint32_t* selection, const uint32_t* edited, const float4* edit_a,
                                  const float4* edit_b, const float highlight[4]) {
    if (n)  // (n: an upper bound of the list length; the count is on the device)
        GSX_LAUNCH(k_edit_apply_list, dim3(std::min<uint32_t>((n + 255) / 256, 2048u)), dim3(256), 0, s, pairs, d_n, skip, rec.b, rec.c,
                           selection, edited, edit_a, edit_b, make_float4(highlight[0], highlight[1], highlight[2], highlight[3]));
    return hipGetLastError();
}

hipError_t launch_query(hipStream_t s, uint32_t n, const Records& rec, const gsx_query& q, const uint8_t* texture, uint32_t tex_w,
                        uint32_t tex_h, const FrameConsts& f, uint32_t* flags, gsx_query_hit* hits, uint32_t* hit_count,
                        uint32_t hit_capacity) {
    if (n)
        GSX_LAUNCH(k_query, dim3((n + 255) / 256), dim3(256), 0, s, n, rec.key, rec.a, rec.b, rec.c, q, texture, tex_w, tex_h,
                           f.k2, f.alpha_max, f.display_mode, flags, hits, hit_count, hit_capacity);
    return hipGetLastError();
}

hipError_t launch_selection_op(hipStream_t s, uint32_t n_words, uint32_t op, const uint32_t* flags, uint32_t* selection) {
    if (n_words) GSX_LAUNCH(k_selection_op, dim3((n_words + 255) / 256), dim3(256), 0, s, n_words, op, flags, selection);
    return hipGetLastError();
}

}  // namespace gsx
