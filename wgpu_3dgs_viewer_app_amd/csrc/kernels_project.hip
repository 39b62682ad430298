// kernels_project.hip — upload conversion (Gaussian -> pod planes) and the projection pass for gfx950.
//
// Projection pass = the reference's K1 (`preprocessor.preprocess`, src/tab/scene.rs:856-863: cull +
// depth key) fused with the per-instance half of K3 (`renderer.render_with_pass`, scene.rs:2306-2313:
// SH colour + 3D->2D covariance), so the 220-byte pod is streamed from HBM exactly once per frame.
// Layout: every attribute is a contiguous plane of float4 (or float2 / float) over the model, so each
// wave64 load instruction moves 64 x 16 B = 1 KiB fully coalesced; 15 loads per Gaussian.
// HBM-bound: algorithmic bytes N*220 in + N_vis*40 out.  No MFMA (no contraction).
//
// Built with -ffp-contract=off: every operation that feeds an integer decision (cull, tile rectangle)
// is written in the exact order of spec/RENDER_SPEC.md §4 so the CPU oracle reproduces it bit-for-bit.
#include <hip/hip_fp16.h>

#include "edit_math.h"
#include "gsx_internal.h"
#include "project_math.h"
#include "window_scan.h"

namespace gsx {

// ------------------------------------------------------------------------------------------------
// host: frame constants (spec §3)
// ------------------------------------------------------------------------------------------------
void quat_to_rows(const float q[4], float r[9]);
static void quat_rows(const float q[4], float r[9]) { quat_to_rows(q, r); }
void quat_to_rows(const float q[4], float r[9]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    r[0] = 1.0f - (yy + zz); r[1] = xy - wz;          r[2] = xz + wy;
    r[3] = xy + wz;          r[4] = 1.0f - (xx + zz); r[5] = yz - wx;
    r[6] = xz - wy;          r[7] = yz + wx;          r[8] = 1.0f - (xx + yy);
}

static inline float hdot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

void frame_consts_setup(const float view[16], const float proj[16], uint32_t width, uint32_t height,
                        const ModelTransform& mt, float size, uint32_t display_mode, uint32_t sh_deg, uint32_t no_sh0,
                        const gsx_spec_params& sp, FrameConsts* f) {
    f->band_lo = 0;
    f->band_hi = (height + GSX_TILE - 1) / GSX_TILE;
    float R[9], W[9], WR[9];
    quat_rows(mt.quat, R);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W[r * 3 + c] = view[c * 4 + r];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            WR[r * 3 + c] = hdot3(W[r * 3 + 0], W[r * 3 + 1], W[r * 3 + 2], R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) f->T[r * 3 + c] = WR[r * 3 + c] * mt.scale[c];
    for (int r = 0; r < 3; ++r)
        f->vt[r] = hdot3(W[r * 3 + 0], W[r * 3 + 1], W[r * 3 + 2], mt.pos[0], mt.pos[1], mt.pos[2]) + view[12 + r];
    for (int i = 0; i < 16; ++i) f->P[i] = proj[i];
    float cam[3];
    for (int c = 0; c < 3; ++c) cam[c] = -hdot3(W[0 * 3 + c], W[1 * 3 + c], W[2 * 3 + c], view[12], view[13], view[14]);
    float rel[3] = {cam[0] - mt.pos[0], cam[1] - mt.pos[1], cam[2] - mt.pos[2]};
    for (int c = 0; c < 3; ++c) f->cam_m[c] = hdot3(R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c], rel[0], rel[1], rel[2]);
    for (int c = 0; c < 3; ++c) f->s_m[c] = mt.scale[c];
    f->width = (float)width;
    f->height = (float)height;
    f->fx = proj[0] * f->width * 0.5f;
    f->fy = proj[5] * f->height * 0.5f;
    f->limx = sp.jacobian_clamp / proj[0];
    f->limy = sp.jacobian_clamp / proj[5];
    f->size2 = size * size;
    f->k = sp.max_std_dev;
    f->k2 = sp.max_std_dev * sp.max_std_dev;
    f->low_pass = sp.low_pass;
    f->cull_margin = sp.cull_margin;
    f->alpha_max = sp.alpha_max;
    f->alpha_min = sp.alpha_min;
    f->point_radius = sp.point_radius;
    f->t_eps = sp.t_epsilon;
    f->w_px = width;
    f->h_px = height;
    f->tiles_x = (width + kTile - 1) / kTile;
    f->tiles_y = (height + kTile - 1) / kTile;
    f->sh_deg = sh_deg;
    f->no_sh0 = no_sh0;
    f->display_mode = display_mode;
}

// ------------------------------------------------------------------------------------------------
// upload: gs::Gaussian (AoS, 224 B) -> pod planes.  `gaussians_buffer.update_range`, scene.rs:2083-2084.
// Load-time only; one thread per Gaussian.
// ------------------------------------------------------------------------------------------------
__device__ inline float ddot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// ---- quantisation (spec/RENDER_SPEC.md §2b) ----
// f16: IEEE binary16, round to nearest even (what `half::f16::from_f32` does; v_cvt_f16_f32).
// snorm8: q = floor(clamp(v,-1,1) * 127 + 0.5) as int8; decode = max(q / 127, -1) (WGSL unpack4x8snorm).
__device__ inline uint32_t pack_h2(float a, float b) {
    return (uint32_t)__half_as_ushort(__float2half_rn(a)) | ((uint32_t)__half_as_ushort(__float2half_rn(b)) << 16);
}
__device__ inline float h_lo(uint32_t u) { return __half2float(__ushort_as_half((unsigned short)(u & 0xFFFFu))); }
__device__ inline float h_hi(uint32_t u) { return __half2float(__ushort_as_half((unsigned short)(u >> 16))); }
__device__ inline uint32_t q_snorm8(float v) {
    float c = fminf(fmaxf(v, -1.0f), 1.0f);
    return (uint32_t)(int)floorf(c * 127.0f + 0.5f) & 0xFFu;
}
__device__ inline float dq_snorm8(uint32_t word, int byte) {
    int q = (int)(signed char)((word >> (8 * byte)) & 0xFFu);
    return fmaxf((float)q * (1.0f / 127.0f), -1.0f);
}

__device__ inline void store_sh(const PodPlanes& pod, uint64_t model_n, uint64_t i, const float* s45) {
    if (pod.sh_kind == GSX_SH_SINGLE) {
#pragma unroll
        for (int p = 0; p < kShPlanes4; ++p) {
            const float4 v = make_float4(s45[4 * p], s45[4 * p + 1], s45[4 * p + 2], s45[4 * p + 3]);
            pod.sh4[(uint64_t)p * model_n + i] = v;
            if (pod.sh_aos) pod.sh_aos[i * pod.aos_stride + p] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
        }
        pod.sh1[i] = s45[44];
        if (pod.sh_aos) pod.sh_aos[i * pod.aos_stride + 11] = make_uint4(__float_as_uint(s45[44]), 0u, 0u, 0u);
    } else if (pod.sh_kind == GSX_SH_HALF) {
        for (int p = 0; p < 6; ++p) {
            uint32_t w[4];
            for (int k = 0; k < 4; ++k) {
                int f0 = 8 * p + 2 * k, f1 = f0 + 1;
                w[k] = pack_h2(f0 < 45 ? s45[f0] : 0.0f, f1 < 45 ? s45[f1] : 0.0f);
            }
            pod.sh_h[(uint64_t)p * model_n + i] = make_uint4(w[0], w[1], w[2], w[3]);
            if (pod.sh_aos) pod.sh_aos[i * pod.aos_stride + p] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else if (pod.sh_kind == GSX_SH_NORM8) {
        for (int p = 0; p < 3; ++p) {
            uint32_t w[4];
            for (int k = 0; k < 4; ++k) {
                uint32_t v = 0;
                for (int b = 0; b < 4; ++b) {
                    int f = 16 * p + 4 * k + b;
                    v |= (f < 45 ? q_snorm8(s45[f]) : 0u) << (8 * b);
                }
                w[k] = v;
            }
            pod.sh_q[(uint64_t)p * model_n + i] = make_uint4(w[0], w[1], w[2], w[3]);
            if (pod.sh_aos) pod.sh_aos[i * pod.aos_stride + p] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// the shade record's geometry words: position + colour word and the covariance (as the pod stores it) beside the SH words
__device__ inline void store_aos_geometry(const PodPlanes& pod, uint64_t i, float4 pc, float c0, float c1, float c2, float c3, float c4,
                                          float c5) {
    if (!pod.sh_aos || !pod.aos_geo) return;
    uint4* r = pod.sh_aos + i * pod.aos_stride + pod.aos_geo;
    r[0] = make_uint4(__float_as_uint(pc.x), __float_as_uint(pc.y), __float_as_uint(pc.z), __float_as_uint(pc.w));
    if (pod.cov_kind == GSX_COV3D_SINGLE) {
        r[1] = make_uint4(__float_as_uint(c0), __float_as_uint(c1), __float_as_uint(c2), __float_as_uint(c3));
        r[2] = make_uint4(__float_as_uint(c4), __float_as_uint(c5), 0u, 0u);
    } else {
        r[1] = make_uint4(pack_h2(c0, c1), pack_h2(c2, c3), pack_h2(c4, c5), 0u);  // (the same halves store_cov writes)
    }
}

__device__ inline void store_cov(const PodPlanes& pod, uint64_t i, float c0, float c1, float c2, float c3, float c4, float c5) {
    if (pod.cov_kind == GSX_COV3D_SINGLE) {
        pod.cov_a[i] = make_float4(c0, c1, c2, c3);
        pod.cov_b[i] = make_float2(c4, c5);
    } else {
        pod.cov_h[i] = make_uint2(pack_h2(c0, c1), pack_h2(c2, c3));
        pod.cov_h2[i] = pack_h2(c4, c5);
    }
}

// dequantised SH floats [0, 48) of Gaussian i (zero when the pod has no SH)
__device__ inline void load_sh_dq(const PodPlanes& pod, uint64_t model_n, uint64_t i, float* s48) {
    for (int f = 0; f < 48; ++f) s48[f] = 0.0f;
    if (pod.sh_kind == GSX_SH_SINGLE) {
        for (int p = 0; p < kShPlanes4; ++p) {
            float4 v = pod.sh4[(uint64_t)p * model_n + i];
            s48[4 * p] = v.x; s48[4 * p + 1] = v.y; s48[4 * p + 2] = v.z; s48[4 * p + 3] = v.w;
        }
        s48[44] = pod.sh1[i];
    } else if (pod.sh_kind == GSX_SH_HALF) {
        for (int p = 0; p < 6; ++p) {
            uint4 v = pod.sh_h[(uint64_t)p * model_n + i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 4; ++k) {
                s48[8 * p + 2 * k] = h_lo(w[k]);
                s48[8 * p + 2 * k + 1] = h_hi(w[k]);
            }
        }
    } else if (pod.sh_kind == GSX_SH_NORM8) {
        for (int p = 0; p < 3; ++p) {
            uint4 v = pod.sh_q[(uint64_t)p * model_n + i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 4; ++k)
                for (int b = 0; b < 4; ++b) s48[16 * p + 4 * k + b] = dq_snorm8(w[k], b);
        }
    }
}

__device__ inline void load_cov_dq(const PodPlanes& pod, uint64_t i, float* c6) {
    if (pod.cov_kind == GSX_COV3D_SINGLE) {
        float4 a = pod.cov_a[i];
        float2 b = pod.cov_b[i];
        c6[0] = a.x; c6[1] = a.y; c6[2] = a.z; c6[3] = a.w; c6[4] = b.x; c6[5] = b.y;
    } else {
        uint2 a = pod.cov_h[i];
        uint32_t b = pod.cov_h2[i];
        c6[0] = h_lo(a.x); c6[1] = h_hi(a.x); c6[2] = h_lo(a.y); c6[3] = h_hi(a.y); c6[4] = h_lo(b); c6[5] = h_hi(b);
    }
}

__global__ __launch_bounds__(256) void k_convert(const gsx_gaussian* __restrict__ src, uint64_t n, uint64_t start,
                                                  uint64_t model_n, PodPlanes pod) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const gsx_gaussian& g = src[t];
    uint64_t i = start + t;
    float x = g.rot[0], y = g.rot[1], z = g.rot[2], w = g.rot[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    float R[9] = {1.0f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0f - (xx + zz), yz - wx, xz - wy, yz + wx,
                  1.0f - (xx + yy)};
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M[r * 3 + c] = R[r * 3 + c] * g.scale[c];
    float c0 = ddot3(M[0], M[1], M[2], M[0], M[1], M[2]);
    float c1 = ddot3(M[0], M[1], M[2], M[3], M[4], M[5]);
    float c2 = ddot3(M[0], M[1], M[2], M[6], M[7], M[8]);
    float c3 = ddot3(M[3], M[4], M[5], M[3], M[4], M[5]);
    float c4 = ddot3(M[3], M[4], M[5], M[6], M[7], M[8]);
    float c5 = ddot3(M[6], M[7], M[8], M[6], M[7], M[8]);
    uint32_t col = (uint32_t)g.color[0] | ((uint32_t)g.color[1] << 8) | ((uint32_t)g.color[2] << 16) |
                   ((uint32_t)g.color[3] << 24);
    pod.pc[i] = make_float4(g.pos[0], g.pos[1], g.pos[2], __uint_as_float(col));
    store_cov(pod, i, c0, c1, c2, c3, c4, c5);
    store_aos_geometry(pod, i, make_float4(g.pos[0], g.pos[1], g.pos[2], __uint_as_float(col)), c0, c1, c2, c3, c4, c5);
    store_sh(pod, model_n, i, &g.sh[0][0]);
}

// pod-ready planes (pos 3n, color n, sh 45n, cov 6n) -> resident float4 planes, and back (parity readback).
__global__ __launch_bounds__(256) void k_pack_pod(const float* __restrict__ pos, const uint32_t* __restrict__ color,
                                                   const float* __restrict__ sh, const float* __restrict__ cov,
                                                   uint64_t n, uint64_t start, uint64_t model_n, PodPlanes pod) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t i = start + t;
    pod.pc[i] = make_float4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], __uint_as_float(color[t]));
    store_cov(pod, i, cov[6 * t], cov[6 * t + 1], cov[6 * t + 2], cov[6 * t + 3], cov[6 * t + 4], cov[6 * t + 5]);
    store_aos_geometry(pod, i, make_float4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], __uint_as_float(color[t])), cov[6 * t], cov[6 * t + 1],
                       cov[6 * t + 2], cov[6 * t + 3], cov[6 * t + 4], cov[6 * t + 5]);
    if (sh) store_sh(pod, model_n, i, sh + 45 * t);
}

// parity readback: the pod as the kernels see it, i.e. AFTER dequantisation
__global__ __launch_bounds__(256) void k_unpack_pod(PodPlanes pod, uint64_t model_n, float* __restrict__ pos,
                                                     uint32_t* __restrict__ color, float* __restrict__ sh,
                                                     float* __restrict__ cov) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= model_n) return;
    float4 pc = pod.pc[i];
    pos[3 * i] = pc.x; pos[3 * i + 1] = pc.y; pos[3 * i + 2] = pc.z;
    color[i] = __float_as_uint(pc.w);
    float c6[6];
    load_cov_dq(pod, i, c6);
    for (int k = 0; k < 6; ++k) cov[6 * i + k] = c6[k];
    if (sh) {
        float s48[48];
        load_sh_dq(pod, model_n, i, s48);
        for (int k = 0; k < 45; ++k) sh[45 * i + k] = s48[k];
    }
}

// ------------------------------------------------------------------------------------------------
// projection pass.  One Gaussian per lane, 256 lanes per workgroup; N/256 workgroups (>> 256 CUs).
// DEG = SH degree evaluated (0 = DC only, no SH planes touched).
// Load schedule chosen by measurement (tools/bench_project.hip, 10 M Gaussians on MI355X): SoA planes
// with survivor-only dependent loads reach 4.8 TB/s algorithmic; a 256-Gaussian chunked AoSoA layout
// was slower (4.0 TB/s), and eager "all loads first" was no faster.  The visible count is reduced per
// workgroup: one same-address atomic per wave serialises at ~12 ns each and alone cost 1.8 ms at 10 M.
// ------------------------------------------------------------------------------------------------
// Streaming (read-once) loads of the pod planes: non-temporal, so 2.2 GB of input does not churn L2 / the
// Infinity Cache.  Measured within one process on 10 M Gaussians (tools/bench_project.hip): 0.587 ms vs
// 0.61-0.68 ms with default-policy loads, i.e. exactly the float4 streaming-copy time for the same bytes.
typedef float gsx_f4v __attribute__((ext_vector_type(4)));
typedef float gsx_f2v __attribute__((ext_vector_type(2)));
__device__ inline float4 ld_stream(const float4* p) {
    gsx_f4v v = __builtin_nontemporal_load(reinterpret_cast<const gsx_f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline float2 ld_stream(const float2* p) {
    gsx_f2v v = __builtin_nontemporal_load(reinterpret_cast<const gsx_f2v*>(p));
    return make_float2(v.x, v.y);
}
__device__ inline float ld_stream(const float* p) { return __builtin_nontemporal_load(p); }

__device__ inline uint4 ld_stream(const uint4* p) {
    typedef unsigned gsx_u4v __attribute__((ext_vector_type(4)));
    gsx_u4v v = __builtin_nontemporal_load(reinterpret_cast<const gsx_u4v*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint2 ld_stream(const uint2* p) {
    typedef unsigned gsx_u2v __attribute__((ext_vector_type(2)));
    gsx_u2v v = __builtin_nontemporal_load(reinterpret_cast<const gsx_u2v*>(p));
    return make_uint2(v.x, v.y);
}
__device__ inline uint32_t ld_stream(const uint32_t* p) { return __builtin_nontemporal_load(p); }

// Streaming stores of the projection records: written once, 440 MB per frame at 10 M Gaussians, read back sparsely much
// later.  tools/bench_hbm.hip (14 planes read + 3 written, this pass's shape): 0.500 ms with default stores, 0.462 ms with
// non-temporal ones — a write costs this part about twice a read, and write-allocating L2 / Infinity Cache lines for data
// nobody reads soon makes it worse.  And every lane stores: culled Gaussians write a record nobody reads
// (key = kCulledKey marks it), so that a wave always writes whole 128-byte lines — with ~15 % of the lanes masked nearly
// every line was a partial write: k_project<3,0,0> 516 -> 471 us on cfg4 (A/B on one box), at 8 % more bytes written.
__device__ inline void st_stream(float4* p, float4 v) {
    __builtin_nontemporal_store(gsx_f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<gsx_f4v*>(p));
}
__device__ inline void st_stream(uint32_t* p, uint32_t v) {
    __builtin_nontemporal_store(v, p);
}

// SHK / COVK: storage of the SH and cov3d planes (gsx_sh_kind / gsx_cov3d_kind); dequantisation is exact
// (f16 -> f32, snorm8 -> f32), so cull set and tile rectangles stay bit-exact against the oracle, which
// projects the dequantised pod.
// Admission (kernels_admit.hip) is decided here, where key and rectangle are still in registers: adm.pyramid.data ==
// nullptr admits every visible Gaussian; otherwise the conservative max-pyramid test of the temporal occlusion
// speculation (window_scan.h).  One ballot word per wave + one count per workgroup feed the compaction.
// covariance planes of Gaussian i -> cov2d, screen position, tile rectangle (false: culled)
template <int COVK>
__device__ inline bool load_cov2d_rect(const FrameConsts& f, const PodPlanes& pod, uint32_t i, const ViewClip& vc, Splat2D& sp) {
    float c0, c1, c2, c3, c4, c5;
    if (COVK == GSX_COV3D_SINGLE) {
        const float4 cva = ld_stream(&pod.cov_a[i]);
        const float2 cvb = ld_stream(&pod.cov_b[i]);
        c0 = cva.x; c1 = cva.y; c2 = cva.z; c3 = cva.w; c4 = cvb.x; c5 = cvb.y;
    } else {
        const uint2 a = ld_stream(&pod.cov_h[i]);
        const uint32_t b = ld_stream(&pod.cov_h2[i]);
        c0 = h_lo(a.x); c1 = h_hi(a.x); c2 = h_lo(a.y); c3 = h_hi(a.y); c4 = h_lo(b); c5 = h_hi(b);
    }
    return pm_cov2d_rect(f, vc, c0, c1, c2, c3, c4, c5, sp);
}

// SH planes of Gaussian i (only the planes the degree needs) -> colour.  AOS = false: the streaming SoA planes (every
// lane of the wave shades, non-temporal loads); AOS = true: the per-Gaussian record copy (sh_aos: the same words, P
// consecutive uint4 per Gaussian), for the sparse shading of admitted Gaussians — whole cache lines are used.
template <int DEG, int SHK, bool AOS>
__device__ inline void load_shade(const FrameConsts& f, const PodPlanes& pod, uint32_t n, uint32_t i, const float4& pc,
                                  float& r, float& g, float& b) {
    constexpr int kFloats = ShNeed<DEG>::floats;
    // every SH float is consumed as it arrives (ShStream, project_math.h): same value as pm_color, 45 fewer live registers
    ShStream<DEG> st;
    st.begin(f, pc.x, pc.y, pc.z, __float_as_uint(pc.w));
    if (SHK == GSX_SH_SINGLE) {
        float4 v[ShNeed<DEG>::planes4 ? ShNeed<DEG>::planes4 : 1];
        float last = 0.0f;
#pragma unroll
        for (int p = 0; p < ShNeed<DEG>::planes4; ++p) {
            if (AOS) {
                const uint4 w = pod.sh_aos[(uint64_t)i * pod.aos_stride + p];
                v[p] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
            } else {
                v[p] = ld_stream(&pod.sh4[(uint64_t)p * n + i]);
            }
        }
        if (DEG == 3) last = AOS ? __uint_as_float(pod.sh_aos[(uint64_t)i * pod.aos_stride + 11].x) : ld_stream(&pod.sh1[i]);
#pragma unroll
        for (int p = 0; p < ShNeed<DEG>::planes4; ++p) {
            st.feed(4 * p, v[p].x); st.feed(4 * p + 1, v[p].y); st.feed(4 * p + 2, v[p].z); st.feed(4 * p + 3, v[p].w);
        }
        if (DEG == 3) st.feed(44, last);
    } else if (SHK == GSX_SH_HALF) {
        constexpr int kP = (kFloats + 7) / 8;
        uint4 v[kP ? kP : 1];
#pragma unroll
        for (int p = 0; p < kP; ++p) v[p] = AOS ? pod.sh_aos[(uint64_t)i * pod.aos_stride + p] : ld_stream(&pod.sh_h[(uint64_t)p * n + i]);
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            st.feed(8 * p, h_lo(v[p].x)); st.feed(8 * p + 1, h_hi(v[p].x)); st.feed(8 * p + 2, h_lo(v[p].y)); st.feed(8 * p + 3, h_hi(v[p].y));
            st.feed(8 * p + 4, h_lo(v[p].z)); st.feed(8 * p + 5, h_hi(v[p].z)); st.feed(8 * p + 6, h_lo(v[p].w)); st.feed(8 * p + 7, h_hi(v[p].w));
        }
    } else if (SHK == GSX_SH_NORM8) {
        constexpr int kP = (kFloats + 15) / 16;
        uint4 v[kP ? kP : 1];
#pragma unroll
        for (int p = 0; p < kP; ++p) v[p] = AOS ? pod.sh_aos[(uint64_t)i * pod.aos_stride + p] : ld_stream(&pod.sh_q[(uint64_t)p * n + i]);
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const uint32_t w[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) st.feed(16 * p + 4 * k + bb, dq_snorm8(w[k], bb));
        }
    }
    st.finish(r, g, b);
}

// (adm.lazy — geometry only, nobody shaded here: the SH planes, 180 of the pod's 220 bytes, are not read and no conic /
// colour record is written — is a kernel of its own, k_project_geom below; k_shade then shades the Gaussians the admission
// let through, a few per cent of the visible ones, and later the few more the repair round turns out to need.)
template <int DEG, int SHK, int COVK>
__global__ __launch_bounds__(256) void k_project(const FrameConsts f, const uint32_t n, const PodPlanes pod,
                                                  const Records rec, uint32_t* __restrict__ block_visible,
                                                  const ProjectAdmission adm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool vis = i < n;
    float4 pc = make_float4(0, 0, 0, 0);
    if (vis) pc = ld_stream(&pod.pc[i]);
    if (vis && pod.mask) vis = (pod.mask[i >> 5] >> (i & 31)) & 1u;
    const uint32_t color = __float_as_uint(pc.w);

    ViewClip vc;
    vis = pm_view_cull(f, pc.x, pc.y, pc.z, vc) && vis;

    Splat2D sp{};
    // the covariance planes are only fetched for Gaussians that survive the frustum test
    if (vis) vis = load_cov2d_rect<COVK>(f, pod, i, vc, sp);
    // screen-band rendering (one band of tile rows per GPU over a replicated scene): what misses the band is culled
    if (vis && ((sp.ry >> 16) <= f.band_lo || (sp.ry & 0xFFFFu) >= f.band_hi)) vis = false;

    bool take = vis;
    if (adm.pyramid.data && vis) take = pyramid_admits(adm.pyramid, __float_as_uint(vc.d), sp.rx, sp.ry);
    const bool shade = vis;

    float r = 0, g = 0, b = 0;
    if (shade) load_shade<DEG, SHK, false>(f, pod, n, i, pc, r, g, b);  // SH planes: survivors only

    if (i < n) {
        st_stream(&rec.key[i], vis ? __float_as_uint(vc.d) : kCulledKey);
        st_stream(&rec.a[i], make_float4(sp.mx, sp.my, __uint_as_float(sp.rx), __uint_as_float(sp.ry)));
        st_stream(&rec.b[i], make_float4(sp.con_a, sp.con_b, sp.con_c, (float)(color >> 24) * (1.0f / 255.0f)));
        st_stream(&rec.c[i], make_float4(r, g, b, vc.d));
    }
    __shared__ uint32_t wave_cnt[4], wave_adm[4];
    const unsigned long long bal = __ballot(vis);
    const unsigned long long bal_adm = __ballot(take);
    if ((threadIdx.x & 63u) == 0) {
        wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
        wave_adm[threadIdx.x >> 6] = (uint32_t)__popcll(bal_adm);
        adm.ballots[blockIdx.x * 4u + (threadIdx.x >> 6)] = bal_adm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_visible[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        adm.block_counts[blockIdx.x] = wave_adm[0] + wave_adm[1] + wave_adm[2] + wave_adm[3];
    }
}

// Geometry-only projection (the lazy variant of k_project: same values, same outputs).  A lane carries PER Gaussians, 256 apart,
// and every load of a stage is issued before any is consumed; the covariance loads do not wait for the cull (a culled lane
// reads its own element: the lines are fetched for its neighbours anyway), key and rectangle are stored before the admission
// test's pyramid cells are waited for.  Round 4, A/B on one box (tools/ab.sh, profiles/r04_ab_geom.txt), cfg4: PER = 1 with
// covariance loads behind the cull 100.0 us; loads ahead of the cull 100.5; + early stores 102; PER = 2: 95.1 / 94.2 / 92.9;
// PER = 4: 101.  The pass moves 480 MB in 93 us (5.2 TB/s) and is NOT waiting for memory: by the ISA a wave of 64 Gaussians
// issues 362 vector and 206 scalar instructions (the 3x3 products, J W S W^T J^T, three correctly rounded divisions, the tile
// rectangle: the spec's operation order, -ffp-contract=off) = 1450 cycles of its SIMD; 152 waves per SIMD at 10 M Gaussians are
// 220 k cycles = 91 us at 2.4 GHz.  The kernel sits on its vector issue rate; HBM would allow ~66 us (tools/bench_hbm).
// (A fixed grid striding over the groups instead of one workgroup per group — an empty 39 K-workgroup launch costs 9 us of
// dispatch — was slower: 131-166 us at 2048-16384 workgroups; the dispatcher streams workgroups better than a loop with a
// barrier per group.)  A workgroup covers PER consecutive 256-Gaussian groups and writes their ballots / counts exactly where
// PER workgroups of k_project would.
constexpr int kProjGeomPer = 2;  // Gaussians per lane of the geometry-only kernel
// QUERY: also answer adm.query (rect / brush / texture) from the projected centre: the flag words k_query would write
template <int COVK, int PER, bool QUERY>
__global__ __launch_bounds__(256) void k_project_geom(const FrameConsts f, const uint32_t n, const PodPlanes pod,
                                                       const Records rec, uint32_t* __restrict__ block_visible,
                                                       const ProjectAdmission adm) {
    const uint32_t base = blockIdx.x * (256u * PER) + threadIdx.x;
    float4 pc[PER];
    bool vis[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = base + 256u * k;
        vis[k] = i < n;
        pc[k] = ld_stream(&pod.pc[vis[k] ? i : 0u]);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = base + 256u * k;
        if (vis[k] && pod.mask) vis[k] = (pod.mask[i >> 5] >> (i & 31)) & 1u;
    }
    ViewClip vc[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) vis[k] = pm_view_cull(f, pc[k].x, pc[k].y, pc[k].z, vc[k]) && vis[k];
    // covariance: every load first (culled lanes read element 0: one line per wave, no branch around the load)
    float cv[PER][6];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = min(base + 256u * k, n - 1u);   // its own element whatever the cull says: the address does not wait for the position
        if (COVK == GSX_COV3D_SINGLE) {
            const float4 a = ld_stream(&pod.cov_a[i]);
            const float2 b = ld_stream(&pod.cov_b[i]);
            cv[k][0] = a.x; cv[k][1] = a.y; cv[k][2] = a.z; cv[k][3] = a.w; cv[k][4] = b.x; cv[k][5] = b.y;
        } else {
            const uint2 a = ld_stream(&pod.cov_h[i]);
            const uint32_t b = ld_stream(&pod.cov_h2[i]);
            cv[k][0] = h_lo(a.x); cv[k][1] = h_hi(a.x); cv[k][2] = h_lo(a.y); cv[k][3] = h_hi(a.y); cv[k][4] = h_lo(b); cv[k][5] = h_hi(b);
        }
    }
    Splat2D sp[PER];
    bool take[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        sp[k] = Splat2D{};
        if (vis[k]) vis[k] = pm_cov2d_rect(f, vc[k], cv[k][0], cv[k][1], cv[k][2], cv[k][3], cv[k][4], cv[k][5], sp[k]);
        if (vis[k] && ((sp[k].ry >> 16) <= f.band_lo || (sp[k].ry & 0xFFFFu) >= f.band_hi)) vis[k] = false;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {  // key and rectangle leave before the admission test's loads are waited for
        const uint32_t i = base + 256u * k;
        if (i < n) {
            st_stream(&rec.key[i], vis[k] ? __float_as_uint(vc[k].d) : kCulledKey);
            if (rec.rect8) {
                const uint32_t rx = sp[k].rx, ry = sp[k].ry;
                st_stream(&rec.rect8[i], vis[k] ? ((rx & 0xFFu) | ((ry & 0xFFu) << 8) | ((rx >> 16) << 16) | ((ry >> 16) << 24)) : 0u);
            } else {
                st_stream(&rec.a[i], make_float4(sp[k].mx, sp[k].my, __uint_as_float(sp[k].rx), __uint_as_float(sp[k].ry)));
            }
            if (rec.code8) {   // the coarse cells of the rectangle (slab shading: later depth slabs refuse by them, k_block_bin)
                const uint32_t code = coarse_code(sp[k].rx, sp[k].ry, f.tiles_x, f.tiles_y);
                rec.code8[i] = vis[k] ? (uint8_t)code : (uint8_t)0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        take[k] = vis[k];
        if (adm.pyramid.data && vis[k]) take[k] = pyramid_admits(adm.pyramid, __float_as_uint(vc[k].d), sp[k].rx, sp[k].ry);
    }
    if (QUERY) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const uint32_t i = base + 256u * k;
            const gsx_query& q = adm.query.q;
            bool flag = false;
            if (vis[k]) {
                const float mx = sp[k].mx, my = sp[k].my;
                if (q.kind == GSX_QUERY_RECT) {
                    flag = em_in_rect(mx, my, q);
                } else if (q.kind == GSX_QUERY_BRUSH) {
                    flag = em_in_brush(mx, my, q);
                } else {  // GSX_QUERY_TEXTURE
                    const float fx = floorf(mx), fy = floorf(my);
                    if (adm.query.texture && fx >= 0.0f && fy >= 0.0f && fx < (float)adm.query.tex_w && fy < (float)adm.query.tex_h)
                        flag = adm.query.texture[(size_t)fy * adm.query.tex_w + (size_t)fx] != 0;
                }
            }
            const unsigned long long qb = __ballot(flag);
            const uint32_t lane = threadIdx.x & 63u;
            if ((lane & 31u) == 0 && i < n) adm.query.flags[i >> 5] = (uint32_t)(qb >> lane);
        }
    }
    __shared__ uint32_t wave_cnt[PER][4], wave_adm[PER][4];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const unsigned long long bal = __ballot(vis[k]);
        const unsigned long long bal_adm = __ballot(take[k]);
        if ((threadIdx.x & 63u) == 0) {
            wave_cnt[k][threadIdx.x >> 6] = (uint32_t)__popcll(bal);
            wave_adm[k][threadIdx.x >> 6] = (uint32_t)__popcll(bal_adm);
            if ((blockIdx.x * PER + k) * 256u < n) adm.ballots[(blockIdx.x * PER + k) * 4u + (threadIdx.x >> 6)] = bal_adm;
        }
    }
    __syncthreads();
    if (threadIdx.x < PER) {
        const uint32_t k = threadIdx.x, g = blockIdx.x * PER + k;
        if (g * 256u < n) {  // groups past the end belong to nobody (the buffers are sized by ceil(n / 256))
            block_visible[g] = wave_cnt[k][0] + wave_cnt[k][1] + wave_cnt[k][2] + wave_cnt[k][3];
            adm.block_counts[g] = wave_adm[k][0] + wave_adm[k][1] + wave_adm[k][2] + wave_adm[k][3];
        }
    }
}

// Shading of a lazily projected frame: pairs[0 .. *d_n) are admitted records; each gets its conic / colour records —
// same code, same values as the unlazy projection.  skip (nullable): ballots of the records that are shaded already
// (repair round: what the first round admitted).  One record per lane, gathered: position, covariance, and the SH
// record copy whose cache lines are used whole.
template <int DEG, int SHK, int COVK>
__global__ __launch_bounds__(256) void k_shade(const FrameConsts f, const uint32_t n, const PodPlanes pod, const Records rec,
                                                const uint2* __restrict__ pairs, const uint32_t* __restrict__ d_n,
                                                const unsigned long long* __restrict__ skip, const int write_a) {
    const uint32_t count = *d_n;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < count; j += gridDim.x * 256u) {
        const uint32_t i = pairs[j].y;
        if (skip && ((skip[i >> 6] >> (i & 63u)) & 1ull)) continue;
        // the shade record holds position and covariance too: no other gather (Sh None models run the f32 instantiation without a record)
        const bool full = pod.sh_aos != nullptr && pod.aos_geo != 0u;
        const uint4* __restrict__ geo = pod.sh_aos + (uint64_t)i * pod.aos_stride + pod.aos_geo;
        float4 pc;
        if (full) {
            const uint4 w = geo[0];
            pc = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z), __uint_as_float(w.w));
        } else {
            pc = pod.pc[i];
        }
        ViewClip vc;
        Splat2D sp{};
        if (!pm_view_cull(f, pc.x, pc.y, pc.z, vc)) continue;  // cannot happen: it is visible
        if (full && COVK == GSX_COV3D_SINGLE) {
            const uint4 a = geo[1], b2 = geo[2];
            if (!pm_cov2d_rect(f, vc, __uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                               __uint_as_float(b2.x), __uint_as_float(b2.y), sp))
                continue;
        } else if (full) {
            const uint4 a = geo[1];
            if (!pm_cov2d_rect(f, vc, h_lo(a.x), h_hi(a.x), h_lo(a.y), h_hi(a.y), h_lo(a.z), h_hi(a.z), sp)) continue;
        } else if (!load_cov2d_rect<COVK>(f, pod, i, vc, sp)) {
            continue;
        }
        float r, g, b;
        load_shade<DEG, SHK, true>(f, pod, n, i, pc, r, g, b);
        if (write_a) rec.a[i] = make_float4(sp.mx, sp.my, __uint_as_float(sp.rx), __uint_as_float(sp.ry));
        rec.b[i] = make_float4(sp.con_a, sp.con_b, sp.con_c, (float)(__float_as_uint(pc.w) >> 24) * (1.0f / 255.0f));
        rec.c[i] = make_float4(r, g, b, vc.d);
    }
}

// k_shade for the 256-byte record copy (f32 SH + f32 covariance), four lanes to a record.  One record per lane meant sixteen 16-byte
// loads 256 bytes apart from lane to lane: every load instruction touched 64 different lines, a wave's working set was 16 KB of a
// 32 KB L1 shared by eight waves, and the sectors were fetched again and again (75 MB of records in 41 us: 1.8 TB/s).  Here the
// four lanes of a quad load the record side by side — lane s takes words s, s + 4, s + 8, s + 12: every instruction reads whole
// 64-byte sectors, sixteen sectors a wave — and hand each other their words by quad broadcasts (v_mov_dpp: no LDS, no barrier).
// All four lanes then run the SAME arithmetic on the same sixteen words — the code of k_shade, value for value — and lane 0 stores.
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v, const int s) {
    // quad_perm [s, s, s, s]
    switch (s) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, false);
    }
}
// the SH words of a shade record (plane order, as the pod stores them) -> the stream's floats: what load_shade<.., AOS = true> feeds
template <int DEG, int SHK>
__device__ __forceinline__ void feed_record_words(ShStream<DEG>& st, const uint4* w) {
    constexpr int kFloats = ShNeed<DEG>::floats;
    if (SHK == GSX_SH_SINGLE) {
#pragma unroll
        for (int p = 0; p < ShNeed<DEG>::planes4; ++p) {
            st.feed(4 * p, __uint_as_float(w[p].x)); st.feed(4 * p + 1, __uint_as_float(w[p].y));
            st.feed(4 * p + 2, __uint_as_float(w[p].z)); st.feed(4 * p + 3, __uint_as_float(w[p].w));
        }
        if (DEG == 3) st.feed(44, __uint_as_float(w[11].x));
    } else if (SHK == GSX_SH_HALF) {
        constexpr int kP = (kFloats + 7) / 8;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            st.feed(8 * p, h_lo(w[p].x)); st.feed(8 * p + 1, h_hi(w[p].x)); st.feed(8 * p + 2, h_lo(w[p].y)); st.feed(8 * p + 3, h_hi(w[p].y));
            st.feed(8 * p + 4, h_lo(w[p].z)); st.feed(8 * p + 5, h_hi(w[p].z)); st.feed(8 * p + 6, h_lo(w[p].w)); st.feed(8 * p + 7, h_hi(w[p].w));
        }
    } else {
        constexpr int kP = (kFloats + 15) / 16;
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const uint32_t q[4] = {w[p].x, w[p].y, w[p].z, w[p].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) st.feed(16 * p + 4 * k + bb, dq_snorm8(q[k], bb));
        }
    }
}

// Four lanes to a shade record, every pod kind (round 5; f32 pods since round 4): the quad loads the record side by side — STRIDE / 4
// coalesced 16-byte loads per lane, whole 64-byte sectors — and every lane gets every word by quad broadcasts.
template <int DEG, int SHK, int COVK>
__global__ __launch_bounds__(256) void k_shade_quads(const FrameConsts f, const PodPlanes pod, const Records rec,
                                                      const uint2* __restrict__ pairs, const uint32_t* __restrict__ d_n,
                                                      const unsigned long long* __restrict__ skip, const int write_a) {
    constexpr uint32_t kStride = SHK == GSX_SH_SINGLE ? 16u : (SHK == GSX_SH_HALF ? (COVK == GSX_COV3D_SINGLE ? 12u : 8u) : 8u);
    constexpr uint32_t kGeo = SHK == GSX_SH_SINGLE ? 12u : (SHK == GSX_SH_HALF ? 6u : 3u);
    constexpr int kLoads = (int)(kStride / 4u);
    const uint32_t count = *d_n;
    const uint32_t sub = threadIdx.x & 3u;
    for (uint32_t j = (blockIdx.x * 256u + threadIdx.x) >> 2; j < count; j += gridDim.x * 64u) {  // (the quad's four lanes share j)
        const uint32_t i = pairs[j].y;
        if (skip && ((skip[i >> 6] >> (i & 63u)) & 1ull)) continue;
        uint4 mine[kLoads];
#pragma unroll
        for (int k = 0; k < kLoads; ++k) mine[k] = pod.sh_aos[(uint64_t)i * kStride + sub + 4u * (uint32_t)k];
        uint4 w[kStride];   // the record's words, in every lane of the quad
#pragma unroll
        for (int k = 0; k < kLoads; ++k)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                w[sl + 4 * k] = make_uint4(quad_bcast(mine[k].x, sl), quad_bcast(mine[k].y, sl), quad_bcast(mine[k].z, sl), quad_bcast(mine[k].w, sl));
        const float4 pc = make_float4(__uint_as_float(w[kGeo].x), __uint_as_float(w[kGeo].y), __uint_as_float(w[kGeo].z), __uint_as_float(w[kGeo].w));
        ViewClip vc;
        Splat2D sp{};
        if (!pm_view_cull(f, pc.x, pc.y, pc.z, vc)) continue;  // cannot happen: it is visible
        if (COVK == GSX_COV3D_SINGLE) {
            if (!pm_cov2d_rect(f, vc, __uint_as_float(w[kGeo + 1].x), __uint_as_float(w[kGeo + 1].y), __uint_as_float(w[kGeo + 1].z), __uint_as_float(w[kGeo + 1].w),
                               __uint_as_float(w[kGeo + 2].x), __uint_as_float(w[kGeo + 2].y), sp))
                continue;
        } else {
            const uint4 a = w[kGeo + 1];
            if (!pm_cov2d_rect(f, vc, h_lo(a.x), h_hi(a.x), h_lo(a.y), h_hi(a.y), h_lo(a.z), h_hi(a.z), sp)) continue;
        }
        ShStream<DEG> st;   // (load_shade<DEG, SHK, true>, fed from registers)
        st.begin(f, pc.x, pc.y, pc.z, __float_as_uint(pc.w));
        feed_record_words<DEG, SHK>(st, w);
        float r, g, b;
        st.finish(r, g, b);
        if (sub == 0u) {
            if (write_a) rec.a[i] = make_float4(sp.mx, sp.my, __uint_as_float(sp.rx), __uint_as_float(sp.ry));
            rec.b[i] = make_float4(sp.con_a, sp.con_b, sp.con_c, (float)(__float_as_uint(pc.w) >> 24) * (1.0f / 255.0f));
            rec.c[i] = make_float4(r, g, b, vc.d);
        }
    }
}

// N_vis = sum of the per-workgroup counts (single workgroup; <= 40 K entries at 10 M Gaussians)
__global__ __launch_bounds__(1024) void k_sum_counts(const uint32_t* __restrict__ block_visible, uint32_t nblocks,
                                                      uint32_t* __restrict__ n_visible) {
    __shared__ uint32_t red[16];
    uint32_t s = 0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += 1024) s += block_visible[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        *n_visible = t;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

hipError_t launch_convert(hipStream_t s, const gsx_gaussian* d_src, uint64_t n, uint64_t start, uint64_t model_n,
                          const PodPlanes& pod) {
    if (n == 0) return hipSuccess;
    GSX_LAUNCH(k_convert, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_src, n, start, model_n, pod);
    return hipGetLastError();
}

hipError_t launch_pack_pod(hipStream_t s, const float* d_pos, const uint32_t* d_color, const float* d_sh,
                           const float* d_cov, uint64_t n, uint64_t start, uint64_t model_n, const PodPlanes& pod) {
    if (n == 0) return hipSuccess;
    GSX_LAUNCH(k_pack_pod, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_pos, d_color, d_sh, d_cov, n, start,
                       model_n, pod);
    return hipGetLastError();
}

hipError_t launch_unpack_pod(hipStream_t s, const PodPlanes& pod, uint64_t model_n, float* d_pos, uint32_t* d_color,
                             float* d_sh, float* d_cov) {
    if (model_n == 0) return hipSuccess;
    GSX_LAUNCH(k_unpack_pod, dim3(blocks_for(model_n, 256)), dim3(256), 0, s, pod, model_n, d_pos, d_color, d_sh,
                       d_cov);
    return hipGetLastError();
}

size_t project_blocks(uint64_t n) { return (size_t)((n + 255) / 256); }

hipError_t launch_sum_counts(hipStream_t s, const uint32_t* d_block_visible, uint32_t n, uint32_t* d_n_visible) {
    GSX_LAUNCH(k_sum_counts, dim3(1), dim3(1024), 0, s, d_block_visible, (uint32_t)project_blocks(n), d_n_visible);
    return hipGetLastError();
}

template <int SHK, int COVK>
static void launch_project_deg(hipStream_t s, dim3 grid, int deg, const FrameConsts& f, uint32_t n, const PodPlanes& pod,
                               const Records& rec, uint32_t* bv, const ProjectAdmission& adm, const LateProjection* late) {
    dim3 block(256);
#define GSX_PROJECT(D)                                                                                                       \
    if (late)                                                                                                                \
        GSX_LAUNCH((k_shade<D, SHK, COVK>), grid, block, 0, s, f, n, pod, rec, late->pairs, late->d_n, late->shaded, late->write_a ? 1 : 0); \
    else if (adm.lazy && adm.query.flags)                                                                                    \
        GSX_LAUNCH((k_project_geom<COVK, kProjGeomPer, true>), dim3((grid.x + kProjGeomPer - 1) / kProjGeomPer),     \
                           block, 0, s, f, n, pod, rec, bv, adm);                                                            \
    else if (adm.lazy)                                                                                                       \
        GSX_LAUNCH((k_project_geom<COVK, kProjGeomPer, false>), dim3((grid.x + kProjGeomPer - 1) / kProjGeomPer),    \
                           block, 0, s, f, n, pod, rec, bv, adm);                                                            \
    else                                                                                                                     \
        GSX_LAUNCH((k_project<D, SHK, COVK>), grid, block, 0, s, f, n, pod, rec, bv, adm)
    switch (deg) {
        case 0: GSX_PROJECT(0); break;
        case 1: GSX_PROJECT(1); break;
        case 2: GSX_PROJECT(2); break;
        default: GSX_PROJECT(3); break;
    }
#undef GSX_PROJECT
}

static hipError_t dispatch_project(hipStream_t s, dim3 grid, const FrameConsts& f, uint32_t n, const PodPlanes& pod, const Records& rec,
                                   uint32_t* d_block_visible, const ProjectAdmission& adm, const LateProjection* late) {
    const int deg = pod.sh_kind == GSX_SH_NONE ? 0 : (int)f.sh_deg;
    const bool ch = pod.cov_kind == GSX_COV3D_HALF;
    switch (pod.sh_kind) {
        case GSX_SH_HALF:
            if (ch) launch_project_deg<GSX_SH_HALF, GSX_COV3D_HALF>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            else launch_project_deg<GSX_SH_HALF, GSX_COV3D_SINGLE>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            break;
        case GSX_SH_NORM8:
            if (ch) launch_project_deg<GSX_SH_NORM8, GSX_COV3D_HALF>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            else launch_project_deg<GSX_SH_NORM8, GSX_COV3D_SINGLE>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            break;
        default:  // Single, or None (deg 0 touches no SH plane)
            if (ch) launch_project_deg<GSX_SH_SINGLE, GSX_COV3D_HALF>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            else launch_project_deg<GSX_SH_SINGLE, GSX_COV3D_SINGLE>(s, grid, deg, f, n, pod, rec, d_block_visible, adm, late);
            break;
    }
    return hipGetLastError();
}

hipError_t launch_project(hipStream_t s, const FrameConsts& f, uint32_t n, const PodPlanes& pod, const Records& rec,
                          uint32_t* d_block_visible, const ProjectAdmission& adm) {
    if (n == 0) return hipSuccess;
    return dispatch_project(s, dim3(blocks_for(n, 256)), f, n, pod, rec, d_block_visible, adm, nullptr);
}

hipError_t launch_shade(hipStream_t s, const FrameConsts& f, uint32_t n, const PodPlanes& pod, const Records& rec,
                        const LateProjection& late) {
    if (n == 0) return hipSuccess;
    // (four lanes to a record: + 3 % on the frame against one lane per record, same values — rounds 4 and 5, profiles/r05_*; the A/B switch is gone)
    if (pod.sh_kind != GSX_SH_NONE && pod.sh_aos != nullptr && pod.aos_geo != 0u) {
        const dim3 grid(8192), block(256);   // 0.5 M quads stride over the admitted records
#define GSX_QUADS(SHK, COVK)                                                                                                                      \
    switch ((int)f.sh_deg) {                                                                                                                      \
        case 0: GSX_LAUNCH((k_shade_quads<0, SHK, COVK>), grid, block, 0, s, f, pod, rec, late.pairs, late.d_n, late.shaded, late.write_a ? 1 : 0); break; \
        case 1: GSX_LAUNCH((k_shade_quads<1, SHK, COVK>), grid, block, 0, s, f, pod, rec, late.pairs, late.d_n, late.shaded, late.write_a ? 1 : 0); break; \
        case 2: GSX_LAUNCH((k_shade_quads<2, SHK, COVK>), grid, block, 0, s, f, pod, rec, late.pairs, late.d_n, late.shaded, late.write_a ? 1 : 0); break; \
        default: GSX_LAUNCH((k_shade_quads<3, SHK, COVK>), grid, block, 0, s, f, pod, rec, late.pairs, late.d_n, late.shaded, late.write_a ? 1 : 0); break; \
    }
        const bool ch = pod.cov_kind == GSX_COV3D_HALF;
        if (pod.sh_kind == GSX_SH_SINGLE) {
            if (ch) { GSX_QUADS(GSX_SH_SINGLE, GSX_COV3D_HALF) } else { GSX_QUADS(GSX_SH_SINGLE, GSX_COV3D_SINGLE) }
        } else if (pod.sh_kind == GSX_SH_HALF) {
            if (ch) { GSX_QUADS(GSX_SH_HALF, GSX_COV3D_HALF) } else { GSX_QUADS(GSX_SH_HALF, GSX_COV3D_SINGLE) }
        } else {
            if (ch) { GSX_QUADS(GSX_SH_NORM8, GSX_COV3D_HALF) } else { GSX_QUADS(GSX_SH_NORM8, GSX_COV3D_SINGLE) }
        }
#undef GSX_QUADS
        return hipGetLastError();
    }
    return dispatch_project(s, dim3(4096), f, n, pod, rec, nullptr, ProjectAdmission{}, &late);  // 1 M lanes stride over the admitted records
}

}  // namespace gsx
