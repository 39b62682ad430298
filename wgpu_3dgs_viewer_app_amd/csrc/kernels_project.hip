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
#include "gsx_internal.h"

namespace gsx {

// ------------------------------------------------------------------------------------------------
// host: frame constants (spec §3)
// ------------------------------------------------------------------------------------------------
static void quat_rows(const float q[4], float r[9]) {
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

__device__ inline void store_sh_planes(const PodPlanes& pod, uint64_t model_n, uint64_t i, const float* s45) {
#pragma unroll
    for (int p = 0; p < kShPlanes4; ++p)
        pod.sh4[(uint64_t)p * model_n + i] = make_float4(s45[4 * p], s45[4 * p + 1], s45[4 * p + 2], s45[4 * p + 3]);
    pod.sh1[i] = s45[44];
}

__global__ __launch_bounds__(256) void k_convert(const gsx_gaussian* __restrict__ src, uint64_t n, uint64_t start,
                                                  uint64_t model_n, PodPlanes pod, int has_sh) {
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
    pod.cov_a[i] = make_float4(c0, c1, c2, c3);
    pod.cov_b[i] = make_float2(c4, c5);
    if (has_sh) store_sh_planes(pod, model_n, i, &g.sh[0][0]);
}

// pod-ready planes (pos 3n, color n, sh 45n, cov 6n) -> resident float4 planes, and back (parity readback).
__global__ __launch_bounds__(256) void k_pack_pod(const float* __restrict__ pos, const uint32_t* __restrict__ color,
                                                   const float* __restrict__ sh, const float* __restrict__ cov,
                                                   uint64_t n, uint64_t start, uint64_t model_n, PodPlanes pod) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t i = start + t;
    pod.pc[i] = make_float4(pos[3 * t], pos[3 * t + 1], pos[3 * t + 2], __uint_as_float(color[t]));
    pod.cov_a[i] = make_float4(cov[6 * t], cov[6 * t + 1], cov[6 * t + 2], cov[6 * t + 3]);
    pod.cov_b[i] = make_float2(cov[6 * t + 4], cov[6 * t + 5]);
    if (sh) store_sh_planes(pod, model_n, i, sh + 45 * t);
}

__global__ __launch_bounds__(256) void k_unpack_pod(PodPlanes pod, uint64_t model_n, float* __restrict__ pos,
                                                     uint32_t* __restrict__ color, float* __restrict__ sh,
                                                     float* __restrict__ cov, int has_sh) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= model_n) return;
    float4 pc = pod.pc[i];
    float4 ca = pod.cov_a[i];
    float2 cb = pod.cov_b[i];
    pos[3 * i] = pc.x; pos[3 * i + 1] = pc.y; pos[3 * i + 2] = pc.z;
    color[i] = __float_as_uint(pc.w);
    cov[6 * i] = ca.x; cov[6 * i + 1] = ca.y; cov[6 * i + 2] = ca.z; cov[6 * i + 3] = ca.w;
    cov[6 * i + 4] = cb.x; cov[6 * i + 5] = cb.y;
    if (sh) {
        for (int p = 0; p < kShPlanes4; ++p) {
            float4 v = has_sh ? pod.sh4[(uint64_t)p * model_n + i] : make_float4(0, 0, 0, 0);
            sh[45 * i + 4 * p] = v.x; sh[45 * i + 4 * p + 1] = v.y; sh[45 * i + 4 * p + 2] = v.z; sh[45 * i + 4 * p + 3] = v.w;
        }
        sh[45 * i + 44] = has_sh ? pod.sh1[i] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// projection pass.  One Gaussian per lane, 256 lanes per workgroup; N/256 workgroups (>> 256 CUs).
// DEG = SH degree evaluated (0 = DC only, no SH planes touched).
// ------------------------------------------------------------------------------------------------
__device__ __constant__ const float kShC1 = 0.4886025119029199f;
__device__ __constant__ const float kShC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                               -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kShC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                               -0.5900435899266435f};

__device__ inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

template <int DEG>
__global__ __launch_bounds__(256) void k_project(const FrameConsts f, const uint32_t n, const PodPlanes pod,
                                                  const Records rec, uint32_t* __restrict__ n_visible) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool vis = i < n;
    float4 pc = make_float4(0, 0, 0, 0);
    if (vis) pc = pod.pc[i];
    if (vis && pod.mask) vis = (pod.mask[i >> 5] >> (i & 31)) & 1u;
    const uint32_t color = __float_as_uint(pc.w);

    // view / clip space and the frustum cull (spec §4.1-4.2)
    float pv0 = ddot3(f.T[0], f.T[1], f.T[2], pc.x, pc.y, pc.z) + f.vt[0];
    float pv1 = ddot3(f.T[3], f.T[4], f.T[5], pc.x, pc.y, pc.z) + f.vt[1];
    float pv2 = ddot3(f.T[6], f.T[7], f.T[8], pc.x, pc.y, pc.z) + f.vt[2];
    float xc = ((f.P[0] * pv0 + f.P[4] * pv1) + f.P[8] * pv2) + f.P[12];
    float yc = ((f.P[1] * pv0 + f.P[5] * pv1) + f.P[9] * pv2) + f.P[13];
    float zc = ((f.P[2] * pv0 + f.P[6] * pv1) + f.P[10] * pv2) + f.P[14];
    float wc = ((f.P[3] * pv0 + f.P[7] * pv1) + f.P[11] * pv2) + f.P[15];
    float lim = f.cull_margin * wc;
    float d = -pv2;
    vis = vis && (wc > 0.0f) && (xc >= -lim && xc <= lim && yc >= -lim && yc <= lim && zc >= 0.0f && zc <= wc) &&
          (d > 0.0f);

    float mx = 0, my = 0, con_a = 0, con_b = 0, con_c = 0;
    uint32_t rx = 0, ry = 0;
    if (vis) {
        // the covariance planes are only fetched for Gaussians that survive the frustum test
        float4 cva = pod.cov_a[i];
        float2 cvb = pod.cov_b[i];
        float inv_d = 1.0f / d;
        float tx = clampf(pv0 * inv_d, -f.limx, f.limx);
        float ty = clampf(pv1 * inv_d, -f.limy, f.limy);
        float j00 = f.fx * inv_d, j02 = (f.fx * tx) * inv_d;
        float j11 = -(f.fy * inv_d), j12 = -((f.fy * ty) * inv_d);
        float a00 = j00 * f.T[0] + j02 * f.T[6], a01 = j00 * f.T[1] + j02 * f.T[7], a02 = j00 * f.T[2] + j02 * f.T[8];
        float a10 = j11 * f.T[3] + j12 * f.T[6], a11 = j11 * f.T[4] + j12 * f.T[7], a12 = j11 * f.T[5] + j12 * f.T[8];
        // Sigma rows: (xx xy xz) (xy yy yz) (xz yz zz)
        float v00 = ddot3(cva.x, cva.y, cva.z, a00, a01, a02);
        float v01 = ddot3(cva.y, cva.w, cvb.x, a00, a01, a02);
        float v02 = ddot3(cva.z, cvb.x, cvb.y, a00, a01, a02);
        float v10 = ddot3(cva.x, cva.y, cva.z, a10, a11, a12);
        float v11 = ddot3(cva.y, cva.w, cvb.x, a10, a11, a12);
        float v12 = ddot3(cva.z, cvb.x, cvb.y, a10, a11, a12);
        float ca = ddot3(a00, a01, a02, v00, v01, v02);
        float cb = ddot3(a10, a11, a12, v00, v01, v02);
        float cc = ddot3(a10, a11, a12, v10, v11, v12);
        if (f.display_mode == GSX_DISPLAY_POINT) {
            float rp = f.point_radius / f.k;
            ca = rp * rp - f.low_pass;
            cb = 0.0f;
            cc = rp * rp - f.low_pass;
        }
        ca = (ca + f.low_pass) * f.size2;
        cb = cb * f.size2;
        cc = (cc + f.low_pass) * f.size2;
        float det = ca * cc - cb * cb;
        vis = det > 0.0f;
        float inv_det = 1.0f / det;
        float inv_w = 1.0f / wc;
        float ndcx = xc * inv_w, ndcy = yc * inv_w;
        mx = (ndcx * 0.5f + 0.5f) * f.width;
        my = (0.5f - ndcy * 0.5f) * f.height;
        float ex = f.k * sqrtf(ca), ey = f.k * sqrtf(cc);
        float x0f = ceilf((mx - ex) - 0.5f), x1f = floorf((mx + ex) - 0.5f);
        float y0f = ceilf((my - ey) - 0.5f), y1f = floorf((my + ey) - 0.5f);
        x0f = fmaxf(x0f, 0.0f);
        y0f = fmaxf(y0f, 0.0f);
        x1f = fminf(x1f, f.width - 1.0f);
        y1f = fminf(y1f, f.height - 1.0f);
        vis = vis && (x0f <= x1f && y0f <= y1f);
        if (vis) {
            uint32_t x0 = (uint32_t)(int)x0f, x1 = (uint32_t)(int)x1f, y0 = (uint32_t)(int)y0f, y1 = (uint32_t)(int)y1f;
            rx = (x0 / kTile) | (((x1 / kTile) + 1u) << 16);
            ry = (y0 / kTile) | (((y1 / kTile) + 1u) << 16);
        }
        con_a = cc * inv_det;
        con_b = -(cb * inv_det);
        con_c = ca * inv_det;
    }

    float r = 0, g = 0, b = 0;
    if (vis) {
        if (!f.no_sh0) {
            r = (float)(color & 255u) * (1.0f / 255.0f);
            g = (float)((color >> 8) & 255u) * (1.0f / 255.0f);
            b = (float)((color >> 16) & 255u) * (1.0f / 255.0f);
        }
        if (DEG > 0) {
            // SH planes: float index 3*coeff + channel; plane p = floats 4p..4p+3.  Loaded only for survivors.
            constexpr int kFloats = DEG == 1 ? 9 : (DEG == 2 ? 24 : 45);
            constexpr int kPlanes = (kFloats + 3) / 4 > kShPlanes4 ? kShPlanes4 : (kFloats + 3) / 4;
            float s[48];
#pragma unroll
            for (int p = 0; p < kPlanes; ++p) {
                float4 v = pod.sh4[(uint64_t)p * n + i];
                s[4 * p] = v.x; s[4 * p + 1] = v.y; s[4 * p + 2] = v.z; s[4 * p + 3] = v.w;
            }
            if (DEG == 3) s[44] = pod.sh1[i];
            float dx = f.s_m[0] * pc.x - f.cam_m[0];
            float dy = f.s_m[1] * pc.y - f.cam_m[1];
            float dz = f.s_m[2] * pc.z - f.cam_m[2];
            float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            float il = 1.0f / len;
            float x = dx * il, y = dy * il, z = dz * il;
            float acc[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float v = -kShC1 * y * s[0 + ch] + kShC1 * z * s[3 + ch] - kShC1 * x * s[6 + ch];
                if (DEG > 1) {
                    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    v += kShC2[0] * xy * s[9 + ch] + kShC2[1] * yz * s[12 + ch] +
                         kShC2[2] * (2.0f * zz - xx - yy) * s[15 + ch] + kShC2[3] * xz * s[18 + ch] +
                         kShC2[4] * (xx - yy) * s[21 + ch];
                    if (DEG > 2) {
                        v += kShC3[0] * y * (3.0f * xx - yy) * s[24 + ch] + kShC3[1] * xy * z * s[27 + ch] +
                             kShC3[2] * y * (4.0f * zz - xx - yy) * s[30 + ch] +
                             kShC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[33 + ch] +
                             kShC3[4] * x * (4.0f * zz - xx - yy) * s[36 + ch] + kShC3[5] * z * (xx - yy) * s[39 + ch] +
                             kShC3[6] * x * (xx - 3.0f * yy) * s[42 + ch];
                    }
                }
                acc[ch] = v;
            }
            r += acc[0];
            g += acc[1];
            b += acc[2];
        }
        r = fmaxf(r, 0.0f);
        g = fmaxf(g, 0.0f);
        b = fmaxf(b, 0.0f);
    }

    if (i < n) {
        rec.key[i] = vis ? __float_as_uint(d) : kCulledKey;
        if (vis) {
            rec.a[i] = make_float4(mx, my, __uint_as_float(rx), __uint_as_float(ry));
            rec.b[i] = make_float4(con_a, con_b, con_c, (float)(color >> 24) * (1.0f / 255.0f));
            rec.c[i] = make_float4(r, g, b, d);
        }
    }
    // one atomic per wave: N_vis is order-independent, so this stays deterministic
    unsigned long long bal = __ballot(vis);
    if ((threadIdx.x & 63u) == 0 && bal) atomicAdd(n_visible, (uint32_t)__popcll(bal));
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

hipError_t launch_convert(hipStream_t s, const gsx_gaussian* d_src, uint64_t n, uint64_t start, uint64_t model_n,
                          const PodPlanes& pod, bool has_sh) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_convert, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_src, n, start, model_n, pod, has_sh ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_pack_pod(hipStream_t s, const float* d_pos, const uint32_t* d_color, const float* d_sh,
                           const float* d_cov, uint64_t n, uint64_t start, uint64_t model_n, const PodPlanes& pod) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_pod, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_pos, d_color, d_sh, d_cov, n, start,
                       model_n, pod);
    return hipGetLastError();
}

hipError_t launch_unpack_pod(hipStream_t s, const PodPlanes& pod, uint64_t model_n, float* d_pos, uint32_t* d_color,
                             float* d_sh, float* d_cov, bool has_sh) {
    if (model_n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack_pod, dim3(blocks_for(model_n, 256)), dim3(256), 0, s, pod, model_n, d_pos, d_color, d_sh,
                       d_cov, has_sh ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_project(hipStream_t s, const FrameConsts& f, uint32_t n, const PodPlanes& pod, bool has_sh,
                          const Records& rec, uint32_t* d_n_visible) {
    if (n == 0) return hipSuccess;
    dim3 grid(blocks_for(n, 256)), block(256);
    int deg = has_sh ? (int)f.sh_deg : 0;
    switch (deg) {
        case 0: hipLaunchKernelGGL(k_project<0>, grid, block, 0, s, f, n, pod, rec, d_n_visible); break;
        case 1: hipLaunchKernelGGL(k_project<1>, grid, block, 0, s, f, n, pod, rec, d_n_visible); break;
        case 2: hipLaunchKernelGGL(k_project<2>, grid, block, 0, s, f, n, pod, rec, d_n_visible); break;
        default: hipLaunchKernelGGL(k_project<3>, grid, block, 0, s, f, n, pod, rec, d_n_visible); break;
    }
    return hipGetLastError();
}

}  // namespace gsx
