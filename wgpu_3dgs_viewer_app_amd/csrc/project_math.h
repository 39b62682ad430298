// project_math.h — per-Gaussian arithmetic of the projection pass (spec/RENDER_SPEC.md §4), shared by
// the product kernel (kernels_project.hip) and the layout micro-benchmark (tools/bench_project.hip).
// Operation order is part of the spec: oracle/gsx_oracle.c:project_one performs the same float32
// operations in the same order (no contraction), which makes cull set, depth key and tile rectangle
// bit-exact between the two.
#pragma once
#include "gsx_internal.h"

namespace gsx {

__device__ inline float pm_dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}
__device__ inline float pm_clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

struct ViewClip {
    float pv0, pv1, pv2;  // view-space position
    float xc, yc, wc;     // clip x, y, w
    float d;              // view depth along -Z
};

// spec §4.1-4.2: view / clip transform and frustum cull.  Returns visibility.
__device__ inline bool pm_view_cull(const FrameConsts& f, float x, float y, float z, ViewClip& o) {
    o.pv0 = pm_dot3(f.T[0], f.T[1], f.T[2], x, y, z) + f.vt[0];
    o.pv1 = pm_dot3(f.T[3], f.T[4], f.T[5], x, y, z) + f.vt[1];
    o.pv2 = pm_dot3(f.T[6], f.T[7], f.T[8], x, y, z) + f.vt[2];
    o.xc = ((f.P[0] * o.pv0 + f.P[4] * o.pv1) + f.P[8] * o.pv2) + f.P[12];
    o.yc = ((f.P[1] * o.pv0 + f.P[5] * o.pv1) + f.P[9] * o.pv2) + f.P[13];
    float zc = ((f.P[2] * o.pv0 + f.P[6] * o.pv1) + f.P[10] * o.pv2) + f.P[14];
    o.wc = ((f.P[3] * o.pv0 + f.P[7] * o.pv1) + f.P[11] * o.pv2) + f.P[15];
    float lim = f.cull_margin * o.wc;
    o.d = -o.pv2;
    return (o.wc > 0.0f) && (o.xc >= -lim && o.xc <= lim && o.yc >= -lim && o.yc <= lim && zc >= 0.0f && zc <= o.wc) &&
           (o.d > 0.0f);
}

struct Splat2D {
    float mx, my;               // pixel-space mean
    float con_a, con_b, con_c;  // conic (inverse 2D covariance)
    uint32_t rx, ry;            // tile rect: x0 | x1<<16, y0 | y1<<16 (max exclusive)
};

// spec §4.3-4.6: EWA covariance projection, low-pass, conic, screen position, tile rectangle.
// cov = (xx, xy, xz, yy, yz, zz).  Returns visibility.
__device__ inline bool pm_cov2d_rect(const FrameConsts& f, const ViewClip& v, float sxx, float sxy, float sxz,
                                     float syy, float syz, float szz, Splat2D& o) {
    float inv_d = 1.0f / v.d;
    float tx = pm_clamp(v.pv0 * inv_d, -f.limx, f.limx);
    float ty = pm_clamp(v.pv1 * inv_d, -f.limy, f.limy);
    float j00 = f.fx * inv_d, j02 = (f.fx * tx) * inv_d;
    float j11 = -(f.fy * inv_d), j12 = -((f.fy * ty) * inv_d);
    float a00 = j00 * f.T[0] + j02 * f.T[6], a01 = j00 * f.T[1] + j02 * f.T[7], a02 = j00 * f.T[2] + j02 * f.T[8];
    float a10 = j11 * f.T[3] + j12 * f.T[6], a11 = j11 * f.T[4] + j12 * f.T[7], a12 = j11 * f.T[5] + j12 * f.T[8];
    float v00 = pm_dot3(sxx, sxy, sxz, a00, a01, a02);
    float v01 = pm_dot3(sxy, syy, syz, a00, a01, a02);
    float v02 = pm_dot3(sxz, syz, szz, a00, a01, a02);
    float v10 = pm_dot3(sxx, sxy, sxz, a10, a11, a12);
    float v11 = pm_dot3(sxy, syy, syz, a10, a11, a12);
    float v12 = pm_dot3(sxz, syz, szz, a10, a11, a12);
    float ca = pm_dot3(a00, a01, a02, v00, v01, v02);
    float cb = pm_dot3(a10, a11, a12, v00, v01, v02);
    float cc = pm_dot3(a10, a11, a12, v10, v11, v12);
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
    bool vis = det > 0.0f;
    float inv_det = 1.0f / det;
    float inv_w = 1.0f / v.wc;
    float ndcx = v.xc * inv_w, ndcy = v.yc * inv_w;
    o.mx = (ndcx * 0.5f + 0.5f) * f.width;
    o.my = (0.5f - ndcy * 0.5f) * f.height;
    float ex = f.k * sqrtf(ca), ey = f.k * sqrtf(cc);
    float x0f = ceilf((o.mx - ex) - 0.5f), x1f = floorf((o.mx + ex) - 0.5f);
    float y0f = ceilf((o.my - ey) - 0.5f), y1f = floorf((o.my + ey) - 0.5f);
    x0f = fmaxf(x0f, 0.0f);
    y0f = fmaxf(y0f, 0.0f);
    x1f = fminf(x1f, f.width - 1.0f);
    y1f = fminf(y1f, f.height - 1.0f);
    vis = vis && (x0f <= x1f && y0f <= y1f);
    o.rx = 0;
    o.ry = 0;
    if (vis) {
        uint32_t x0 = (uint32_t)(int)x0f, x1 = (uint32_t)(int)x1f, y0 = (uint32_t)(int)y0f, y1 = (uint32_t)(int)y1f;
        o.rx = (x0 / kTile) | (((x1 / kTile) + 1u) << 16);
        o.ry = (y0 / kTile) | (((y1 / kTile) + 1u) << 16);
    }
    o.con_a = cc * inv_det;
    o.con_b = -(cb * inv_det);
    o.con_c = ca * inv_det;
    return vis;
}

// spec §4.7: colour = DC (UNORM8) + SH degrees 1..DEG evaluated along the model-space view direction.
// s[] holds SH floats, index 3*coeff + channel.
template <int DEG>
__device__ inline void pm_color(const FrameConsts& f, float x_, float y_, float z_, uint32_t color, const float* s,
                                float& r, float& g, float& b) {
    constexpr float C1 = 0.4886025119029199f;
    constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                    C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                    C3_6 = -0.5900435899266435f;
    r = g = b = 0.0f;
    if (!f.no_sh0) {
        r = (float)(color & 255u) * (1.0f / 255.0f);
        g = (float)((color >> 8) & 255u) * (1.0f / 255.0f);
        b = (float)((color >> 16) & 255u) * (1.0f / 255.0f);
    }
    if (DEG > 0) {
        float dx = f.s_m[0] * x_ - f.cam_m[0];
        float dy = f.s_m[1] * y_ - f.cam_m[1];
        float dz = f.s_m[2] * z_ - f.cam_m[2];
        float len = sqrtf((dx * dx + dy * dy) + dz * dz);
        float il = 1.0f / len;
        float x = dx * il, y = dy * il, z = dz * il;
        float acc[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = -C1 * y * s[0 + ch] + C1 * z * s[3 + ch] - C1 * x * s[6 + ch];
            if (DEG > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v += C2_0 * xy * s[9 + ch] + C2_1 * yz * s[12 + ch] + C2_2 * (2.0f * zz - xx - yy) * s[15 + ch] +
                     C2_3 * xz * s[18 + ch] + C2_4 * (xx - yy) * s[21 + ch];
                if (DEG > 2) {
                    v += C3_0 * y * (3.0f * xx - yy) * s[24 + ch] + C3_1 * xy * z * s[27 + ch] +
                         C3_2 * y * (4.0f * zz - xx - yy) * s[30 + ch] +
                         C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[33 + ch] +
                         C3_4 * x * (4.0f * zz - xx - yy) * s[36 + ch] + C3_5 * z * (xx - yy) * s[39 + ch] +
                         C3_6 * x * (xx - 3.0f * yy) * s[42 + ch];
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

// The same colour, evaluated as the SH floats ARRIVE (one float4 plane at a time) instead of from an array of 45: the 15 basis
// factors come first (they need only the position), every loaded float is consumed by one multiply-add, and nothing but the
// loads still in flight occupies registers (pm_color keeps all 45 floats live: 84 VGPRs, 5 waves per SIMD).  Bit-identical to
// pm_color by construction: per channel the spec's expression is three left-to-right chains — degree 1, then "v += (degree-2
// terms)", then "v += (degree-3 terms)" — and each term is ((constant * polynomial) * s), so a chain per degree fed in
// coefficient order, joined at the end, performs the same float32 operations in the same order.
template <int DEG> struct ShStream {
    float basis[15];
    float g1[3], g2[3], g3[3];
    float dc[3];

    __device__ inline void begin(const FrameConsts& f, float x_, float y_, float z_, uint32_t color) {
        constexpr float C1 = 0.4886025119029199f;
        constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                        C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
        constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                        C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                        C3_6 = -0.5900435899266435f;
        dc[0] = dc[1] = dc[2] = 0.0f;
        if (!f.no_sh0) {
            dc[0] = (float)(color & 255u) * (1.0f / 255.0f);
            dc[1] = (float)((color >> 8) & 255u) * (1.0f / 255.0f);
            dc[2] = (float)((color >> 16) & 255u) * (1.0f / 255.0f);
        }
        if (DEG > 0) {
            float dx = f.s_m[0] * x_ - f.cam_m[0];
            float dy = f.s_m[1] * y_ - f.cam_m[1];
            float dz = f.s_m[2] * z_ - f.cam_m[2];
            float len = sqrtf((dx * dx + dy * dy) + dz * dz);
            float il = 1.0f / len;
            float x = dx * il, y = dy * il, z = dz * il;
            basis[0] = -C1 * y;
            basis[1] = C1 * z;
            basis[2] = -(C1 * x);  // "- C1 * x * s": a - b*s == a + (-b)*s exactly
            if (DEG > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                basis[3] = C2_0 * xy;
                basis[4] = C2_1 * yz;
                basis[5] = C2_2 * (2.0f * zz - xx - yy);
                basis[6] = C2_3 * xz;
                basis[7] = C2_4 * (xx - yy);
                if (DEG > 2) {
                    basis[8] = C3_0 * y * (3.0f * xx - yy);
                    basis[9] = C3_1 * xy * z;
                    basis[10] = C3_2 * y * (4.0f * zz - xx - yy);
                    basis[11] = C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                    basis[12] = C3_4 * x * (4.0f * zz - xx - yy);
                    basis[13] = C3_5 * z * (xx - yy);
                    basis[14] = C3_6 * x * (xx - 3.0f * yy);
                }
            }
        }
    }
    // SH float F (= 3 * coefficient + channel; a compile-time constant once the caller's loops are unrolled) has arrived
    __device__ __forceinline__ void feed(const int F, float s) {
        const int c = F / 3, ch = F % 3;
        if (c >= (DEG == 1 ? 3 : (DEG == 2 ? 8 : 15))) return;
        const float t = basis[c] * s;
        if (c == 0) g1[ch] = t;
        else if (c < 3) g1[ch] = g1[ch] + t;
        else if (c == 3) g2[ch] = t;
        else if (c < 8) g2[ch] = g2[ch] + t;
        else if (c == 8) g3[ch] = t;
        else g3[ch] = g3[ch] + t;
    }
    __device__ inline void finish(float& r, float& g, float& b) const {
        float o[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = dc[ch];
            if (DEG > 0) {
                float a = g1[ch];
                if (DEG > 1) a += g2[ch];
                if (DEG > 2) a += g3[ch];
                v += a;
            }
            o[ch] = fmaxf(v, 0.0f);
        }
        r = o[0]; g = o[1]; b = o[2];
    }
};

// number of SH floats / float4 planes needed for a degree
template <int DEG> struct ShNeed {
    static constexpr int floats = DEG == 0 ? 0 : (DEG == 1 ? 9 : (DEG == 2 ? 24 : 45));
    static constexpr int planes4 = floats == 45 ? 11 : (floats + 3) / 4;
};

}  // namespace gsx
