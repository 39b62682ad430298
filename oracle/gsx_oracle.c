/*
 * gsx_oracle.c — CPU restatement (float32, plain C) of the 3DGS render path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the un-vendored crate
 * `wgpu-3dgs-viewer 0.2.0` (/root/reference/Cargo.lock:3731-3746, checksum 65b8834c…aadd5), which is
 * absent from /root/reference and cannot be built here (no cargo/rustc, no network).  The reference
 * tree holds no tests, fixtures or golden vectors.  This oracle therefore restates the
 * written spec in spec/RENDER_SPEC.md; what IS anchored on the reference is the call protocol and the
 * conventions it proves:
 *   - camera uniform {view, proj, size}, clip -> screen convention    src/shader/measurement.wgsl:14-18, 33-62
 *   - view = look_at_rh(pos,target,Y), proj = perspective_rh(fovy,aspect,near,far) (glam 0.29.2, z in [0,1])
 *                                                                       src/app.rs:1236-1244
 *   - model TRS, quaternion from Euler ZYX degrees                      src/app.rs:1123-1130, src/tab/scene.rs:796-802
 *   - gaussian transform {size, display_mode, sh_deg, no_sh0}           src/app.rs:1141-1165, src/tab/scene.rs:803-809
 *   - per model: preprocess (cull + depth key) -> radix sort -> draw in sorted order with "over" blending,
 *     models painted far -> near by centre distance                    src/tab/scene.rs:856-869, 533-558, 2302-2314
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 * It follows the REFERENCE's algorithm shape (global depth sort, splat-major back-to-front "over"
 * rasterisation); the tile lists it also produces exist to check the HIP path's integer stages
 * bit-for-bit.  All arithmetic that decides an integer (cull, tile rectangle, support test) uses only
 * + - * / sqrtf fmaf floorf ceilf in a fixed order, so that the HIP kernels reproduce it exactly.
 * Build: see oracle/Makefile (-ffp-contract=off is mandatory).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gsx.h"

#define TILE 16

/* ------------------------------------------------------------------------------------------------
 * Frame constants: everything derived once per (camera, model transform, gaussian transform).
 * spec/RENDER_SPEC.md §3.  The HIP library derives the same struct on the host with the same
 * operation order (csrc/frame_consts.h).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gsxo_frame {
    float T[9];      /* row-major 3x3: W * R_m * diag(s_m)   (model -> view, linear part) */
    float vt[3];     /* W * t_m + view translation */
    float P[16];     /* projection, column-major as given */
    float cam_m[3];  /* camera position in unscaled model space: R_m^T (cam - t_m) */
    float s_m[3];    /* model scale */
    float fx, fy;    /* focal lengths in pixels: P[0][0]*W/2, P[1][1]*H/2 */
    float limx, limy; /* jacobian_clamp * tan(fov/2) */
    float width, height;
    float size2;     /* gaussian_transform.size^2 */
    float k, k2;     /* max_std_dev and its square */
    float low_pass, cull_margin, alpha_max, alpha_min, point_radius;
    uint32_t w_px, h_px, tiles_x, tiles_y;
    uint32_t sh_deg, no_sh0, display_mode;
} gsxo_frame;

static void quat_to_rows(const float q[4], float r[9]) {
    /* glam Mat3::from_quat, written out; r is row-major */
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    r[0] = 1.0f - (yy + zz); r[1] = xy - wz;          r[2] = xz + wy;
    r[3] = xy + wz;          r[4] = 1.0f - (xx + zz); r[5] = yz - wx;
    r[6] = xz - wy;          r[7] = yz + wx;          r[8] = 1.0f - (xx + yy);
}

static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void gsxo_frame_setup(const float view[16], const float proj[16], uint32_t width, uint32_t height,
                      const float m_pos[3], const float m_quat[4], const float m_scale[3], float size,
                      uint32_t display_mode, uint32_t sh_deg, uint32_t no_sh0, const gsx_spec_params* sp,
                      gsxo_frame* f) {
    float R[9], W[9], WR[9];
    quat_to_rows(m_quat, R);
    /* W = upper-left 3x3 of view, row-major: W[r][c] = view[c*4+r] */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) W[r * 3 + c] = view[c * 4 + r];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float col[3] = {R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]};
            WR[r * 3 + c] = dot3(&W[r * 3], col);
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) f->T[r * 3 + c] = WR[r * 3 + c] * m_scale[c];
    for (int r = 0; r < 3; ++r) f->vt[r] = dot3(&W[r * 3], m_pos) + view[12 + r];
    memcpy(f->P, proj, sizeof(float) * 16);
    /* camera world position: cam = -W^T * view_translation */
    float vtr[3] = {view[12], view[13], view[14]};
    float cam[3];
    for (int c = 0; c < 3; ++c) {
        float col[3] = {W[0 * 3 + c], W[1 * 3 + c], W[2 * 3 + c]};
        cam[c] = -dot3(col, vtr);
    }
    float rel[3] = {cam[0] - m_pos[0], cam[1] - m_pos[1], cam[2] - m_pos[2]};
    for (int c = 0; c < 3; ++c) {
        float col[3] = {R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]};
        f->cam_m[c] = dot3(col, rel);
    }
    for (int c = 0; c < 3; ++c) f->s_m[c] = m_scale[c];
    f->width = (float)width;
    f->height = (float)height;
    f->fx = proj[0] * f->width * 0.5f;
    f->fy = proj[5] * f->height * 0.5f;
    f->limx = sp->jacobian_clamp / proj[0];
    f->limy = sp->jacobian_clamp / proj[5];
    f->size2 = size * size;
    f->k = sp->max_std_dev;
    f->k2 = sp->max_std_dev * sp->max_std_dev;
    f->low_pass = sp->low_pass;
    f->cull_margin = sp->cull_margin;
    f->alpha_max = sp->alpha_max;
    f->alpha_min = sp->alpha_min;
    f->point_radius = sp->point_radius;
    f->w_px = width;
    f->h_px = height;
    f->tiles_x = (width + TILE - 1) / TILE;
    f->tiles_y = (height + TILE - 1) / TILE;
    f->sh_deg = sh_deg;
    f->no_sh0 = no_sh0;
    f->display_mode = display_mode;
}

/* ------------------------------------------------------------------------------------------------
 * Gaussian -> pod (update_range, scene.rs:2083-2084): cov3d = (R S)(R S)^T, upper triangle
 * xx,xy,xz,yy,yz,zz.  spec §2.
 * ---------------------------------------------------------------------------------------------- */
void gsxo_convert(const gsx_gaussian* g, uint64_t n, float* pos, uint32_t* color, float* sh, float* cov3d) {
    for (uint64_t i = 0; i < n; ++i) {
        float R[9], M[9];
        quat_to_rows(g[i].rot, R);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M[r * 3 + c] = R[r * 3 + c] * g[i].scale[c];
        float* cv = cov3d + 6 * i;
        cv[0] = dot3(&M[0], &M[0]);
        cv[1] = dot3(&M[0], &M[3]);
        cv[2] = dot3(&M[0], &M[6]);
        cv[3] = dot3(&M[3], &M[3]);
        cv[4] = dot3(&M[3], &M[6]);
        cv[5] = dot3(&M[6], &M[6]);
        pos[3 * i + 0] = g[i].pos[0];
        pos[3 * i + 1] = g[i].pos[1];
        pos[3 * i + 2] = g[i].pos[2];
        color[i] = (uint32_t)g[i].color[0] | ((uint32_t)g[i].color[1] << 8) | ((uint32_t)g[i].color[2] << 16) |
                   ((uint32_t)g[i].color[3] << 24);
        memcpy(sh + 45 * i, g[i].sh, sizeof(float) * 45);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Compressed pods (the reference's Sh{Half,Norm8} / Cov3d Half configs, app.rs:386-418): what the GPU
 * stores is a quantised value and what it computes with is the exact dequantisation, so the oracle
 * round-trips the float pod in place and then runs the ordinary float path.  spec §2b.
 *   f16    : IEEE binary16, round to nearest even, overflow to infinity, subnormals kept.
 *   snorm8 : q = floor(clamp(v,-1,1)*127 + 0.5) as int8, decode max(q/127, -1)  (WGSL unpack4x8snorm).
 * ---------------------------------------------------------------------------------------------- */
static uint16_t f32_to_f16_rne(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t x = v.u, sign = (x >> 16) & 0x8000u, mant = x & 0x7FFFFFu;
    int32_t e = (int32_t)((x >> 23) & 0xFF);
    if (e == 255) return (uint16_t)(sign | 0x7C00u | (mant ? 0x200u : 0));
    int32_t he = e - 127 + 15;
    if (he >= 31) return (uint16_t)(sign | 0x7C00u);
    if (he <= 0) {
        if (he < -10) return (uint16_t)sign;
        mant |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - he);
        uint32_t hm = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t hm = mant >> 13, rem = mant & 0x1FFFu;
    uint32_t h = (uint32_t)(he << 10) | hm;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    union { float f; uint32_t u; } v;
    if (e == 0) {
        if (m == 0) { v.u = sign; return v.f; }
        float r = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
        return sign ? -r : r;
    }
    if (e == 31) { v.u = sign | 0x7F800000u | (m << 13); return v.f; }
    v.u = sign | ((e - 15 + 127) << 23) | (m << 13);
    return v.f;
}

static float snorm8_roundtrip(float x) {
    float c = fminf(fmaxf(x, -1.0f), 1.0f);
    int q = (int)floorf(c * 127.0f + 0.5f);
    return fmaxf((float)(signed char)q * (1.0f / 127.0f), -1.0f);
}

/* in place: sh (45 floats per Gaussian, may be NULL) and cov3d (6 floats per Gaussian) become what the GPU
 * dequantises from a pod of the given kinds */
void gsxo_quantize_roundtrip(int sh_kind, int cov_kind, uint64_t n, float* sh, float* cov3d) {
    if (sh && sh_kind == GSX_SH_HALF)
        for (uint64_t i = 0; i < 45 * n; ++i) sh[i] = f16_to_f32(f32_to_f16_rne(sh[i]));
    if (sh && sh_kind == GSX_SH_NORM8)
        for (uint64_t i = 0; i < 45 * n; ++i) sh[i] = snorm8_roundtrip(sh[i]);
    if (sh && sh_kind == GSX_SH_NONE)
        for (uint64_t i = 0; i < 45 * n; ++i) sh[i] = 0.0f;
    if (cov_kind == GSX_COV3D_HALF)
        for (uint64_t i = 0; i < 6 * n; ++i) cov3d[i] = f16_to_f32(f32_to_f16_rne(cov3d[i]));
}

/* ------------------------------------------------------------------------------------------------
 * Projection of one Gaussian.  spec §4.  Returns 1 if visible.
 * ---------------------------------------------------------------------------------------------- */
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

static int project_one(const gsxo_frame* f, const float p[3], uint32_t color, const float* sh, const float cv[6],
                       uint32_t* key, uint32_t rect[4], float mean2d[2], float conic_op[4], float rgb[3]) {
    /* view space */
    float pv[3];
    for (int r = 0; r < 3; ++r) pv[r] = dot3(&f->T[r * 3], p) + f->vt[r];
    /* clip space: pc = P * [pv,1] */
    const float* P = f->P;
    float xc = ((P[0] * pv[0] + P[4] * pv[1]) + P[8] * pv[2]) + P[12];
    float yc = ((P[1] * pv[0] + P[5] * pv[1]) + P[9] * pv[2]) + P[13];
    float zc = ((P[2] * pv[0] + P[6] * pv[1]) + P[10] * pv[2]) + P[14];
    float wc = ((P[3] * pv[0] + P[7] * pv[1]) + P[11] * pv[2]) + P[15];
    float lim = f->cull_margin * wc;
    if (!(wc > 0.0f)) return 0;
    if (!(xc >= -lim && xc <= lim && yc >= -lim && yc <= lim && zc >= 0.0f && zc <= wc)) return 0;
    float d = -pv[2]; /* view depth along -Z (RH) */
    if (!(d > 0.0f)) return 0;

    /* 2x3 screen Jacobian times T:  A = J * T */
    float inv_d = 1.0f / d;
    float tx = clampf(pv[0] * inv_d, -f->limx, f->limx);
    float ty = clampf(pv[1] * inv_d, -f->limy, f->limy);
    float j00 = f->fx * inv_d, j02 = (f->fx * tx) * inv_d;
    float j11 = -(f->fy * inv_d), j12 = -((f->fy * ty) * inv_d);
    float a0[3], a1[3];
    for (int c = 0; c < 3; ++c) {
        a0[c] = j00 * f->T[0 * 3 + c] + j02 * f->T[2 * 3 + c];
        a1[c] = j11 * f->T[1 * 3 + c] + j12 * f->T[2 * 3 + c];
    }
    /* cov2d = A * Sigma * A^T */
    float s0[3] = {cv[0], cv[1], cv[2]}, s1[3] = {cv[1], cv[3], cv[4]}, s2[3] = {cv[2], cv[4], cv[5]};
    float v0[3] = {dot3(s0, a0), dot3(s1, a0), dot3(s2, a0)};
    float v1[3] = {dot3(s0, a1), dot3(s1, a1), dot3(s2, a1)};
    float ca = dot3(a0, v0), cb = dot3(a1, v0), cc = dot3(a1, v1);
    if (f->display_mode == GSX_DISPLAY_POINT) {
        float rp = f->point_radius / f->k;
        ca = rp * rp - f->low_pass;
        cb = 0.0f;
        cc = rp * rp - f->low_pass;
    }
    ca = (ca + f->low_pass) * f->size2;
    cb = cb * f->size2;
    cc = (cc + f->low_pass) * f->size2;
    float det = ca * cc - cb * cb;
    if (!(det > 0.0f)) return 0;
    float inv_det = 1.0f / det;

    /* screen position in pixels (wgpu viewport: y down, pixel centre at +0.5) */
    float inv_w = 1.0f / wc;
    float ndcx = xc * inv_w, ndcy = yc * inv_w;
    float mx = (ndcx * 0.5f + 0.5f) * f->width;
    float my = (0.5f - ndcy * 0.5f) * f->height;

    /* pixel AABB of the cutoff ellipse, then tile rectangle */
    float ex = f->k * sqrtf(ca), ey = f->k * sqrtf(cc);
    float x0f = ceilf((mx - ex) - 0.5f), x1f = floorf((mx + ex) - 0.5f);
    float y0f = ceilf((my - ey) - 0.5f), y1f = floorf((my + ey) - 0.5f);
    x0f = fmaxf(x0f, 0.0f);
    y0f = fmaxf(y0f, 0.0f);
    x1f = fminf(x1f, f->width - 1.0f);
    y1f = fminf(y1f, f->height - 1.0f);
    if (!(x0f <= x1f && y0f <= y1f)) return 0;
    int x0 = (int)x0f, x1 = (int)x1f, y0 = (int)y0f, y1 = (int)y1f;
    rect[0] = (uint32_t)(x0 / TILE);
    rect[1] = (uint32_t)(y0 / TILE);
    rect[2] = (uint32_t)(x1 / TILE) + 1u;
    rect[3] = (uint32_t)(y1 / TILE) + 1u;

    mean2d[0] = mx;
    mean2d[1] = my;
    conic_op[0] = cc * inv_det;
    conic_op[1] = -(cb * inv_det);
    conic_op[2] = ca * inv_det;
    conic_op[3] = (float)(color >> 24) * (1.0f / 255.0f);

    /* colour: DC from the UNORM8 colour, higher orders from SH, view direction in model space */
    float c[3] = {0.0f, 0.0f, 0.0f};
    if (!f->no_sh0) {
        c[0] = (float)(color & 255u) * (1.0f / 255.0f);
        c[1] = (float)((color >> 8) & 255u) * (1.0f / 255.0f);
        c[2] = (float)((color >> 16) & 255u) * (1.0f / 255.0f);
    }
    if (f->sh_deg > 0 && sh) {
        float dx = f->s_m[0] * p[0] - f->cam_m[0];
        float dy = f->s_m[1] * p[1] - f->cam_m[1];
        float dz = f->s_m[2] * p[2] - f->cam_m[2];
        float len = sqrtf((dx * dx + dy * dy) + dz * dz);
        float il = 1.0f / len;
        float x = dx * il, y = dy * il, z = dz * il;
        for (int ch = 0; ch < 3; ++ch) {
            const float* s = sh + ch; /* s[3*k] = coefficient k, channel ch */
            float r = -SH_C1 * y * s[0] + SH_C1 * z * s[3] - SH_C1 * x * s[6];
            if (f->sh_deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r += SH_C2[0] * xy * s[9] + SH_C2[1] * yz * s[12] + SH_C2[2] * (2.0f * zz - xx - yy) * s[15] +
                     SH_C2[3] * xz * s[18] + SH_C2[4] * (xx - yy) * s[21];
                if (f->sh_deg > 2) {
                    r += SH_C3[0] * y * (3.0f * xx - yy) * s[24] + SH_C3[1] * xy * z * s[27] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * s[30] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[33] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * s[36] + SH_C3[5] * z * (xx - yy) * s[39] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * s[42];
                }
            }
            c[ch] += r;
        }
    }
    rgb[0] = fmaxf(c[0], 0.0f);
    rgb[1] = fmaxf(c[1], 0.0f);
    rgb[2] = fmaxf(c[2], 0.0f);

    union { float f; uint32_t u; } bits;
    bits.f = d;
    *key = bits.u; /* d > 0, so the IEEE bit pattern is monotone in d */
    return 1;
}

/* Projection pass over a model (K1 + the per-instance half of K3).  mask: one bit per Gaussian, 1 = keep
 * (NULL = all kept).  sh may be NULL (Sh None pod).  Outputs have length n; culled entries get
 * key 0xFFFFFFFF and zeroed records.  Returns n_visible. */
uint64_t gsxo_project(const gsxo_frame* f, uint64_t n, const float* pos, const uint32_t* color, const float* sh,
                      const float* cov3d, const uint32_t* mask, uint32_t* key, uint32_t* rect, float* mean2d,
                      float* conic_op, float* rgb) {
    uint64_t nvis = 0;
#pragma omp parallel for reduction(+ : nvis) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        uint32_t k = 0xFFFFFFFFu, r[4] = {0, 0, 0, 0};
        float m[2] = {0, 0}, co[4] = {0, 0, 0, 0}, c[3] = {0, 0, 0};
        int keep = mask ? (int)((mask[i >> 5] >> (i & 31)) & 1u) : 1;
        if (keep && project_one(f, pos + 3 * i, color[i], sh ? sh + 45 * i : NULL, cov3d + 6 * i, &k, r, m, co, c))
            nvis += 1;
        else
            k = 0xFFFFFFFFu;
        key[i] = k;
        memcpy(rect + 4 * i, r, sizeof r);
        memcpy(mean2d + 2 * i, m, sizeof m);
        memcpy(conic_op + 4 * i, co, sizeof co);
        memcpy(rgb + 3 * i, c, sizeof c);
    }
    return nvis;
}

/* ------------------------------------------------------------------------------------------------
 * Depth order (K2): stable LSD radix sort of (key, index), 8-bit digits, 4 passes.  Front-to-back:
 * ascending view depth, ties by Gaussian index.  Returns n_visible (keys != 0xFFFFFFFF).
 * ---------------------------------------------------------------------------------------------- */
uint64_t gsxo_depth_sort(uint64_t n, const uint32_t* key, uint32_t* sorted_idx) {
    uint32_t* k0 = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* k1 = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* v1 = sorted_idx;
    uint64_t nvis = 0;
    for (uint64_t i = 0; i < n; ++i) {
        k0[i] = key[i];
        v0[i] = (uint32_t)i;
        nvis += key[i] != 0xFFFFFFFFu;
    }
    uint32_t *ks = k0, *kd = k1, *vs = v0, *vd = v1;
    for (int pass = 0; pass < 4; ++pass) {
        uint64_t hist[257];
        memset(hist, 0, sizeof hist);
        int sh = pass * 8;
        for (uint64_t i = 0; i < n; ++i) hist[((ks[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t o = hist[(ks[i] >> sh) & 255u]++;
            kd[o] = ks[i];
            vd[o] = vs[i];
        }
        uint32_t* t = ks; ks = kd; kd = t;
        t = vs; vs = vd; vd = t;
    }
    /* after 4 passes the result is back in (k0,v0) */
    if (vs != sorted_idx) memcpy(sorted_idx, vs, sizeof(uint32_t) * n);
    free(k0);
    free(k1);
    free(v0);
    return nvis;
}

/* ------------------------------------------------------------------------------------------------
 * Tile lists (integer stage of the HIP path; the reference has no tiles).  For every tile, the
 * Gaussian indices whose tile rectangle contains it, in front-to-back order.  tile_offsets has
 * tiles+1 entries.  Call with list == NULL to get D only.
 * ---------------------------------------------------------------------------------------------- */
uint64_t gsxo_tile_lists(uint32_t tiles_x, uint32_t tiles_y, uint64_t n_visible, const uint32_t* sorted_idx,
                         const uint32_t* rect, uint32_t* tile_offsets, uint32_t* list) {
    uint64_t tiles = (uint64_t)tiles_x * tiles_y;
    uint64_t* cnt = (uint64_t*)calloc(tiles + 1, sizeof(uint64_t));
    for (uint64_t j = 0; j < n_visible; ++j) {
        const uint32_t* r = rect + 4 * (uint64_t)sorted_idx[j];
        for (uint32_t ty = r[1]; ty < r[3]; ++ty)
            for (uint32_t tx = r[0]; tx < r[2]; ++tx) cnt[(uint64_t)ty * tiles_x + tx + 1]++;
    }
    for (uint64_t t = 0; t < tiles; ++t) cnt[t + 1] += cnt[t];
    uint64_t D = cnt[tiles];
    if (tile_offsets)
        for (uint64_t t = 0; t <= tiles; ++t) tile_offsets[t] = (uint32_t)cnt[t];
    if (list) {
        for (uint64_t j = 0; j < n_visible; ++j) {
            uint32_t idx = sorted_idx[j];
            const uint32_t* r = rect + 4 * (uint64_t)idx;
            for (uint32_t ty = r[1]; ty < r[3]; ++ty)
                for (uint32_t tx = r[0]; tx < r[2]; ++tx) list[cnt[(uint64_t)ty * tiles_x + tx]++] = idx;
        }
    }
    free(cnt);
    return D;
}

/* ------------------------------------------------------------------------------------------------
 * Rasterisation in the reference's shape (K3): splats in sorted order drawn BACK-TO-FRONT with
 * premultiplied "over" into a float (r,g,b,T) framebuffer, each splat over the pixel span of its
 * tile rectangle with the per-pixel support test.  spec §6.  fb: [h][w][4]; on entry it holds
 * whatever is behind this model (far models already painted; initialise to 0,0,0,1).
 * OpenMP parallel over row bands; within a band the blend order is the global sorted order.
 * ---------------------------------------------------------------------------------------------- */
static inline float splat_alpha(const gsxo_frame* f, float px, float py, const float m[2], const float co[4]) {
    float dx = px - m[0], dy = py - m[1];
    float q = fmaf(co[0] * dx, dx, fmaf(co[2] * dy, dy, ((2.0f * co[1]) * dx) * dy));
    if (!(q <= f->k2) || q < 0.0f) return -1.0f;
    float w = (f->display_mode == GSX_DISPLAY_SPLAT) ? expf(-0.5f * q) : 1.0f;
    float a = fminf(f->alpha_max, co[3] * w);
    if (a < f->alpha_min) return -1.0f;
    return a;
}

void gsxo_rasterize(const gsxo_frame* f, uint64_t n_visible, const uint32_t* sorted_idx, const uint32_t* rect,
                    const float* mean2d, const float* conic_op, const float* rgb, float* fb) {
    int H = (int)f->h_px, Wd = (int)f->w_px;
    int bands = (int)f->tiles_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int band = 0; band < bands; ++band) {
        int by0 = band * TILE, by1 = by0 + TILE < H ? by0 + TILE : H;
        for (int64_t j = (int64_t)n_visible - 1; j >= 0; --j) { /* far -> near */
            uint64_t i = sorted_idx[j];
            const uint32_t* r = rect + 4 * i;
            if ((uint32_t)band < r[1] || (uint32_t)band >= r[3]) continue;
            int x0 = (int)r[0] * TILE, x1 = (int)r[2] * TILE < Wd ? (int)r[2] * TILE : Wd;
            const float* m = mean2d + 2 * i;
            const float* co = conic_op + 4 * i;
            const float* c = rgb + 3 * i;
            for (int y = by0; y < by1; ++y) {
                float py = (float)y + 0.5f;
                float* row = fb + ((size_t)y * Wd) * 4;
                for (int x = x0; x < x1; ++x) {
                    float a = splat_alpha(f, (float)x + 0.5f, py, m, co);
                    if (a < 0.0f) continue;
                    float* p = row + 4 * x;
                    float om = 1.0f - a;
                    p[0] = fmaf(a, c[0], om * p[0]);
                    p[1] = fmaf(a, c[1], om * p[1]);
                    p[2] = fmaf(a, c[2], om * p[2]);
                    p[3] = om * p[3];
                }
            }
        }
    }
}

/* Front-to-back tile compositor — the HIP path's algorithm restated on the CPU WITHOUT early
 * termination, used only to localise a parity failure (tile lists vs blending). */
void gsxo_composite_tiles(const gsxo_frame* f, const uint32_t* tile_offsets, const uint32_t* list,
                          const float* mean2d, const float* conic_op, const float* rgb, float* fb) {
    int H = (int)f->h_px, Wd = (int)f->w_px;
    int tiles = (int)(f->tiles_x * f->tiles_y);
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles; ++t) {
        int tx = t % (int)f->tiles_x, ty = t / (int)f->tiles_x;
        for (int y = ty * TILE; y < ty * TILE + TILE && y < H; ++y)
            for (int x = tx * TILE; x < tx * TILE + TILE && x < Wd; ++x) {
                float* p = fb + ((size_t)y * Wd + x) * 4;
                float C0 = p[0], C1 = p[1], C2 = p[2], T = p[3];
                float T0 = T;
                float A0 = 0, A1 = 0, A2 = 0, Tm = 1.0f;
                for (uint32_t e = tile_offsets[t]; e < tile_offsets[t + 1]; ++e) {
                    uint64_t i = list[e];
                    float a = splat_alpha(f, (float)x + 0.5f, (float)y + 0.5f, mean2d + 2 * i, conic_op + 4 * i);
                    if (a < 0.0f) continue;
                    float wgt = Tm * a;
                    A0 = fmaf(wgt, rgb[3 * i + 0], A0);
                    A1 = fmaf(wgt, rgb[3 * i + 1], A1);
                    A2 = fmaf(wgt, rgb[3 * i + 2], A2);
                    Tm = Tm * (1.0f - a);
                }
                /* this model sits IN FRONT of what fb already holds */
                p[0] = fmaf(Tm, C0, A0);
                p[1] = fmaf(Tm, C1, A1);
                p[2] = fmaf(Tm, C2, A2);
                p[3] = Tm * T0;
            }
    }
}

/* Whole frame for one model on top of fb (far models first): project -> sort -> rasterise.
 * Scratch is allocated internally.  Returns n_visible. */
uint64_t gsxo_render_model(const gsxo_frame* f, uint64_t n, const float* pos, const uint32_t* color,
                           const float* sh, const float* cov3d, const uint32_t* mask, float* fb) {
    uint32_t* key = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t* rect = (uint32_t*)malloc(sizeof(uint32_t) * 4 * (n ? n : 1));
    float* mean2d = (float*)malloc(sizeof(float) * 2 * (n ? n : 1));
    float* conic = (float*)malloc(sizeof(float) * 4 * (n ? n : 1));
    float* rgb = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    uint32_t* sorted = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    gsxo_project(f, n, pos, color, sh, cov3d, mask, key, rect, mean2d, conic, rgb);
    uint64_t nvis = gsxo_depth_sort(n, key, sorted);
    gsxo_rasterize(f, nvis, sorted, rect, mean2d, conic, rgb, fb);
    free(key); free(rect); free(mean2d); free(conic); free(rgb); free(sorted);
    return nvis;
}

/* ------------------------------------------------------------------------------------------------
 * Mask evaluation (K5, gs::MaskEvaluator::evaluate, scene.rs:2124-2131): postfix set algebra over box /
 * ellipsoid shapes on the world-space Gaussian position.  words: ceil(n/32) u32, bit = kept.
 * ---------------------------------------------------------------------------------------------- */
void gsxo_mask_evaluate(uint64_t n, const float* pos, const float m_pos[3], const float m_quat[4], const float m_scale[3],
                        const gsx_mask_op* ops, uint32_t n_ops, const gsx_mask_shape* shapes, uint32_t n_shapes,
                        uint32_t* words) {
    float Rm[9], Rs[GSX_MASK_MAX_SHAPES][9];
    quat_to_rows(m_quat, Rm);
    for (uint32_t s = 0; s < n_shapes; ++s) quat_to_rows(shapes[s].quat_xyzw, Rs[s]);
    memset(words, 0, sizeof(uint32_t) * ((n + 31) / 32));
    for (uint64_t i = 0; i < n; ++i) {
        float sp[3] = {m_scale[0] * pos[3 * i], m_scale[1] * pos[3 * i + 1], m_scale[2] * pos[3 * i + 2]};
        float w[3];
        for (int r = 0; r < 3; ++r) w[r] = dot3(&Rm[r * 3], sp) + m_pos[r];
        uint32_t inside = 0;
        for (uint32_t s = 0; s < n_shapes; ++s) {
            float rel[3] = {w[0] - shapes[s].pos[0], w[1] - shapes[s].pos[1], w[2] - shapes[s].pos[2]};
            float q[3];
            for (int c = 0; c < 3; ++c) {
                float col[3] = {Rs[s][0 * 3 + c], Rs[s][1 * 3 + c], Rs[s][2 * 3 + c]};
                q[c] = dot3(col, rel) / shapes[s].scale[c];
            }
            int in = shapes[s].kind == GSX_MASK_BOX ? (fabsf(q[0]) <= 1.0f && fabsf(q[1]) <= 1.0f && fabsf(q[2]) <= 1.0f)
                                                    : ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2] <= 1.0f);
            inside |= (uint32_t)(in ? 1 : 0) << s;
        }
        int stack[64], depth = 0;
        for (uint32_t k = 0; k < n_ops; ++k) {
            uint32_t op = ops[k].opcode;
            if (op == GSX_MASK_OP_SHAPE) stack[depth++] = (int)((inside >> ops[k].arg) & 1u);
            else if (op == GSX_MASK_OP_COMPLEMENT) stack[depth - 1] = !stack[depth - 1];
            else {
                int b = stack[--depth], a = stack[depth - 1];
                stack[depth - 1] = op == GSX_MASK_OP_UNION ? (a | b) : op == GSX_MASK_OP_INTERSECTION ? (a & b)
                                   : op == GSX_MASK_OP_DIFFERENCE ? (a & !b) : (a ^ b);
            }
        }
        int keep = n_ops == 0 ? 1 : stack[0];
        if (keep) words[i >> 5] |= 1u << (i & 31);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Selection, edits, queries (spec §7).  Call sites in the reference: viewer.update_query / update_selection_* /
 * postprocessor.postprocess / gs::query::download (src/tab/scene.rs:785-835, 601-611, 651-657); the arithmetic is the
 * build's [BUILD-SPEC].  Restated here as passes over the projection arrays.
 * ---------------------------------------------------------------------------------------------- */
static void o_rgb_to_hsv(float r, float g, float b, float* h, float* s, float* v) {
    float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), d = mx - mn;
    *v = mx;
    *s = mx > 0.0f ? d / mx : 0.0f;
    float hh;
    if (!(d > 0.0f)) hh = 0.0f;
    else if (mx == r) { hh = (g - b) / d; if (hh < 0.0f) hh += 6.0f; }
    else if (mx == g) hh = (b - r) / d + 2.0f;
    else hh = (r - g) / d + 4.0f;
    *h = hh / 6.0f;
}

static void o_hsv_to_rgb(float h, float s, float v, float* r, float* g, float* b) {
    float k = h * 6.0f, fl = floorf(k), f = k - fl;
    int sec = (int)fl;
    if (sec < 0 || sec > 5) sec = 0;
    float p = v * (1.0f - s), q = v * (1.0f - s * f), t = v * (1.0f - s * (1.0f - f));
    switch (sec) {
        case 0: *r = v; *g = t; *b = p; break;
        case 1: *r = q; *g = v; *b = p; break;
        case 2: *r = p; *g = v; *b = t; break;
        case 3: *r = p; *g = q; *b = v; break;
        case 4: *r = t; *g = p; *b = v; break;
        default: *r = v; *g = p; *b = q; break;
    }
}

/* colour ops 1-5 of spec §7; e has ENABLED */
void gsxo_apply_edit(const gsx_gaussian_edit* e, float rgb[3], float* opacity) {
    float r = rgb[0], g = rgb[1], b = rgb[2];
    if (e->flag & GSX_EDIT_OVERRIDE_COLOR) {
        r = e->color[0]; g = e->color[1]; b = e->color[2];
    } else {
        float h, s, v;
        o_rgb_to_hsv(r, g, b, &h, &s, &v);
        h = h + e->color[0];
        h = h - floorf(h);
        s = clampf(s * e->color[1], 0.0f, 1.0f);
        v = v * e->color[2];
        o_hsv_to_rgb(h, s, v, &r, &g, &b);
    }
    if (e->contrast != 0.0f) {
        float c = 1.0f + e->contrast;
        r = (r - 0.5f) * c + 0.5f; g = (g - 0.5f) * c + 0.5f; b = (b - 0.5f) * c + 0.5f;
    }
    if (e->exposure != 0.0f) {
        float m = exp2f(e->exposure);
        r *= m; g *= m; b *= m;
    }
    r = fmaxf(r, 0.0f); g = fmaxf(g, 0.0f); b = fmaxf(b, 0.0f);
    if (e->gamma != 1.0f) { r = powf(r, e->gamma); g = powf(g, e->gamma); b = powf(b, e->gamma); }
    rgb[0] = r; rgb[1] = g; rgb[2] = b;
    *opacity = clampf(*opacity * e->alpha, 0.0f, 1.0f);
}

/* Preprocess-time edit handling over a finished projection: persists the selection edit into the selected
 * Gaussians' records (visible or not), culls HIDDEN ones, applies colour ops and the highlight.  selection may be
 * NULL; edits has n records (in/out).  Returns the new n_visible. */
uint64_t gsxo_edit_pass(uint64_t n, const uint32_t* selection, gsx_gaussian_edit* edits, const gsx_gaussian_edit* sel_edit,
                        const float highlight[4], uint32_t* key, uint32_t* rect, float* mean2d, float* conic_op,
                        float* rgb) {
    uint64_t nvis = 0;
    for (uint64_t i = 0; i < n; ++i) {
        int sel = selection ? (int)((selection[i >> 5] >> (i & 31)) & 1u) : 0;
        if (sel && (sel_edit->flag & GSX_EDIT_ENABLED)) edits[i] = *sel_edit;
        const gsx_gaussian_edit* e = &edits[i];
        if (key[i] == 0xFFFFFFFFu) continue;
        if ((e->flag & GSX_EDIT_ENABLED) && (e->flag & GSX_EDIT_HIDDEN)) {
            key[i] = 0xFFFFFFFFu;
            memset(rect + 4 * i, 0, 16); memset(mean2d + 2 * i, 0, 8); memset(conic_op + 4 * i, 0, 16); memset(rgb + 3 * i, 0, 12);
            continue;
        }
        if (e->flag & GSX_EDIT_ENABLED) gsxo_apply_edit(e, rgb + 3 * i, conic_op + 4 * i + 3);
        if (sel && highlight[3] > 0.0f)
            for (int c = 0; c < 3; ++c) rgb[3 * i + c] = rgb[3 * i + c] + (highlight[c] - rgb[3 * i + c]) * highlight[3];
        nvis += 1;
    }
    return nvis;
}

/* Rect / Brush / Texture query -> one flag bit per Gaussian */
void gsxo_query_flags(uint64_t n, const uint32_t* key, const float* mean2d, const gsx_query* q, const uint8_t* texture,
                      uint32_t tex_w, uint32_t tex_h, uint32_t* flags) {
    memset(flags, 0, 4 * ((n + 31) / 32));
    for (uint64_t i = 0; i < n; ++i) {
        if (key[i] == 0xFFFFFFFFu) continue;
        float mx = mean2d[2 * i], my = mean2d[2 * i + 1];
        int f = 0;
        if (q->kind == GSX_QUERY_RECT) {
            float x0 = fminf(q->p0[0], q->p1[0]), x1 = fmaxf(q->p0[0], q->p1[0]);
            float y0 = fminf(q->p0[1], q->p1[1]), y1 = fmaxf(q->p0[1], q->p1[1]);
            f = mx >= x0 && mx <= x1 && my >= y0 && my <= y1;
        } else if (q->kind == GSX_QUERY_BRUSH) {
            float ax = q->p0[0], ay = q->p0[1], dx = q->p1[0] - ax, dy = q->p1[1] - ay;
            float len2 = dx * dx + dy * dy, t = 0.0f;
            if (len2 > 0.0f) t = clampf(((mx - ax) * dx + (my - ay) * dy) / len2, 0.0f, 1.0f);
            float ex = mx - (ax + t * dx), ey = my - (ay + t * dy);
            f = ex * ex + ey * ey <= q->radius * q->radius;
        } else if (q->kind == GSX_QUERY_TEXTURE) {
            float fx = floorf(mx), fy = floorf(my);
            if (texture && fx >= 0.0f && fy >= 0.0f && fx < (float)tex_w && fy < (float)tex_h)
                f = texture[(size_t)fy * tex_w + (size_t)fx] != 0;
        }
        if (f) flags[i >> 5] |= 1u << (i & 31);
    }
}

static int hit_cmp(const void* a, const void* b) {
    const gsx_query_hit *x = (const gsx_query_hit*)a, *y = (const gsx_query_hit*)b;
    if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
    return x->index < y->index ? -1 : (x->index > y->index ? 1 : 0);
}

/* Hit query at p0: (index, view depth, alpha) of every visible Gaussian covering the point, sorted by (depth, index) */
uint64_t gsxo_query_hits(const gsxo_frame* f, uint64_t n, const uint32_t* key, const float* mean2d, const float* conic_op,
                         const float p0[2], gsx_query_hit* out, uint64_t capacity) {
    uint64_t cnt = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (key[i] == 0xFFFFFFFFu) continue;
        const float* co = conic_op + 4 * i;
        float dx = p0[0] - mean2d[2 * i], dy = p0[1] - mean2d[2 * i + 1];
        float q = fmaf(co[0] * dx, dx, fmaf(co[2] * dy, dy, ((2.0f * co[1]) * dx) * dy));
        if (!(q <= f->k2) || q < 0.0f) continue;
        float w = (f->display_mode == GSX_DISPLAY_SPLAT) ? expf(-0.5f * q) : 1.0f;
        float a = fminf(f->alpha_max, co[3] * w);
        if (!(a >= 1.0f / 255.0f)) continue;
        if (cnt < capacity) {
            union { uint32_t u; float f; } d;
            d.u = key[i];
            out[cnt].index = (uint32_t)i; out[cnt].depth = d.f; out[cnt].alpha = a; out[cnt].reserved = 0;
        }
        cnt += 1;
    }
    qsort(out, cnt < capacity ? cnt : capacity, sizeof *out, hit_cmp);
    return cnt;
}

/* K4: selection = op(selection, flags) */
void gsxo_selection_op(uint64_t n_words, uint32_t op, const uint32_t* flags, uint32_t* selection) {
    for (uint64_t w = 0; w < n_words; ++w)
        selection[w] = op == GSX_SELECTION_SET ? flags[w] : (op == GSX_SELECTION_ADD ? (selection[w] | flags[w]) : (selection[w] & ~flags[w]));
}

int gsxo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

size_t gsxo_frame_sizeof(void) { return sizeof(gsxo_frame); }
