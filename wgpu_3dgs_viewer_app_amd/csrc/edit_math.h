// edit_math.h — colour ops of a GaussianEditPod and the query predicates (spec/RENDER_SPEC.md §7), shared by the
// device kernels (kernels_edit.hip) and the host-side PLY export (gsx_ply.cpp).  [BUILD-SPEC]: the app only
// builds the pod (app.rs:1546-1564) and hands it to the crate; what the crate's shader does with it is not in the tree.
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/gsx.h"

#if defined(__HIPCC__)
#define GSX_HD __host__ __device__
#else
#define GSX_HD
#endif

namespace gsx {

GSX_HD inline float em_clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

GSX_HD inline void em_rgb_to_hsv(float r, float g, float b, float& h, float& s, float& v) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b)), d = mx - mn;
    v = mx;
    s = mx > 0.0f ? d / mx : 0.0f;
    if (!(d > 0.0f)) {
        h = 0.0f;
    } else if (mx == r) {
        h = (g - b) / d;
        if (h < 0.0f) h += 6.0f;
    } else if (mx == g) {
        h = (b - r) / d + 2.0f;
    } else {
        h = (r - g) / d + 4.0f;
    }
    h = h / 6.0f;
}

GSX_HD inline void em_hsv_to_rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float k = h * 6.0f, fl = floorf(k), f = k - fl;
    int sec = (int)fl;
    if (sec < 0 || sec > 5) sec = 0;
    const float p = v * (1.0f - s), q = v * (1.0f - s * f), t = v * (1.0f - s * (1.0f - f));
    switch (sec) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// colour ops 1-5 of spec §7 on a surviving Gaussian's colour and opacity; e.flag has ENABLED
GSX_HD inline void em_apply_edit(const gsx_gaussian_edit& e, float& r, float& g, float& b, float& opacity) {
    if (e.flag & GSX_EDIT_OVERRIDE_COLOR) {
        r = e.color[0];
        g = e.color[1];
        b = e.color[2];
    } else {
        float h, s, v;
        em_rgb_to_hsv(r, g, b, h, s, v);
        h = h + e.color[0];
        h = h - floorf(h);
        s = em_clamp(s * e.color[1], 0.0f, 1.0f);
        v = v * e.color[2];
        em_hsv_to_rgb(h, s, v, r, g, b);
    }
    if (e.contrast != 0.0f) {
        const float c = 1.0f + e.contrast;
        r = (r - 0.5f) * c + 0.5f;
        g = (g - 0.5f) * c + 0.5f;
        b = (b - 0.5f) * c + 0.5f;
    }
    if (e.exposure != 0.0f) {
        const float m = exp2f(e.exposure);
        r *= m;
        g *= m;
        b *= m;
    }
    r = fmaxf(r, 0.0f);
    g = fmaxf(g, 0.0f);
    b = fmaxf(b, 0.0f);
    if (e.gamma != 1.0f) {
        r = powf(r, e.gamma);
        g = powf(g, e.gamma);
        b = powf(b, e.gamma);
    }
    opacity = em_clamp(opacity * e.alpha, 0.0f, 1.0f);
}

// Rect / Brush predicates on a screen mean (px)
GSX_HD inline bool em_in_rect(float mx, float my, const gsx_query& q) {
    const float x0 = fminf(q.p0[0], q.p1[0]), x1 = fmaxf(q.p0[0], q.p1[0]);
    const float y0 = fminf(q.p0[1], q.p1[1]), y1 = fmaxf(q.p0[1], q.p1[1]);
    return mx >= x0 && mx <= x1 && my >= y0 && my <= y1;
}

GSX_HD inline bool em_in_brush(float mx, float my, const gsx_query& q) {
    const float ax = q.p0[0], ay = q.p0[1], dx = q.p1[0] - ax, dy = q.p1[1] - ay;
    const float len2 = dx * dx + dy * dy;
    float t = 0.0f;
    if (len2 > 0.0f) t = em_clamp(((mx - ax) * dx + (my - ay) * dy) / len2, 0.0f, 1.0f);
    const float ex = mx - (ax + t * dx), ey = my - (ay + t * dy);
    return ex * ex + ey * ey <= q.radius * q.radius;
}

}  // namespace gsx
