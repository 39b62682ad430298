"""float64 NumPy statement of spec/RENDER_SPEC.md, written from the mathematics rather than from the
float32 operation order of gsx_oracle.c.  TEST INFRASTRUCTURE ONLY — it validates the C oracle and
generates tests/golden/*.npz; nothing on the product path imports it.

PARITY UNPINNED (see gsx_oracle.c): there is no executable reference and no reference golden vector
for this path; this file is the independent second opinion on the written spec.

Differences from gsx_oracle.c that make it an independent check:
  * world-space formulation: p_w = R_m (s_m * p) + t_m, Sigma_w = M Sigma M^T, then the view and the
    full perspective Jacobian; the C oracle folds model and view into one 3x3;
  * SH direction computed in world space and rotated back by R_m^T;
  * support = every pixel of the frame with d^T Sigma^-1 d <= k^2 (no tile rectangles);
  * compositing front-to-back per pixel in exact depth order (ties by index), no early termination.
"""
from __future__ import annotations

import numpy as np

SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)

DEFAULT_PARAMS = dict(max_std_dev=3.0, cull_margin=1.3, jacobian_clamp=1.3, low_pass=0.3, alpha_max=1.0,
                      alpha_min=0.0, point_radius=2.0)


def quat_to_mat(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float64)


def cov3d_from_gaussians(rot_xyzw, scale):
    """Sigma = (R S)(R S)^T as 6 upper-triangular values xx,xy,xz,yy,yz,zz."""
    n = rot_xyzw.shape[0]
    out = np.empty((n, 6), np.float64)
    for i in range(n):
        m = quat_to_mat(rot_xyzw[i]) * np.asarray(scale[i], np.float64)[None, :]
        s = m @ m.T
        out[i] = (s[0, 0], s[0, 1], s[0, 2], s[1, 1], s[1, 2], s[2, 2])
    return out


def sh_color(dirs, sh, deg):
    """Higher-order SH (degree 1..deg) for unit directions dirs[n,3], sh[n,15,3] -> [n,3]."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = np.zeros((dirs.shape[0], 3))
    if deg > 0:
        r += -SH_C1 * y * sh[:, 0] + SH_C1 * z * sh[:, 1] - SH_C1 * x * sh[:, 2]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r += (SH_C2[0] * xy * sh[:, 3] + SH_C2[1] * yz * sh[:, 4] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 5]
              + SH_C2[3] * xz * sh[:, 6] + SH_C2[4] * (xx - yy) * sh[:, 7])
    if deg > 2:
        r += (SH_C3[0] * y * (3 * xx - yy) * sh[:, 8] + SH_C3[1] * xy * z * sh[:, 9]
              + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 10] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 11]
              + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 12] + SH_C3[5] * z * (xx - yy) * sh[:, 13]
              + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 14])
    return r


def quantise_pod(sh, cov3d, sh_kind=0, cov_kind=0):
    """spec 2b: what a compressed pod holds, i.e. the EXACT dequantisation of the stored value, as float32 arrays.
    sh_kind 0 Single | 1 Half (binary16, round to nearest even) | 2 Norm8 (snorm8) | 3 None; cov_kind 0 Single | 1 Half.
    Written with numpy's own float16 conversion, not with the oracle's bit twiddling."""
    sh = np.asarray(sh, np.float32)
    cov = np.asarray(cov3d, np.float32)
    if sh_kind == 1:
        with np.errstate(over="ignore"):
            sh = sh.astype(np.float16).astype(np.float32)
    elif sh_kind == 2:
        q = np.floor(np.clip(sh.astype(np.float64), -1.0, 1.0) * 127.0 + 0.5)
        # the decode is ONE float32 multiply by the float32 constant 1/127 (unpack4x8snorm), then max(-1)
        sh = np.maximum(q.astype(np.float32) * np.float32(1.0 / 127.0), np.float32(-1.0)).astype(np.float32)
    elif sh_kind == 3:
        sh = np.zeros_like(sh)
    if cov_kind == 1:
        with np.errstate(over="ignore"):
            cov = cov.astype(np.float16).astype(np.float32)
    return sh, cov


def mask_evaluate(pos, tree, shapes, m_pos=(0, 0, 0), m_quat=(0, 0, 0, 1), m_scale=(1, 1, 1)):
    """spec 2c in float64.  tree: nested tuples ("shape", i) | ("not", a) | (op, a, b) with op in "|&-^" (mask.MaskOp.tree);
    shapes: list of dict(kind 0 box | 1 ellipsoid, pos, quat, scale).  Returns (bool[n] kept, float[n] margin) — margin =
    the smallest distance of any shape's decision value from its boundary, so a fixture can assert that no Gaussian sits
    within float32 rounding of a boundary."""
    pos = np.asarray(pos, np.float64)
    pw = (pos * np.asarray(m_scale, np.float64)) @ quat_to_mat(m_quat).T + np.asarray(m_pos, np.float64)
    margin = np.full(pos.shape[0], np.inf)
    inside = []
    for sh in shapes:
        q = ((pw - np.asarray(sh["pos"], np.float64)) @ quat_to_mat(sh["quat"])) / np.asarray(sh["scale"], np.float64)
        if int(sh["kind"]) == 0:
            val = np.abs(q).max(axis=1)
        else:
            val = (q * q).sum(axis=1)
        inside.append(val <= 1.0)
        margin = np.minimum(margin, np.abs(val - 1.0))

    def ev(t):
        if t[0] == "shape":
            return inside[t[1]]
        if t[0] == "not":
            return ~ev(t[1])
        a, b = ev(t[1]), ev(t[2])
        return {"|": a | b, "&": a & b, "-": a & ~b, "^": a ^ b}[t[0]]

    return ev(tree), margin


def mask_words(bits):
    """bool[n] -> uint32 words, bit i of word i >> 5 (tail bits of the last word set: they belong to no Gaussian)."""
    n = bits.shape[0]
    w = np.zeros((n + 31) // 32, np.uint32)
    idx = np.nonzero(bits)[0]
    np.bitwise_or.at(w, idx >> 5, (np.uint32(1) << (idx & 31).astype(np.uint32)))
    if n & 31:
        w[-1] |= np.uint32((0xFFFFFFFF << (n & 31)) & 0xFFFFFFFF)
    return w


def edit_colour_ops(rgb, opacity, edit):
    """spec 7 colour ops in float64 for ONE edit record applied to arrays rgb[n,3], opacity[n].  edit: dict(flag, color,
    contrast, exposure, gamma, alpha).  HSV by the textbook piecewise formulas (not the oracle's branch order)."""
    rgb = np.asarray(rgb, np.float64).copy()
    flag = int(edit["flag"])
    c = [float(x) for x in edit["color"]]
    if flag & 4:
        rgb[:] = c
    else:
        mx, mn = rgb.max(axis=1), rgb.min(axis=1)
        d = mx - mn
        r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        with np.errstate(all="ignore"):
            h = np.where(d == 0, 0.0, np.where(mx == r, ((g - b) / d) % 6.0, np.where(mx == g, (b - r) / d + 2.0, (r - g) / d + 4.0))) / 6.0
            s = np.where(mx > 0, d / mx, 0.0)
        h = (h + c[0]) % 1.0
        s = np.clip(s * c[1], 0.0, 1.0)
        v = mx * c[2]
        k = lambda n_: (n_ + h * 6.0) % 6.0  # noqa: E731
        f = lambda n_: v - v * s * np.clip(np.minimum(k(n_), 4.0 - k(n_)), 0.0, 1.0)  # noqa: E731
        rgb = np.stack([f(5.0), f(3.0), f(1.0)], 1)
    if float(edit["contrast"]) != 0:
        rgb = (rgb - 0.5) * (1.0 + float(edit["contrast"])) + 0.5
    if float(edit["exposure"]) != 0:
        rgb = rgb * 2.0 ** float(edit["exposure"])
    rgb = np.maximum(rgb, 0.0)
    if float(edit["gamma"]) != 1:
        rgb = rgb ** float(edit["gamma"])
    return rgb, np.clip(np.asarray(opacity, np.float64) * float(edit["alpha"]), 0.0, 1.0)


def project(view, proj, width, height, pos, color_u32, sh, cov3d, m_pos=(0, 0, 0), m_quat=(0, 0, 0, 1),
            m_scale=(1, 1, 1), size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None, mask=None,
            selection=None, sel_edit=None, highlight=None):
    """Per-Gaussian projection in float64.  Returns dict(visible, depth, mean2d, cov2d, conic, opacity, rgb).
    selection (bool[n]) + sel_edit (dict, spec 7): the selected Gaussians carry that edit (HIDDEN culls them, otherwise the
    colour ops apply); highlight (r, g, b, a): blended over the selected Gaussians' colour."""
    P_ = dict(DEFAULT_PARAMS)
    P_.update(params or {})
    k = P_["max_std_dev"]
    V = np.asarray(view, np.float64).reshape(4, 4).T  # column-major flat -> math matrix
    P = np.asarray(proj, np.float64).reshape(4, 4).T
    Rm = quat_to_mat(m_quat)
    sm = np.asarray(m_scale, np.float64)
    tm = np.asarray(m_pos, np.float64)
    pos = np.asarray(pos, np.float64)
    n = pos.shape[0]
    pw = (pos * sm) @ Rm.T + tm
    pv = pw @ V[:3, :3].T + V[:3, 3]
    pc = np.concatenate([pv, np.ones((n, 1))], 1) @ P.T
    w = pc[:, 3]
    m = P_["cull_margin"]
    with np.errstate(all="ignore"):
        vis = (w > 0) & (np.abs(pc[:, 0]) <= m * w) & (np.abs(pc[:, 1]) <= m * w) & (pc[:, 2] >= 0) & (pc[:, 2] <= w)
        d = -pv[:, 2]
        vis &= d > 0
        if mask is not None:
            bits = (np.asarray(mask, np.uint32)[np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1
            vis &= bits.astype(bool)
        # world covariance, then view
        S = np.zeros((n, 3, 3))
        c = np.asarray(cov3d, np.float64)
        S[:, 0, 0], S[:, 0, 1], S[:, 0, 2] = c[:, 0], c[:, 1], c[:, 2]
        S[:, 1, 0], S[:, 1, 1], S[:, 1, 2] = c[:, 1], c[:, 3], c[:, 4]
        S[:, 2, 0], S[:, 2, 1], S[:, 2, 2] = c[:, 2], c[:, 4], c[:, 5]
        M = Rm * sm[None, :]
        Sw = M @ S @ M.T
        W3 = V[:3, :3]
        Sv = W3 @ Sw @ W3.T
        fx, fy = P[0, 0] * width / 2.0, P[1, 1] * height / 2.0
        limx, limy = P_["jacobian_clamp"] / P[0, 0], P_["jacobian_clamp"] / P[1, 1]
        tx = np.clip(pv[:, 0] / d, -limx, limx)
        ty = np.clip(pv[:, 1] / d, -limy, limy)
        # screen x = fx * x_v / d + cx ; screen y (down) = -fy * y_v / d + cy ; d = -z_v
        J = np.zeros((n, 2, 3))
        J[:, 0, 0] = fx / d
        J[:, 0, 2] = fx * tx / d
        J[:, 1, 1] = -fy / d
        J[:, 1, 2] = -fy * ty / d
        C2 = J @ Sv @ J.transpose(0, 2, 1)
        a, b, cc = C2[:, 0, 0], C2[:, 0, 1], C2[:, 1, 1]
        if display_mode == 2:
            rp = P_["point_radius"] / k
            a = np.full(n, rp * rp - P_["low_pass"])
            b = np.zeros(n)
            cc = np.full(n, rp * rp - P_["low_pass"])
        s2 = float(size) ** 2
        a = (a + P_["low_pass"]) * s2
        b = b * s2
        cc = (cc + P_["low_pass"]) * s2
        det = a * cc - b * b
        vis &= det > 0
        ndc = pc[:, :2] / w[:, None]
        mx = (ndc[:, 0] * 0.5 + 0.5) * width
        my = (0.5 - ndc[:, 1] * 0.5) * height
        ex, ey = k * np.sqrt(np.abs(a)), k * np.sqrt(np.abs(cc))
        # at least one pixel centre inside the AABB of the cutoff ellipse, clipped to the frame
        x0 = np.maximum(np.ceil(mx - ex - 0.5), 0)
        x1 = np.minimum(np.floor(mx + ex - 0.5), width - 1)
        y0 = np.maximum(np.ceil(my - ey - 0.5), 0)
        y1 = np.minimum(np.floor(my + ey - 0.5), height - 1)
        vis &= (x0 <= x1) & (y0 <= y1)
        conic = np.stack([cc / det, -b / det, a / det], 1)
    col = np.asarray(color_u32, np.uint32)
    opacity = (col >> 24).astype(np.float64) / 255.0
    rgb = np.zeros((n, 3))
    if not no_sh0:
        rgb = np.stack([(col & 255), (col >> 8) & 255, (col >> 16) & 255], 1).astype(np.float64) / 255.0
    if sh_deg > 0 and sh is not None:
        cam = -V[:3, :3].T @ V[:3, 3]
        dw = pw - cam
        dw = dw / np.linalg.norm(dw, axis=1, keepdims=True)
        dm = dw @ Rm  # R_m^T applied to each direction
        rgb = rgb + sh_color(dm, np.asarray(sh, np.float64).reshape(n, 15, 3), sh_deg)
    rgb = np.maximum(rgb, 0.0)
    if selection is not None:
        sel = np.asarray(selection, bool)
        if sel_edit is not None and int(sel_edit["flag"]) & 1:
            if int(sel_edit["flag"]) & 2:
                vis = vis & ~sel
            else:
                e_rgb, e_op = edit_colour_ops(rgb[sel], opacity[sel], sel_edit)
                rgb[sel] = e_rgb
                opacity[sel] = e_op
        if highlight is not None and float(highlight[3]) > 0:
            hl = np.asarray(highlight, np.float64)
            rgb[sel] = rgb[sel] + (hl[:3] - rgb[sel]) * hl[3]
    return dict(visible=vis, depth=d, mean2d=np.stack([mx, my], 1), cov2d=np.stack([a, b, cc], 1), conic=conic,
                opacity=opacity, rgb=rgb, pix_aabb=np.stack([x0, y0, x1, y1], 1))


def render(view, proj, width, height, models, size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None, ambiguity_tol=None):
    """Full frame.  ``models`` = list of dicts(pos,color,sh,cov3d[,m_pos,m_quat,m_scale,mask]) in the
    reference's paint order FAR -> NEAR (src/tab/scene.rs:533-558).  Returns float64 [H,W,4] =
    premultiplied rgb + transmittance.

    ambiguity_tol (optional): also returns, per pixel, how much of the frame hangs on support decisions `q <= k^2` that are
    closer than ``ambiguity_tol`` to the cut.  The cut is a discontinuity of the spec itself: a contribution of
    opacity x exp(-k^2 / 2) (1.1 % at k = 3) is in or out, and an implementation in another precision may decide a pair that
    close the other way.  Per such pair the pixel's allowance grows by 2 x T x alpha x max(1, |rgb|) — its own contribution plus
    what the changed transmittance does to everything behind it."""
    P_ = dict(DEFAULT_PARAMS)
    P_.update(params or {})
    k2 = P_["max_std_dev"] ** 2
    C = np.zeros((height, width, 3))
    T = np.ones((height, width))
    amb = np.zeros((height, width)) if ambiguity_tol is not None else None
    ys, xs = np.mgrid[0:height, 0:width]
    px, py = xs + 0.5, ys + 0.5
    for mdl in reversed(models):  # front-to-back across models: nearest model first
        pr = project(view, proj, width, height, mdl["pos"], mdl["color"], mdl.get("sh"), mdl["cov3d"],
                     mdl.get("m_pos", (0, 0, 0)), mdl.get("m_quat", (0, 0, 0, 1)), mdl.get("m_scale", (1, 1, 1)),
                     size, display_mode, sh_deg, no_sh0, params, mdl.get("mask"), mdl.get("selection"), mdl.get("sel_edit"),
                     mdl.get("highlight"))
        idx = np.nonzero(pr["visible"])[0]
        # float32 depth key order (the key is the f32 bit pattern of d), ties by index
        order = idx[np.lexsort((idx, pr["depth"][idx].astype(np.float32)))]
        for i in order:
            mx, my = pr["mean2d"][i]
            a, b, c = pr["cov2d"][i]
            ex, ey = np.sqrt(k2 * a), np.sqrt(k2 * c)
            x0, x1 = int(max(np.floor(mx - ex - 1), 0)), int(min(np.ceil(mx + ex + 1), width))
            y0, y1 = int(max(np.floor(my - ey - 1), 0)), int(min(np.ceil(my + ey + 1), height))
            if x0 >= x1 or y0 >= y1:
                continue
            dx, dy = px[y0:y1, x0:x1] - mx, py[y0:y1, x0:x1] - my
            ca, cb, cc = pr["conic"][i]
            q = ca * dx * dx + cc * dy * dy + 2 * cb * dx * dy
            wgt = np.exp(-0.5 * q) if display_mode == 0 else np.ones_like(q)
            al = np.minimum(P_["alpha_max"], pr["opacity"][i] * wgt)
            al = np.where((q <= k2) & (al >= P_["alpha_min"]), al, 0.0)
            Tl = T[y0:y1, x0:x1]
            if amb is not None:
                edge = np.abs(q - k2) < ambiguity_tol
                if edge.any():
                    a_edge = np.minimum(P_["alpha_max"], pr["opacity"][i] * wgt)
                    amb[y0:y1, x0:x1] += np.where(edge, 2.0 * Tl * a_edge * max(1.0, float(np.abs(pr["rgb"][i]).max())), 0.0)
            C[y0:y1, x0:x1] += (Tl * al)[..., None] * pr["rgb"][i]
            T[y0:y1, x0:x1] = Tl * (1 - al)
    frame = np.concatenate([C, T[..., None]], 2)
    return (frame, amb) if amb is not None else frame


def alpha_min_margin(view, proj, width, height, models, size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None):
    """Smallest relative distance |alpha / alpha_min - 1| over the contributions that matter (T x colour above 1e-4 at that
    pixel) of the frame ``render`` would produce: how safely a float32 implementation reproduces the `alpha >= alpha_min`
    decisions (tests/golden/make_golden.py keeps a fixture only if this is comfortably above float32 rounding)."""
    P_ = dict(DEFAULT_PARAMS)
    P_.update(params or {})
    k2 = P_["max_std_dev"] ** 2
    T = np.ones((height, width))
    ys, xs = np.mgrid[0:height, 0:width]
    px, py = xs + 0.5, ys + 0.5
    best = np.inf
    for mdl in reversed(models):
        pr = project(view, proj, width, height, mdl["pos"], mdl["color"], mdl.get("sh"), mdl["cov3d"],
                     mdl.get("m_pos", (0, 0, 0)), mdl.get("m_quat", (0, 0, 0, 1)), mdl.get("m_scale", (1, 1, 1)),
                     size, display_mode, sh_deg, no_sh0, params, mdl.get("mask"), mdl.get("selection"), mdl.get("sel_edit"),
                     mdl.get("highlight"))
        idx = np.nonzero(pr["visible"])[0]
        order = idx[np.lexsort((idx, pr["depth"][idx].astype(np.float32)))]
        for i in order:
            mx, my = pr["mean2d"][i]
            a, b, c = pr["cov2d"][i]
            ex, ey = np.sqrt(k2 * a), np.sqrt(k2 * c)
            x0, x1 = int(max(np.floor(mx - ex - 1), 0)), int(min(np.ceil(mx + ex + 1), width))
            y0, y1 = int(max(np.floor(my - ey - 1), 0)), int(min(np.ceil(my + ey + 1), height))
            if x0 >= x1 or y0 >= y1:
                continue
            dx, dy = px[y0:y1, x0:x1] - mx, py[y0:y1, x0:x1] - my
            ca, cb, cc = pr["conic"][i]
            q = ca * dx * dx + cc * dy * dy + 2 * cb * dx * dy
            wgt = np.exp(-0.5 * q) if display_mode == 0 else np.ones_like(q)
            raw = np.minimum(P_["alpha_max"], pr["opacity"][i] * wgt)
            Tl = T[y0:y1, x0:x1]
            matters = (q <= k2) & (Tl * max(float(np.abs(pr["rgb"][i]).max()), 1.0) * P_["alpha_min"] > 1e-4)
            if matters.any():
                best = min(best, float(np.abs(raw[matters] / P_["alpha_min"] - 1.0).min()))
            al = np.where((q <= k2) & (raw >= P_["alpha_min"]), raw, 0.0)
            T[y0:y1, x0:x1] = Tl * (1 - al)
    return best
