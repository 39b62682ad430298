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


def project(view, proj, width, height, pos, color_u32, sh, cov3d, m_pos=(0, 0, 0), m_quat=(0, 0, 0, 1),
            m_scale=(1, 1, 1), size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None, mask=None):
    """Per-Gaussian projection in float64.  Returns dict(visible, depth, mean2d, cov2d, conic, opacity, rgb)."""
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
    return dict(visible=vis, depth=d, mean2d=np.stack([mx, my], 1), cov2d=np.stack([a, b, cc], 1), conic=conic,
                opacity=opacity, rgb=rgb, pix_aabb=np.stack([x0, y0, x1, y1], 1))


def render(view, proj, width, height, models, size=1.0, display_mode=0, sh_deg=3, no_sh0=0, params=None):
    """Full frame.  ``models`` = list of dicts(pos,color,sh,cov3d[,m_pos,m_quat,m_scale,mask]) in the
    reference's paint order FAR -> NEAR (src/tab/scene.rs:533-558).  Returns float64 [H,W,4] =
    premultiplied rgb + transmittance."""
    P_ = dict(DEFAULT_PARAMS)
    P_.update(params or {})
    k2 = P_["max_std_dev"] ** 2
    C = np.zeros((height, width, 3))
    T = np.ones((height, width))
    ys, xs = np.mgrid[0:height, 0:width]
    px, py = xs + 0.5, ys + 0.5
    for mdl in reversed(models):  # front-to-back across models: nearest model first
        pr = project(view, proj, width, height, mdl["pos"], mdl["color"], mdl.get("sh"), mdl["cov3d"],
                     mdl.get("m_pos", (0, 0, 0)), mdl.get("m_quat", (0, 0, 0, 1)), mdl.get("m_scale", (1, 1, 1)),
                     size, display_mode, sh_deg, no_sh0, params, mdl.get("mask"))
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
            C[y0:y1, x0:x1] += (Tl * al)[..., None] * pr["rgb"][i]
            T[y0:y1, x0:x1] = Tl * (1 - al)
    return np.concatenate([C, T[..., None]], 2)
