#!/usr/bin/env python3
"""One command to pin the constants the reference tree does not pin (SURVEY.md 8c [BUILD-SPEC]).

The arithmetic of the render path lives in the crate wgpu-3dgs-viewer 0.2.0 (/root/reference/Cargo.toml:25-31, Cargo.lock:3731-3746),
which is not in this image: support cutoff k, low-pass, cull margin, Jacobian clamp, alpha_max / alpha_min are named constants
of spec/RENDER_SPEC.md and fields of gsx_spec_params.  A maintainer who HAS the crate renders ONE frame with it — any PLY, any
camera — saves it (PNG screenshot or float .npz) and runs

    python tools/reconcile.py --ply scene.ply --view "<16 floats>" --proj "<16 floats>" --size 1920x1080 --image ref.png \\
                              [--background 0,0,0] [--sh-degree 3] [--out reconcile_out]

(view / proj: column-major as glam stores them, app.rs:1236-1244; or .npy files).  The tool renders the same PLY and camera through
libgsx over a grid of gsx_spec_params — coordinate descent, two sweeps: each parameter in turn over its candidates, the others at
their best so far — and reports, per parameter and candidate, the mean absolute error, the 99.9th percentile and L-inf of the
frame against the reference (`robust` columns: away from the support cut, i.e. without the pixels whose value moves by more than
1e-3 between two neighbouring cutoffs — the cut is a discontinuity that float32 and the reference's shader may decide differently
on single pixels); then the best preset as JSON (what gsx_viewer_set_spec_params should get), the residual, and a diff map
(<out>/diff.npz, <out>/diff.png when Pillow is there).

Reference image formats: .npz with `rgba` (float, H x W x 4: premultiplied rgb over the background + alpha) or `frame` (float,
premultiplied rgb + transmittance T: libgsx's own framebuffer) or `rgba8` (uint8); .png (RGB / RGBA 8 bit; needs Pillow).
The product path only: this tool never touches oracle/ (tests/test_gpu_reconcile.py feeds it a frame the float64 spec rendered
with perturbed constants and checks that it finds them)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd.ply import Gaussians  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

#: candidates per parameter (the INRIA / reference-paper conventions and their neighbours; extend with --grid name=v1,v2,...)
GRID = {
    "max_std_dev": [2.0, 2.5, 2.8284271, 3.0, 3.3333333, 3.5, 4.0],
    "low_pass": [0.0, 0.1, 0.2, 0.3, 0.4, 0.5],
    "alpha_max": [0.99, 0.999, 1.0],
    "alpha_min": [0.0, 1.0 / 512.0, 1.0 / 255.0, 1.0 / 128.0],
    "cull_margin": [1.0, 1.1, 1.2, 1.3, 1.5],
    "jacobian_clamp": [1.0, 1.15, 1.3, 1.5, 2.0],
}
ORDER = ["max_std_dev", "low_pass", "alpha_max", "alpha_min", "cull_margin", "jacobian_clamp"]


def load_matrix(arg):
    if os.path.exists(arg):
        return np.load(arg).astype(np.float32).reshape(16)
    return np.array([float(x) for x in arg.replace(",", " ").split()], np.float32).reshape(16)


def load_reference(path, background):
    """-> (float32 H x W x 3 colour over the background, float32 H x W alpha or None)"""
    if path.endswith(".npz"):
        z = np.load(path)
        if "frame" in z:
            f = z["frame"].astype(np.float32)
            return f[..., :3] + f[..., 3:4] * np.asarray(background, np.float32), 1.0 - f[..., 3]
        if "rgba" in z:
            f = z["rgba"].astype(np.float32)
            return f[..., :3], f[..., 3]
        if "rgba8" in z:
            f = z["rgba8"].astype(np.float32) / 255.0
            return f[..., :3], f[..., 3] if f.shape[-1] == 4 else None
        raise SystemExit(f"{path}: expected an array named frame, rgba or rgba8")
    try:
        from PIL import Image
    except ImportError:
        raise SystemExit("reading a PNG needs Pillow; save the reference as .npz (rgba8 = uint8 H x W x 4) instead")
    im = np.asarray(Image.open(path).convert("RGBA"), np.float32) / 255.0
    return im[..., :3], im[..., 3]


class Renderer:
    def __init__(self, gaussians, view, proj, size, sh_degree, splat_size):
        self.v = MultiModelViewer()
        self.v.add_model("m", gaussians.shape[0])
        self.v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, gaussians)
        self.v.set_render_options(speculative=0)   # every frame here has other constants: nothing to speculate from
        self.view, self.proj, self.size, self.sh, self.splat = view, proj, size, sh_degree, splat_size

    def frame(self, params, background):
        v = self.v
        v.set_spec_params(**params)
        v.update_camera_with_matrices(self.view, self.proj, self.size)
        v.update_gaussian_transform(self.splat, GaussianDisplayMode.Splat, GaussianShDegree.new(self.sh), False)
        v.render_frame(["m"])
        f = v.download_framebuffer()
        return f[..., :3] + f[..., 3:4] * np.asarray(background, np.float32), 1.0 - f[..., 3]

    def close(self):
        self.v.close()


def errors(rgb, ref_rgb, keep):
    d = np.abs(rgb - ref_rgb).max(axis=2)
    dk = d[keep] if keep is not None and keep.any() else d.ravel()
    return dict(mean=float(d.mean()), p999=float(np.percentile(d, 99.9)), linf=float(d.max()),
                robust_mean=float(dk.mean()), robust_linf=float(dk.max()))


def reconcile(gaussians, view, proj, size, ref_rgb, background=(0.0, 0.0, 0.0), sh_degree=3, splat_size=1.0, grid=None, sweeps=2,
              quantised=False, log=None):
    """-> dict(best=params, residual=errors, table={param: [(value, errors)]}, diff=H x W float32).  quantised: the reference was an
    8-bit image (errors below 1 / 255 are its rounding; the score adds nothing for them)."""
    grid = dict(GRID, **(grid or {}))
    r = Renderer(gaussians, view, proj, size, sh_degree, splat_size)
    best = {k: None for k in ORDER}
    # start from libgsx's defaults
    sp = r.v.set_spec_params()
    cur = {k: float(getattr(sp, k)) for k in ORDER}
    floor = 0.5 / 255.0 if quantised else 0.0

    def score(e):  # mean error decides; the robust L-inf breaks ties between candidates the mean cannot tell apart
        return max(e["robust_mean"], floor) + 1e-3 * e["robust_linf"]

    # pixels near the support cut: where the frame moves by more than 1e-3 between two neighbouring cutoffs at the starting point
    a, _ = r.frame(dict(cur, max_std_dev=cur["max_std_dev"] * 0.98), background)
    b, _ = r.frame(dict(cur, max_std_dev=cur["max_std_dev"] * 1.02), background)
    keep = np.abs(a - b).max(axis=2) <= 1e-3
    table = {}
    for sweep in range(sweeps):
        for name in ORDER:
            rows = []
            for val in grid[name]:
                rgb, _ = r.frame(dict(cur, **{name: val}), background)
                rows.append((float(val), errors(rgb, ref_rgb, keep)))
            rows.sort(key=lambda t: score(t[1]))
            cur[name] = rows[0][0]
            table[name] = sorted(rows, key=lambda t: t[0])
            if log:
                log(f"sweep {sweep + 1} {name}: best {rows[0][0]:.6g} (robust mean {rows[0][1]['robust_mean']:.3e}, L-inf {rows[0][1]['robust_linf']:.3e}); "
                    + ", ".join(f"{v:.4g}: {e['robust_mean']:.2e}" for v, e in table[name]))
    rgb, alpha = r.frame(cur, background)
    res = errors(rgb, ref_rgb, keep)
    diff = np.abs(rgb - ref_rgb).max(axis=2).astype(np.float32)
    r.close()
    best.update(cur)
    return dict(best=best, residual=res, table={k: [(v, e) for v, e in rows] for k, rows in table.items()}, diff=diff,
                pixels_near_the_support_cut=float(1.0 - keep.mean()))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--ply", required=True)
    ap.add_argument("--view", required=True, help="16 floats, column-major (or a .npy file)")
    ap.add_argument("--proj", required=True)
    ap.add_argument("--size", required=True, help="WxH")
    ap.add_argument("--image", required=True, help="the reference frame: .png or .npz (frame | rgba | rgba8)")
    ap.add_argument("--background", default="0,0,0")
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--splat-size", type=float, default=1.0)
    ap.add_argument("--grid", action="append", default=[], help="name=v1,v2,... replaces a parameter's candidates")
    ap.add_argument("--out", default="reconcile_out")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.lower().split("x"))
    bg = tuple(float(x) for x in a.background.split(","))
    with open(a.ply, "rb") as f:
        g = Gaussians.read_ply(f.read()).gaussians
    ref_rgb, _ = load_reference(a.image, bg)
    if ref_rgb.shape[:2] != (h, w):
        raise SystemExit(f"{a.image} is {ref_rgb.shape[1]}x{ref_rgb.shape[0]}, --size says {w}x{h}")
    grid = {}
    for gspec in a.grid:
        name, vals = gspec.split("=")
        grid[name] = [float(x) for x in vals.split(",")]
    res = reconcile(g, load_matrix(a.view), load_matrix(a.proj), (w, h), ref_rgb, bg, a.sh_degree, a.splat_size, grid,
                    quantised=not a.image.endswith(".npz") or "rgba8" in np.load(a.image), log=lambda s: print(s, file=sys.stderr))
    os.makedirs(a.out, exist_ok=True)
    np.savez_compressed(os.path.join(a.out, "diff.npz"), diff=res["diff"])
    try:
        from PIL import Image

        d = res["diff"]
        Image.fromarray(np.uint8(np.clip(d / max(d.max(), 1e-9), 0, 1) ** 0.5 * 255)).save(os.path.join(a.out, "diff.png"))
    except ImportError:
        pass
    report = dict(best_preset=res["best"], residual=res["residual"], pixels_near_the_support_cut=res["pixels_near_the_support_cut"],
                  per_parameter={k: [dict(value=v, **e) for v, e in rows] for k, rows in res["table"].items()})
    with open(os.path.join(a.out, "report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(dict(best_preset=res["best"], residual=res["residual"])))


if __name__ == "__main__":
    main()
