#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the float64 spec oracle (oracle/spec_f64.py).

PARITY UNPINNED: the reference holds no golden vectors for this path and its implementation (the crate
wgpu-3dgs-viewer 0.2.0) is neither vendored nor buildable here, so these fixtures pin the WRITTEN SPEC
(spec/RENDER_SPEC.md), not the reference.  Inputs are stored next to the expected outputs so the
fixtures stay valid if the scene generator changes.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import spec_f64  # noqa: E402
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pod_f64(g):
    """Pod planes from gs::Gaussian records; cov3d in float64 from rot/scale, rounded to float32 like the upload."""
    cov = spec_f64.cov3d_from_gaussians(g["rot"].astype(np.float64), g["scale"].astype(np.float64)).astype(np.float32)
    color = (g["color"][:, 0].astype(np.uint32) | (g["color"][:, 1].astype(np.uint32) << 8)
             | (g["color"][:, 2].astype(np.uint32) << 16) | (g["color"][:, 3].astype(np.uint32) << 24))
    return g["pos"].copy(), color, g["sh"].reshape(-1, 45).copy(), cov


def make(name, models, cam, w, h, **kw):
    view, proj = cam.view(), cam.projection(w / h)
    out = dict(view=view, proj=proj, size=np.array([w, h]), n_models=np.array(len(models)))
    spec_models = []
    for i, (g, mt) in enumerate(models):
        pos, color, sh, cov = pod_f64(g)
        out[f"g{i}"] = g
        out[f"cov{i}"] = cov
        out[f"mt{i}"] = np.concatenate([mt.pos, mt.quat(), mt.scale]).astype(np.float32)
        spec_models.append(dict(pos=pos, color=color, sh=sh, cov3d=cov, m_pos=mt.pos, m_quat=mt.quat(), m_scale=mt.scale))
        pr = spec_f64.project(view, proj, w, h, pos, color, sh, cov, mt.pos, mt.quat(), mt.scale, **kw)
        out[f"visible{i}"] = pr["visible"]
        out[f"mean2d{i}"] = pr["mean2d"]
        out[f"conic{i}"] = pr["conic"]
        out[f"rgb{i}"] = pr["rgb"]
        out[f"depth{i}"] = pr["depth"]
    # paint order far -> near by centre distance (scene.rs:533-558)
    keys = camera.model_render_order(cam.pos, {i: mt.world_center() for i, (_, mt) in enumerate(models)})
    out["paint_order"] = np.array(keys)
    frame = spec_f64.render(view, proj, w, h, [spec_models[k] for k in keys], **kw)
    out["frame"] = frame.astype(np.float32)
    for k, v in kw.items():
        out[f"kw_{k}"] = np.array(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k in ("frame", "g0")},
          "mean T", float(frame[..., 3].mean()))


def scene_small(n, seed, sh_degree=3, mul=6.0):
    g = scene.synthetic_gaussians(n, seed, sh_degree)
    g["scale"] *= np.float32(mul)
    return g


if __name__ == "__main__":
    ident = camera.ModelTransform()
    odd = camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                scale=np.array([1.2, 0.9, 1.1], np.float32))
    make("frame_sh3_identity_96x64_n600_seed101", [(scene_small(600, 101), ident)], camera.orbit_pose(17), 96, 64)
    make("frame_sh3_trs_2models_112x80_n500_seed102",
         [(scene_small(500, 102), odd), (scene_small(400, 103), camera.ModelTransform(pos=np.array([0, 0, 1.5], np.float32)))],
         camera.orbit_pose(200), 112, 80)
    make("frame_sh0_nosh_size_80x48_n400_seed104", [(scene_small(400, 104, 0), ident)], camera.orbit_pose(90), 80, 48,
         size=1.5, sh_deg=0)
