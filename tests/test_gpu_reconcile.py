"""GPU: tools/reconcile.py finds the render constants a reference frame was made with.

The reference frame here comes from the float64 spec (oracle/spec_f64.py, test infrastructure) rendered with constants that are
NOT libgsx's defaults — cutoff k = 2.5, low-pass 0.2, alpha_max 0.99, alpha_min 1 / 255 (the INRIA conventions SURVEY 8c expects
the real crate to use, more or less) — from a PLY that went through gsx_ply_write / gsx_ply_read_gaussians.  The tool sees only the
PLY, the camera matrices and the image, renders through libgsx over its grid of gsx_spec_params, and must come back with those
constants and a residual at float32 level.  Once through the Python function, once through the command line with an 8-bit image."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import spec_f64
from tests import common
from wgpu_3dgs_viewer_app_amd import camera
from wgpu_3dgs_viewer_app_amd.ply import Gaussians

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRUTH = dict(max_std_dev=2.5, low_pass=0.2, alpha_max=0.99, alpha_min=1.0 / 255.0)


def _tool():
    spec = importlib.util.spec_from_file_location("reconcile", os.path.join(ROOT, "tools", "reconcile.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _reference(w, h):
    g0 = common.small_scene(2500, 77, scale_mul=7.0)
    ply = Gaussians(g0).write_ply()
    g = Gaussians.read_ply(ply).gaussians          # what the tool will read
    cam = camera.orbit_pose(33)
    view, proj = cam.view(), cam.projection(w / h)
    cov = spec_f64.cov3d_from_gaussians(g["rot"].astype(np.float64), g["scale"].astype(np.float64)).astype(np.float32)
    color = (g["color"][:, 0].astype(np.uint32) | (g["color"][:, 1].astype(np.uint32) << 8)
             | (g["color"][:, 2].astype(np.uint32) << 16) | (g["color"][:, 3].astype(np.uint32) << 24))
    frame = spec_f64.render(view, proj, w, h, [dict(pos=g["pos"].copy(), color=color, sh=g["sh"].reshape(-1, 45).copy(), cov3d=cov)], params=TRUTH)
    return ply, g, view, proj, frame.astype(np.float32)


def test_reconcile_recovers_perturbed_constants():
    w, h = 160, 120
    _, g, view, proj, frame = _reference(w, h)
    tool = _tool()
    bg = (0.1, 0.2, 0.3)
    ref_rgb = frame[..., :3] + frame[..., 3:4] * np.asarray(bg, np.float32)
    res = tool.reconcile(g, view, proj, (w, h), ref_rgb, bg)
    best = res["best"]
    for k, want in TRUTH.items():
        assert abs(best[k] - want) < 1e-6, (k, best[k], want, res["table"][k])
    assert res["residual"]["robust_mean"] < 2e-5 and res["residual"]["robust_linf"] < 1e-3, res["residual"]
    # libgsx's defaults are NOT the truth here: the tool must have had something to find
    dflt = [e for v, e in res["table"]["max_std_dev"] if abs(v - 3.0) < 1e-6][0]
    assert dflt["robust_mean"] > 5 * res["residual"]["robust_mean"]


def test_reconcile_command_line_with_an_8_bit_image(tmp_path):
    w, h = 160, 120
    ply, _, view, proj, frame = _reference(w, h)
    bg = (0.0, 0.0, 0.0)
    rgb = np.clip(frame[..., :3], 0.0, 1.0)
    rgba8 = np.concatenate([np.floor(rgb * 255.0 + 0.5), np.floor(np.clip(1.0 - frame[..., 3:4], 0, 1) * 255.0 + 0.5)], axis=2).astype(np.uint8)
    (tmp_path / "scene.ply").write_bytes(ply)
    np.savez(tmp_path / "ref.npz", rgba8=rgba8)
    np.save(tmp_path / "view.npy", view)
    np.save(tmp_path / "proj.npy", proj)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reconcile.py"), "--ply", str(tmp_path / "scene.ply"), "--view", str(tmp_path / "view.npy"),
                        "--proj", str(tmp_path / "proj.npy"), "--size", f"{w}x{h}", "--image", str(tmp_path / "ref.npz"), "--out", str(tmp_path / "out")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    # an 8-bit image cannot tell alpha_min 0 from 1 / 512, nor alpha_max 0.99 from 0.999; the cutoff and the low-pass it can
    assert abs(out["best_preset"]["max_std_dev"] - 2.5) < 1e-6 and abs(out["best_preset"]["low_pass"] - 0.2) < 1e-6, out
    assert out["residual"]["robust_mean"] < 0.5 / 255.0 + 1e-4
    rep = json.load(open(tmp_path / "out" / "report.json"))
    assert set(rep["per_parameter"]) == set(_tool().ORDER) and os.path.exists(tmp_path / "out" / "diff.npz")
