"""GPU: the HIP path against the CPU oracle at BASELINE.json's FULL sizes (VERDICT r5 "missing" item 3).

Up to round 5 the largest frame compared with the oracle was cfg1 (50 k Gaussians, 640x480); cfg2 ... cfg5 were checked at
full size through properties and against other schedules of the same kernels only.  The oracle (oracle/gsx_oracle.c, OpenMP)
renders a whole 10 M-Gaussian 1080p frame in seconds on the GPU box's host cores, so here the contract of K1-K3
(src/tab/scene.rs:856-869 preprocess + sort, :2302-2314 render) is checked directly at 8160 / 32 400 tiles:

  * K1: cull set + depth keys + tile rectangles BIT-EXACT against oracle.project; mean / conic / colour within float32 rounding
  * K2: the depth order (ties by Gaussian index) BIT-EXACT against oracle.depth_sort
  * K3: the frame <= 1e-3 per-channel L-infinity (north_star's tolerance) against oracle.rasterize — the reference's algorithm
        shape: back-to-front, splat-major, no tiles, no early termination — from the plainest schedule (speculative = 0,
        progressive = 0) AND from the default schedule's speculated frames (two frames in flight); the observed value is
        printed and bounded by FB_OBSERVED (a regression guard well inside the tolerance)
"""
import numpy as np
import pytest

import oracle
from tests import common
from tests.test_gpu_parity import FB_TOL, FLOAT_TOL
from wgpu_3dgs_viewer_app_amd import camera, scene
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu

# what the frames are expected to differ by: the compositor stops a pixel at T < t_epsilon = 1e-4 (the oracle blends everything) and
# sums front to back where the oracle sums back to front
FB_OBSERVED = 4e-4


def _oracle_model(g, cam, w, h, mt=None, mask=None, fb=None):
    f = common.oracle_frame(cam, w, h, mt)
    pos, color, sh, cov = oracle.convert(g)
    pr = oracle.project(f, pos, color, sh, cov, mask)
    del sh
    idx, nvis = oracle.depth_sort(pr["key"])
    if fb is None:
        fb = oracle.new_framebuffer(f)
    oracle.rasterize(f, pr, idx, nvis, fb)
    return pr, idx, nvis, fb


def _assert_projection(gp, pr, what):
    assert np.array_equal(gp["key"], pr["key"]), f"{what}: depth keys / cull set differ"
    assert np.array_equal(gp["rect"], pr["rect"]), f"{what}: tile rectangles differ"
    vis = pr["key"] != 0xFFFFFFFF
    for name in ("mean2d", "conic_opacity", "rgb"):
        a, b = gp[name][vis], pr[name][vis]
        bad = np.abs(a - b) > FLOAT_TOL + FLOAT_TOL * np.abs(b)
        assert not bad.any(), f"{what}: {name} differs on {int(bad.sum())} values, max {float(np.abs(a - b).max())}"


def _check_config(cfg, poses_speculated, pose):
    n, sh, w, h, seed = scene.CONFIGS[cfg]
    g = scene.synthetic_gaussians(n, seed, sh)
    cam = camera.orbit_pose(pose)
    pr, idx, nvis, fb_ref = _oracle_model(g, cam, w, h)
    assert nvis > n // 2
    out = {}
    with MultiModelViewer() as v:
        v.set_render_options(speculative=0, progressive=0)
        v.add_model("m", n)
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        del g
        v.update_camera(cam, (w, h))
        v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
        v.preprocessor.preprocess("m")
        v.radix_sorter.sort("m")
        v.poll()
        gp = v.download_projection("m")
        _assert_projection(gp, pr, cfg)
        del gp
        order = v.download_sorted("m")
        assert order.size == nvis and np.array_equal(order, idx[:nvis]), f"{cfg}: depth order differs from the oracle's"
        v.renderer.render(["m"])
        st = v.frame_stats("m")
        assert st["n_visible"] == nvis and st["overflow_slabs"] == 0
        fb_plain = v.download_framebuffer().copy()
        out["plain"] = float(np.abs(fb_plain - fb_ref).max())
        # the default schedule: progressive slabs + temporal occlusion speculation, two frames in flight, arriving at `pose` along
        # the orbit so that its frame is a speculated one (geometry-only projection, admission, sparse shading, one slab, repair)
        v.set_render_options(frames_in_flight=2)
        for p in poses_speculated:
            v.update_camera(camera.orbit_pose(p), (w, h))
            v.render_frame(["m"])
        assert poses_speculated[-1] == pose
        fb_spec = v.download_framebuffer()
        st = v.frame_stats("m")
        assert st["speculated"] and st["n_sorted"] < st["n_visible"], f"{cfg}: the last frame was not speculated: {st}"
        out["speculated"] = float(np.abs(fb_spec - fb_ref).max())
        out["speculated_equals_plain"] = bool(np.array_equal(fb_spec, fb_plain))
        out["n_visible"], out["n_sorted_speculated"] = nvis, st["n_sorted"]
    print(f"{cfg} pose {pose}: HIP vs oracle L-inf plain {out['plain']:.3e}, speculated {out['speculated']:.3e}; "
          f"N_vis {nvis}, depth-sorted on the speculated frame {out['n_sorted_speculated']}")
    assert out["speculated_equals_plain"], f"{cfg}: the speculated frame differs from the plain one"
    for k in ("plain", "speculated"):
        assert out[k] <= FB_TOL, f"{cfg} {k} frame: L-inf {out[k]} > 1e-3 against the oracle"
        assert out[k] <= FB_OBSERVED, f"{cfg} {k} frame: L-inf {out[k]} (expected ~t_epsilon)"


def test_cfg2_full_size_against_the_oracle():
    """BASELINE configs[1]: 1 M Gaussians SH-3 at 1920x1080."""
    _check_config("cfg2", [236, 237, 238, 239, 0], 0)


def test_cfg3_full_size_against_the_oracle():
    """BASELINE configs[2] (garden-sized synthetic, 5.8 M): one pose in the expensive quarter of the orbit."""
    _check_config("cfg3", [96, 97, 98, 99, 100], 100)


def test_cfg4_full_size_against_the_oracle():
    """BASELINE configs[3] on one GPU: 10 M Gaussians SH-3 at 1920x1080 — the frame bench.py's cpu_baseline renders."""
    _check_config("cfg4", [236, 237, 238, 239, 0], 0)


def test_cfg5_two_layered_models_with_mask_at_3840x2160_against_the_oracle():
    """BASELINE configs[4] at its resolution: the masked model ('0 - 1': box minus ellipsoid, evaluated on the device and by the
    oracle: mask words bit-exact) with its TRS, layered with a second 6 M-Gaussian model the way the app paints them (far -> near,
    never merged: scene.rs:533-558, 2302-2314); 32 400 tiles, the oracle paints the far model first and the near one over it."""
    from wgpu_3dgs_viewer_app_amd import parallel
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind, pack_program

    n_total, sh, w, h, seed = scene.CONFIGS["cfg5"]
    n = n_total // 4
    tr = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
          "d": camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                     scale=np.array([1.2, 0.9, 1.1], np.float32))}
    seeds = {"a": seed, "d": seed + 3}
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
    op = MaskOp.parse("0 - 1")
    pose = 40
    cam = camera.orbit_pose(pose)
    keys = parallel.model_render_keys(cam.pos, tr)   # far -> near
    assert sorted(keys) == ["a", "d"]
    fb_ref = None
    ref = {}
    with MultiModelViewer() as v:
        v.set_render_options(speculative=0, progressive=0)
        for k in keys:   # the oracle paints in this order; the upload order does not matter
            g = scene.synthetic_gaussians(n, seeds[k], sh)
            mask = None
            if k == "a":
                mask = oracle.mask_evaluate(oracle.convert(g)[0], tr[k].pos, tr[k].quat(), tr[k].scale, *pack_program(op, shapes))
            pr, idx, nvis, fb_ref = _oracle_model(g, cam, w, h, tr[k], mask, fb_ref)
            ref[k] = (pr, idx, nvis, mask)
            v.add_model(k, n)
            v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, g)
            v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale)
            del g
        MaskEvaluator(v).evaluate(op, "a", shapes)
        got = v.models["a"].gaussian_buffers.mask_buffer.download()
        want = ref["a"][3]
        tail = (1 << (n & 31)) - 1 if n & 31 else 0xFFFFFFFF
        assert np.array_equal(got[:-1], want[:-1]) and (got[-1] & tail) == (want[-1] & tail), "mask words differ"
        kept = int(np.unpackbits(want.view(np.uint8)).sum())
        assert n // 50 < kept < n, kept
        v.update_camera(cam, (w, h))
        v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
        for k in keys:
            v.preprocessor.preprocess(k)
            v.radix_sorter.sort(k)
        v.poll()
        for k in keys:
            pr, idx, nvis, _ = ref[k]
            _assert_projection(v.download_projection(k), pr, f"cfg5 model {k}")
            assert np.array_equal(v.download_sorted(k), idx[:nvis]), f"cfg5 model {k}: depth order differs"
        v.renderer.render(keys)
        fb_plain = v.download_framebuffer().copy()
        # default schedule, speculated frames
        v.set_render_options()
        for p in (36, 37, 38, 39, 40):
            c = camera.orbit_pose(p)
            v.update_camera(c, (w, h))
            v.render_frame(parallel.model_render_keys(c.pos, tr))
        fb_spec = v.download_framebuffer()
        assert any(v.frame_stats(k)["speculated"] for k in keys)
    e_plain, e_spec = float(np.abs(fb_plain - fb_ref).max()), float(np.abs(fb_spec - fb_ref).max())
    print(f"cfg5 (2 of 4 models, mask on one, 3840x2160) pose {pose}: HIP vs oracle L-inf plain {e_plain:.3e}, speculated {e_spec:.3e}; "
          f"N_vis {[ref[k][2] for k in keys]}, mask keeps {kept} of {n}")
    assert np.array_equal(fb_spec, fb_plain)
    assert e_plain <= FB_TOL and e_plain <= FB_OBSERVED, e_plain
