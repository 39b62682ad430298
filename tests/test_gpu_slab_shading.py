"""GPU: slab shading (gsx_render_options.slab_shading, round 6).

A progressive frame WITHOUT windows — the first frame of a model, a probe of the speculation tuner, speculative = 0 — projects
geometry only (k_project_geom: position + covariance in, depth key + packed tile rectangle out), depth-sorts every visible Gaussian,
and every depth slab then gives conic / colour records to exactly the records some block of tiles still takes (k_block_bin emits the
slab's shading list; k_shade_quads shades it; the frame's colour ops — selection edit, stored edits, highlight — run on that list).
The reference computes SH colour and the 2D conic for every visible Gaussian (K1 + K3's vertex stage, src/tab/scene.rs:856-863,
2306-2313); the claim under test is that shading only what reaches a pixel list changes NOTHING: frames are bit-identical to
slab_shading = 0 (every Gaussian projected in full by k_project) and to the plainest schedule (progressive = 0), whatever the pod,
the SH degree, the display mode, the mask, the edits, the layering, the viewport — and a readback that wants every record
(gsx_model_download_projection) still gets the oracle's values."""
import numpy as np
import pytest

import oracle
from tests import common
from tests.test_gpu_parity import FB_TOL, assert_projection_equal
from wgpu_3dgs_viewer_app_amd import camera, query
from wgpu_3dgs_viewer_app_amd.viewer import Cov3dKind, GaussianDisplayMode, GaussianShDegree, MultiModelViewer, ShKind

pytestmark = pytest.mark.gpu
W, H = 272, 176
N = 24000


def _viewer(sh=ShKind.Single, cov3d=Cov3dKind.Single, **opts):
    v = MultiModelViewer(sh=sh, cov3d=cov3d)
    v.set_render_options(speculative=0, min_slab=1024, first_slab_divisor=8, **opts)   # several depth slabs on a test-sized model
    return v


def _load(v, key, g, mt=None):
    v.add_model(key, g.shape[0])
    v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)
    if mt is not None:
        v.update_model_transform(key, mt.pos, mt.quat(), mt.scale)


def _frame(v, cam, keys, size=(W, H), sh_deg=3, mode=GaussianDisplayMode.Splat, gsize=1.0, no_sh0=False):
    v.update_camera(cam, size)
    v.update_gaussian_transform(gsize, mode, GaussianShDegree.new(sh_deg), no_sh0)
    v.render_frame(keys)
    return v.download_framebuffer().copy()


@pytest.mark.parametrize("sh,cov,sh_deg,mode,gsize,no_sh0", [
    (ShKind.Single, Cov3dKind.Single, 3, GaussianDisplayMode.Splat, 1.0, False),
    (ShKind.Norm8, Cov3dKind.Half, 3, GaussianDisplayMode.Splat, 1.0, False),       # the app's default pod (app.rs:398-417)
    (ShKind.Half, Cov3dKind.Single, 2, GaussianDisplayMode.Splat, 1.5, True),
    (ShKind.Half, Cov3dKind.Half, 1, GaussianDisplayMode.Ellipse, 1.0, False),
    (ShKind.Single, Cov3dKind.Half, 0, GaussianDisplayMode.Point, 1.0, False),
])
def test_slab_shading_changes_no_pixel(sh, cov, sh_deg, mode, gsize, no_sh0):
    g = common.small_scene(N, 411, scale_mul=9.0)
    poses = [15, 16, 140, 141, 60]
    lazy, full, flat = _viewer(sh, cov), _viewer(sh, cov, slab_shading=0), _viewer(sh, cov, progressive=0)
    for v in (lazy, full, flat):
        _load(v, "m", g)
    for pose in poses:
        cam = camera.orbit_pose(pose)
        a, b, c = (_frame(v, cam, ["m"], sh_deg=sh_deg, mode=mode, gsize=gsize, no_sh0=no_sh0) for v in (lazy, full, flat))
        assert np.array_equal(a, b), f"pose {pose}: slab shading changed the frame, L-inf {np.abs(a - b).max()}"
        assert np.array_equal(a, c), f"pose {pose}: the slab-shaded frame differs from the single-pass frame"
        sa, sb = lazy.frame_stats("m"), full.frame_stats("m")
        assert sa["n_visible"] == sb["n_visible"] and sa["n_tile_entries"] == sb["n_tile_entries"] and not sa["speculated"]
    assert a[..., 3].min() < 1e-4, "the scene must saturate some pixels (later slabs must find saturated tiles)"
    # ... and the slab-shaded viewer really did project geometry only (k_project_geom) where the other ran the full projection (k_project)
    for v in (lazy, full):
        v.set_pass_timing(True, ["project", "project_geom"])
        v.get_pass_timing()
        _frame(v, camera.orbit_pose(61), ["m"], sh_deg=sh_deg, mode=mode, gsize=gsize, no_sh0=no_sh0)
    tl, tf = lazy.get_pass_timing(), full.get_pass_timing()
    assert tf["project"]["launches"] == 1 and tf["project_geom"]["launches"] == 0, tf
    if mode == GaussianDisplayMode.Point:
        # two-pixel dots saturate next to nothing: nearly every visible record is taken by some block, the viewer has measured that
        # (Counters::n_shaded_total against n_visible) and gone back to projecting everything in one streaming pass — same pixels
        assert tl["project"]["launches"] == 1 and tl["project_geom"]["launches"] == 0, tl
    else:
        assert tl["project_geom"]["launches"] == 1 and tl["project"]["launches"] == 0, tl
    for v in (lazy, full, flat):
        v.close()


def test_every_record_is_still_there_for_a_readback():
    """gsx_model_download_projection after a slab-shaded frame: the library completes the records nobody shaded (same kernel code as the
    full projection) — keys and rectangles bit-exact against the oracle, conics and colours within float32 rounding; the frame itself
    within the tolerance of the oracle's frame."""
    g = common.small_scene(N, 412, scale_mul=9.0)
    cam = camera.orbit_pose(77)
    mt = common.odd_transform()
    f, pr, idx, nvis, fb_ref = common.oracle_model_frame(g, cam, W, H, mt)
    with _viewer() as v:
        _load(v, "m", g, mt)
        fb = _frame(v, cam, ["m"])
        assert np.abs(fb - fb_ref).max() <= FB_TOL and np.abs(fb - fb_ref).max() <= 2e-4
        assert_projection_equal(v.download_projection("m"), pr)
        assert np.array_equal(v.download_sorted("m"), idx[:nvis])
        assert np.array_equal(_frame(v, cam, ["m"]), fb), "the frame after the readback"


def test_layered_models_mask_and_colour_ops():
    """Two layered models (the far one composited BEHIND the near one: its slabs start from the tiles the near one saturated), a
    `0 - 1` mask on one, then a rect selection with an HSV edit and the selection highlight: the colour ops run on the slabs' shading
    lists.  Equal, frame by frame, to the viewer that projects everything in full."""
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    ga, gb = common.small_scene(N, 413, scale_mul=8.0), common.small_scene(18000, 414, scale_mul=8.0)
    mta = camera.ModelTransform(pos=np.array([0.0, 0.0, 1.5], np.float32))
    mtb = camera.ModelTransform(pos=np.array([0.5, 0.2, -1.0], np.float32), rot=np.array([0, 40, 0], np.float32))
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 1.5], np.float32), scale=np.array([3.0, 2.5, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.2, 0.1, 1.8], np.float32), scale=np.array([1.2, 1.0, 1.4], np.float32))]
    rect = query.QueryPod.rect((60.0, 40.0), (200.0, 130.0), query.QuerySelectionOp.Set)
    edit = query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.45, 1.2, 0.9), 0.1, 0.3, 1.0, 0.8)
    lazy, full = _viewer(), _viewer(slab_shading=0)
    frames = {}
    for name, v in (("lazy", lazy), ("full", full)):
        _load(v, "a", ga, mta)
        _load(v, "b", gb, mtb)
        MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)
        out = []
        for step, pose in enumerate((30, 31, 32, 150)):
            cam = camera.orbit_pose(pose)
            keys = camera.model_render_order(cam.pos, {"a": mta.world_center(), "b": mtb.world_center()})
            if step == 1:      # a rect query rides on the frame (answered by the geometry-only kernel), then selects
                v.update_query(rect)
            out.append(_frame(v, cam, keys))
            if step == 1:
                for k in keys:
                    v.postprocessor.postprocess(k)
                v.update_query(query.QueryPod.none())
                v.update_selection_edit_with_pod(edit)
                v.update_selection_highlight((1.0, 0.2, 0.1, 0.5))
        frames[name] = out
        v.close()
    for k, (a, b) in enumerate(zip(frames["lazy"], frames["full"])):
        assert np.array_equal(a, b), f"frame {k}: L-inf {np.abs(a - b).max()}"
    assert not np.array_equal(frames["lazy"][1], frames["lazy"][2]), "the edit and the highlight must show"


def test_viewport_changes_and_a_model_that_fills_up():
    """The app streams a model in batches while it draws (scene.rs:341-380) and resizes its viewport at will: slab-shaded frames of a
    partially loaded model, at changing sizes, equal the fully projecting viewer's."""
    g = common.small_scene(N, 415, scale_mul=9.0)
    lazy, full = _viewer(), _viewer(slab_shading=0)
    for v in (lazy, full):
        v.add_model("m", N)
    sizes = [(W, H), (200, 120), (W, H), (320, 200)]
    sent = 0
    for k, size in enumerate(sizes):
        upto = N * (k + 1) // len(sizes)
        cam = camera.orbit_pose(40 + k)
        out = []
        for v in (lazy, full):
            v.models["m"].gaussian_buffers.gaussians_buffer.update_range(sent, g[sent:upto])
            out.append(_frame(v, cam, ["m"], size=size))
        sent = upto
        assert np.array_equal(out[0], out[1]), f"step {k} at {size}: L-inf {np.abs(out[0] - out[1]).max()}"
    lazy.close()
    full.close()


def test_eight_records_per_lane_tiles_on_small_slabs():
    """k_block_bin takes eight records per lane on slabs of a million records and more (the full-size oracle tests reach that by
    themselves); GSX_BIN_BIG_SLAB=1 puts every slab of the small scenes above through the same tile loop — slab cuts at a record,
    overflowing slabs, shading lists, layered models: same pixels, same counters."""
    import os
    import subprocess
    import sys

    if os.environ.get("GSX_BIN_BIG_SLAB"):
        pytest.skip("already inside the rerun")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSX_BIN_BIG_SLAB="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_slab_shading.py", "tests/test_gpu_overflow.py",
                        "tests/test_gpu_blocks.py"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-500:]
