"""GPU: temporal occlusion speculation (gsx_render_options.speculative, kernels_spec.hip / kernels_admit.hip).

The claim under test: whatever the windows inherited from the previous frame — a smooth orbit, a camera jump, a model
that changed underneath (mask, edit, transform), several layered models — the speculated frame is BIT-IDENTICAL to the
frame rendered with speculative = 0, and both stay within the framebuffer tolerance of the oracle.  The statistics
show the speculation engaged (fewer records sorted) and that the repair round ran when it had to."""
import os

import numpy as np
import pytest

import oracle
from tests import common
from tests.test_gpu_parity import FB_TOL
from wgpu_3dgs_viewer_app_amd import camera, query
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu
W, H = 256, 176


def _viewer(speculative, **opts):
    v = MultiModelViewer()
    v.set_render_options(speculative=1 if speculative else 0, min_slab=2048, **opts)
    return v


def _load(v, key, g, mt=None):
    v.add_model(key, g.shape[0])
    v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)
    if mt is not None:
        v.update_model_transform(key, mt.pos, mt.quat(), mt.scale)


def _frame(v, cam, keys, size=(W, H)):
    v.update_camera(cam, size)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
    v.render_frame(keys)
    v.poll()
    return v.download_framebuffer()


@pytest.mark.parametrize("margin,radius,host_verify", [(0.5, 3, 1), (0.0, 0, 1), (2.0, 1, 1), (0.5, 3, 0), (0.0, 0, 0), (0.5, 3, 2), (0.0, 0, 2)])
def test_orbit_with_jumps_bit_identical(margin, radius, host_verify):
    """host_verify 1: the repair round is enqueued only when the device says a tile needs it; 0: always, no host wait;
    2 (the default): 1 while repairs are rare, 0 while they are not."""
    g = common.small_scene(30000, 201, scale_mul=10.0)  # opaque enough that most tiles saturate
    poses = [10, 11, 12, 13, 14, 130, 131, 132, 60, 61, 61, 61, 200]  # smooth runs, jumps, a still camera
    spec, plain = _viewer(True, spec_margin=margin, spec_radius=radius, host_verify=host_verify), _viewer(False)
    _load(spec, "m", g)
    _load(plain, "m", g)
    engaged = repaired = 0
    for k, pose in enumerate(poses):
        cam = camera.orbit_pose(pose)
        a, b = _frame(spec, cam, ["m"]), _frame(plain, cam, ["m"])
        assert np.array_equal(a, b), f"pose {pose} (frame {k}): speculated frame differs, L-inf {np.abs(a - b).max()}"
        st, sp = spec.frame_stats("m"), plain.frame_stats("m")
        assert st["n_visible"] == sp["n_visible"] and not sp["speculated"] and sp["n_sorted"] == sp["n_visible"]
        assert st["speculated"] == (k > 0)
        if st["speculated"]:
            engaged += st["n_sorted"] < 0.8 * st["n_visible"]
            repaired += st["n_repair_tiles"] > 0
            assert (st["n_repair_sorted"] > 0) == (st["n_repair_tiles"] > 0) or st["n_repair_tiles"] > 0
    if margin <= 0.5:
        assert engaged >= 6, "the speculation never reduced the sorted set: the test scene is not opaque enough"
    if margin == 0.0:
        assert repaired >= 3, "zero margin / radius must mispredict on a moving camera, or the repair round is untested"
    # and the oracle agrees with the last frame
    _, _, _, _, fb_ref = common.oracle_model_frame(g, camera.orbit_pose(poses[-1]), W, H)
    assert np.abs(a - fb_ref).max() <= FB_TOL
    spec.close()
    plain.close()


@pytest.mark.parametrize("what", ["edit", "highlight", "both", "stored"])
def test_lazy_shading_with_colour_ops(what):
    """A selection edit / highlight / stored edits no longer switch lazy shading off: a speculated frame shades only what it
    admits and applies the colour ops to exactly those records (shade_admitted) — in the main round, in the repair round (camera
    jumps) and after a full re-projection (download_projection).  Frame for frame equal to the unspeculated viewer's."""
    from wgpu_3dgs_viewer_app_amd.query import GaussianEditFlag as F

    n = 30000
    g = common.small_scene(n, 207, scale_mul=10.0)
    rng = np.random.default_rng(17)
    sel = rng.integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint64).astype(np.uint32)
    sel[-1] &= np.uint32((1 << (n % 32)) - 1) if n % 32 else np.uint32(0xFFFFFFFF)
    edit = query.GaussianEditPod(F.ENABLED, (0.3, 1.5, 0.8), 0.25, -0.75, 2.2, 0.6)   # colour AND opacity change
    spec, plain = _viewer(True, spec_margin=0.1, spec_radius=1), _viewer(False, progressive=0)
    for v in (spec, plain):
        _load(v, "m", g)
        v.models["m"].gaussian_buffers.selection_buffer.upload(sel)
        if what in ("edit", "both", "stored"):
            v.update_selection_edit_with_pod(edit)
        if what in ("highlight", "both"):
            v.update_selection_highlight((1.0, 0.0, 1.0, 0.5))
    engaged = repaired = 0
    for k, pose in enumerate([10, 11, 12, 13, 130, 131, 132, 60, 61, 61, 200, 201]):
        cam = camera.orbit_pose(pose)
        if what == "stored" and k == 3:   # leave edit mode: the edits stored so far keep rendering, no selection
            for v in (spec, plain):
                v.models["m"].gaussian_buffers.selection_buffer.upload(None)
                v.update_selection_edit_with_pod(query.GaussianEditPod.default())
        a, b = _frame(spec, cam, ["m"]), _frame(plain, cam, ["m"])
        assert np.array_equal(a, b), f"{what}: pose {pose} (frame {k}): L-inf {np.abs(a - b).max()}"
        st = spec.frame_stats("m")
        if st["speculated"]:
            engaged += st["n_sorted"] < 0.8 * st["n_visible"]
            repaired += st["n_repair_tiles"] > 0
        if k in (5, 9):   # every record, edited: the lazily shaded frame completes itself
            pa, pb = spec.download_projection("m"), plain.download_projection("m")
            for name in ("key", "rect", "rgb", "conic_opacity"):
                assert np.array_equal(pa[name], pb[name]), (what, k, name)
    assert engaged >= 6 and repaired >= 2, (engaged, repaired)
    unedited = common.small_scene(n, 207, scale_mul=10.0)
    with MultiModelViewer() as u:
        _load(u, "m", unedited)
        assert not np.array_equal(_frame(u, cam, ["m"]), a), "the colour ops must be visible in the frame"
    spec.close()
    plain.close()


def test_scene_changes_under_the_windows():
    """mask, hidden edit, model transform, Gaussian size, viewport: every change invalidates the inherited windows in
    some tiles; the repair round must make up for all of it."""
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    g = common.small_scene(25000, 202, scale_mul=10.0)
    spec, plain = _viewer(True), _viewer(False)
    for v in (spec, plain):
        _load(v, "m", g)
    cam = camera.orbit_pose(40)
    rng = np.random.default_rng(3)
    sel = rng.integers(0, 2 ** 32, (g.shape[0] + 31) // 32, dtype=np.uint64).astype(np.uint32)
    shapes = [MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 0.0], np.float32), scale=np.array([2.5, 2.5, 2.5], np.float32))]

    def both(fn, size=(W, H)):
        out = []
        for v in (spec, plain):
            fn(v)
            out.append(_frame(v, cam, ["m"], size))
        assert np.array_equal(out[0], out[1]), f"L-inf {np.abs(out[0] - out[1]).max()}"
        return spec.frame_stats("m")

    both(lambda v: None)
    both(lambda v: None)
    st = both(lambda v: MaskEvaluator(v).evaluate(MaskOp.parse("0"), "m", shapes))   # everything in front of the core disappears
    assert st["speculated"] and st["n_repair_tiles"] > 0
    both(lambda v: MaskEvaluator(v).evaluate(None, "m"))                               # and comes back
    both(lambda v: (v.models["m"].gaussian_buffers.selection_buffer.upload(sel),
                    v.update_selection_edit_with_pod(query.GaussianEditPod(query.GaussianEditFlag.ENABLED | query.GaussianEditFlag.HIDDEN))))
    both(lambda v: v.update_selection_edit_with_pod(query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.2, 1.0, 1.0), 0, 0, 1.0, 0.3)))
    mt = common.odd_transform()
    both(lambda v: v.update_model_transform("m", mt.pos, mt.quat(), mt.scale))
    both(lambda v: None, size=(W - 40, H + 24))   # another tile grid: the windows do not apply, the frame is unspeculated
    assert not spec.frame_stats("m")["speculated"]
    st = both(lambda v: None, size=(W - 40, H + 24))
    assert st["speculated"]
    spec.close()
    plain.close()


def test_layered_models_speculate_per_model():
    scenes = {"a": (common.small_scene(12000, 203, scale_mul=10.0), camera.ModelTransform(pos=np.array([0.0, 0.0, 2.0], np.float32))),
              "b": (common.small_scene(15000, 204, scale_mul=10.0), common.odd_transform()),
              "c": (common.small_scene(9000, 205, scale_mul=10.0), camera.ModelTransform(pos=np.array([-1.5, 0.2, -2.0], np.float32)))}
    spec, plain = _viewer(True), _viewer(False)
    for v in (spec, plain):
        for k, (g, mt) in scenes.items():
            _load(v, k, g, mt)
    from wgpu_3dgs_viewer_app_amd import parallel

    tr = {k: mt for k, (_, mt) in scenes.items()}
    orders = set()
    for i, pose in enumerate([20, 21, 22, 100, 101, 102, 103, 180]):
        cam = camera.orbit_pose(pose)
        keys = parallel.model_render_keys(cam.pos, tr)
        orders.add(tuple(keys))
        a, b = _frame(spec, cam, keys), _frame(plain, cam, keys)
        assert np.array_equal(a, b), f"pose {pose}: layered speculated frame differs, L-inf {np.abs(a - b).max()}"
        if i:
            assert all(spec.frame_stats(k)["speculated"] for k in keys)
    assert len(orders) > 1, "the layer order must change along the path for the test to cover it"
    # a model leaves the frame and comes back with stale windows
    cam = camera.orbit_pose(180)
    for keys in (["a", "c"], ["c", "b", "a"], ["b"]):
        assert np.array_equal(_frame(spec, cam, keys), _frame(plain, cam, keys))
    spec.close()
    plain.close()


def test_split_protocol_and_statistics():
    """preprocess / sort / render called one by one (the app's protocol) speculate exactly like render_frame; the depth
    order exposed after a speculated frame is the repair round's."""
    g = common.small_scene(20000, 206, scale_mul=10.0)
    cam0, cam1 = camera.orbit_pose(70), camera.orbit_pose(71)
    with _viewer(True) as v, _viewer(False) as p:
        _load(v, "m", g)
        _load(p, "m", g)
        for cam in (cam0, cam1, cam1):
            for x in (v, p):
                x.update_camera(cam, (W, H))
                x.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
                x.preprocessor.preprocess("m")
                x.radix_sorter.sort("m")
                x.renderer.render(["m"])
                x.poll()
            assert np.array_equal(v.download_framebuffer(), p.download_framebuffer())
        st = v.frame_stats("m")
        assert st["speculated"] and st["n_sorted"] < st["n_visible"]
        assert v.download_sorted("m").size == st["n_repair_sorted"]
        # the speculated render consumed its depth order and replaced the windows its admission belonged to: rendering
        # the model once more needs a new preprocess + sort (sorting alone is refused too)
        from wgpu_3dgs_viewer_app_amd.viewer import GsxError

        with pytest.raises(GsxError):
            v.renderer.render(["m"])
        v.radix_sorter.sort("m")
        with pytest.raises(GsxError):
            v.renderer.render(["m"])
        v.preprocessor.preprocess("m")
        v.radix_sorter.sort("m")
        v.renderer.render(["m"])
        v.poll()
        assert np.array_equal(v.download_framebuffer(), p.download_framebuffer())
        p.renderer.render(["m"])  # an unspeculated frame may be rendered again as it is
        p.poll()
        assert np.array_equal(v.download_framebuffer(), p.download_framebuffer())
        # the projection download is untouched by the speculation
        gp, pp = v.download_projection("m"), p.download_projection("m")
        assert all(np.array_equal(gp[k], pp[k]) for k in ("key", "rect", "mean2d", "conic_opacity", "rgb"))
        assert np.array_equal(np.sort(p.download_sorted("m")), np.nonzero(pp["key"] != 0xFFFFFFFF)[0])


@pytest.mark.parametrize("sh_kind,cov_kind,sh_deg", [(0, 0, 2), (0, 1, 1), (1, 0, 3), (1, 1, 2), (2, 0, 3), (2, 1, 1), (3, 0, 3), (0, 0, 0)])
def test_lazy_shading_all_pod_kinds(sh_kind, cov_kind, sh_deg):
    """A speculated frame projects geometry only and shades the admitted Gaussians from the SH record copy (k_shade);
    every pod kind and SH degree must give the very pixels of the unspeculated frame, and the completed projection
    records must equal the unlazy ones."""
    from wgpu_3dgs_viewer_app_amd.viewer import Cov3dKind, ShKind

    g = common.small_scene(20000, 207, scale_mul=10.0)
    spec = MultiModelViewer(sh=ShKind(sh_kind), cov3d=Cov3dKind(cov_kind))
    plain = MultiModelViewer(sh=ShKind(sh_kind), cov3d=Cov3dKind(cov_kind))
    spec.set_render_options(speculative=1, min_slab=2048)
    plain.set_render_options(speculative=0, min_slab=2048)
    _load(spec, "m", g)
    _load(plain, "m", g)
    for k, pose in enumerate([90, 91, 92, 150, 151]):
        cam = camera.orbit_pose(pose)
        out = []
        for v in (spec, plain):
            v.update_camera(cam, (W, H))
            v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh_deg), False)
            v.render_frame(["m"])
            v.poll()
            out.append(v.download_framebuffer())
        assert np.array_equal(out[0], out[1]), f"pose {pose}: L-inf {np.abs(out[0] - out[1]).max()}"
        if k:
            st = spec.frame_stats("m")
            assert st["speculated"] and st["n_sorted"] < st["n_visible"]
    gp, pp = spec.download_projection("m"), plain.download_projection("m")  # completes the lazily shaded records
    assert all(np.array_equal(gp[k], pp[k]) for k in ("key", "rect", "mean2d", "conic_opacity", "rgb"))
    spec.close()
    plain.close()


def test_full_size_bit_identity():
    """BASELINE.json's headline size (10 M Gaussians, SH-3, 1920x1080): speculated and unspeculated frames are the same
    bytes along the bench orbit, across a jump, and the speculation does what the bench line says (a few per cent of the
    visible Gaussians enter the depth sort)."""
    from wgpu_3dgs_viewer_app_amd import scene

    n, sh, w, h, seed = scene.CONFIGS["cfg4"]
    g = scene.synthetic_gaussians(n, seed, sh)
    spec, plain, flat = MultiModelViewer(), MultiModelViewer(), MultiModelViewer()
    plain.set_render_options(speculative=0)
    flat.set_render_options(speculative=0, progressive=0)   # one pass over every tile entry: the plainest schedule
    _load(spec, "m", g)
    _load(plain, "m", g)
    _load(flat, "m", g)
    del g
    for k, pose in enumerate([0, 1, 2, 3, 120, 121]):
        cam = camera.orbit_pose(pose)
        a, b = _frame(spec, cam, ["m"], (w, h)), _frame(plain, cam, ["m"], (w, h))
        assert np.array_equal(a, b), f"pose {pose}: L-inf {np.abs(a - b).max()}"
        if pose in (0, 120):  # includes every viewer's very first frame
            c = _frame(flat, cam, ["m"], (w, h))
            assert np.array_equal(a, c), f"pose {pose}: slabbed / speculated frame differs from the one-pass frame, L-inf {np.abs(a - c).max()}"
            assert flat.frame_stats("m")["n_tile_entries"] > 3 * plain.frame_stats("m")["n_tile_entries"]
        st = spec.frame_stats("m")
        if k:
            assert st["speculated"] and st["n_sorted"] < 0.15 * st["n_visible"], st
    assert a[..., 3].min() < 1e-4  # the scene saturates: there is something to speculate on
    spec.close()
    plain.close()
    flat.close()


@pytest.mark.parametrize("variant", ["surfaces", "translucent"])
def test_full_size_bit_identity_on_scenes_the_speculation_does_not_like(variant):
    """10 M Gaussians at 1920x1080 on the two scenes bench.py's `robustness` object times (scene.VARIANTS): Gaussians on a few large
    surfaces with a heavy-tailed scale distribution — the nearest stand-in for a captured scene like BASELINE configs[2]: some splats
    cover hundreds of tiles, an order of magnitude more list entries per visible Gaussian than the benchmark scene — and the
    benchmark scene made translucent, where next to no tile ever saturates.  Speculated, slabbed and one-pass frames are the same bytes."""
    from wgpu_3dgs_viewer_app_amd import scene

    n, sh, w, h, seed = scene.CONFIGS["cfg4"]
    g = scene.synthetic_gaussians(n, seed, sh, variant=variant)
    spec, plain, flat = MultiModelViewer(), MultiModelViewer(), MultiModelViewer()
    plain.set_render_options(speculative=0)
    flat.set_render_options(speculative=0, progressive=0)
    for v in (spec, plain, flat):
        _load(v, "m", g)
    del g
    for k, pose in enumerate([0, 1, 2, 3, 120, 121]):
        cam = camera.orbit_pose(pose)
        a, b = _frame(spec, cam, ["m"], (w, h)), _frame(plain, cam, ["m"], (w, h))
        assert np.array_equal(a, b), f"{variant} pose {pose}: L-inf {np.abs(a - b).max()}"
        if pose in (0, 121):
            c = _frame(flat, cam, ["m"], (w, h))
            assert np.array_equal(a, c), f"{variant} pose {pose}: slabbed / speculated frame differs from the one-pass frame, L-inf {np.abs(a - c).max()}"
    st = flat.frame_stats("m")
    per_visible = st["n_tile_entries"] / max(st["n_visible"], 1)
    if variant == "surfaces":
        # the default scene: ~7 tile entries per visible Gaussian (60 M over 8.5 M); here the heavy tail dominates
        assert per_visible > 30.0, per_visible
        assert a[..., 3].min() < 1e-4
        big = plain.download_projection("m")["rect"]
        tiles = (big[:, 2].astype(np.int64) - big[:, 0]) * (big[:, 3].astype(np.int64) - big[:, 1])   # rect = x0, y0, x1, y1 in tiles
        assert (tiles > 500).sum() > 100, "some splats must cover more than 500 tiles"
    else:
        # (median opacity 0.004: only rays through the dense clusters pile up enough to saturate)
        assert (a[..., 3] < 1e-4).mean() < 0.25, f"the translucent scene saturates {100 * (a[..., 3] < 1e-4).mean():.1f} % of the pixels"
    for v in (spec, plain, flat):
        v.close()


def test_cfg5_full_size_layered_models():
    """BASELINE.json configs[4] at full size: 4 models x 6 M Gaussians (24 M), each with its own TRS, a `0 - 1` mask op on one
    of them, a rect selection with an HSV edit on another, 3840x2160.  No oracle finishes this; the properties checked:
    speculated frames equal unspeculated ones bit for bit, layering order matters, the mask and the edit take effect."""
    from wgpu_3dgs_viewer_app_amd import parallel, scene
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    n_total, sh, w, h, seed = scene.CONFIGS["cfg5"]
    n = n_total // 4
    tr = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
          "b": camera.ModelTransform(pos=np.array([2.0, 0.2, -1.0], np.float32), rot=np.array([0, 35, 0], np.float32)),
          "c": camera.ModelTransform(pos=np.array([-2.5, -0.1, -0.5], np.float32), scale=np.array([0.9, 0.9, 0.9], np.float32)),
          "d": common.odd_transform()}
    spec, plain = MultiModelViewer(), MultiModelViewer()
    plain.set_render_options(speculative=0)
    for i, k in enumerate(tr):
        g = scene.synthetic_gaussians(n, seed + i, sh)
        for v in (spec, plain):
            _load(v, k, g, tr[k])
        del g
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
    rect = query.QueryPod.rect((1200.0, 600.0), (2600.0, 1500.0), query.QuerySelectionOp.Set)
    edit = query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.5, 1.0, 1.2), 0.1, 0.2, 1.0, 0.9)
    frames = []
    for step, pose in enumerate([10, 11, 12, 13]):
        cam = camera.orbit_pose(pose)
        keys = parallel.model_render_keys(cam.pos, tr)
        out = []
        for v in (spec, plain):
            if step == 1:
                MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)
            v.update_query(rect if step == 2 else query.QueryPod.none())
            if step == 3:
                v.update_selection_edit_with_pod(edit)
            out.append(_frame(v, cam, keys, (w, h)))
            for k in keys:
                v.postprocessor.postprocess(k)
            v.poll()
        if not np.array_equal(out[0], out[1]):
            d = np.abs(out[0] - out[1]).max(-1)
            ys, xs = np.nonzero(d)
            raise AssertionError(f"step {step}: L-inf {d.max()}, {ys.size} px, rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}, "
                                 f"spec {[spec.frame_stats(k) for k in keys]} plain {[plain.frame_stats(k) for k in keys]}")
        frames.append(out[0])
        if step:
            assert any(spec.frame_stats(k)["speculated"] for k in keys)
    nsel = sum(int(np.unpackbits(spec.models[k].gaussian_buffers.selection_buffer.download().view(np.uint8)).sum()) for k in tr)
    assert nsel > 1000, "the rectangle must select something"
    assert sum(spec.frame_stats(k)["n_visible"] for k in tr) > 5_000_000
    kept = int(np.unpackbits(spec.models["a"].gaussian_buffers.mask_buffer.download().view(np.uint8)).sum())
    assert 0 < kept < n
    # layering matters at this size too
    cam = camera.orbit_pose(13)
    keys = parallel.model_render_keys(cam.pos, tr)
    assert not np.array_equal(_frame(plain, cam, keys[::-1], (w, h)), frames[-1])
    spec.close()
    plain.close()


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("GSX_FUZZ_SEEDS", "1,2,3,4,5,6,7,8").split(",")])
def test_fuzz_operation_sequences(seed, monkeypatch):
    """Seeded random walks through the API — camera steps and jumps, viewport changes, models shown / hidden / re-ordered,
    masks, selections (uploaded, and made by rectangle queries + postprocess), selection edits, stored edits uploaded and
    dropped, the unedited view, highlight, display mode, Gaussian size, speculation parameters — applied to a speculating and
    a non-speculating viewer in lock-step: every frame must be the same bytes.  The plain viewer also runs k_edit_prepare every
    frame (GSX_NO_EDIT_CACHE) while the speculating one skips it when none of its inputs changed."""
    from wgpu_3dgs_viewer_app_amd import parallel
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    rng = np.random.default_rng(1000 + seed)
    scenes = {k: common.small_scene(int(rng.integers(6000, 16000)), 300 + 10 * seed + i, scale_mul=float(rng.uniform(6.0, 12.0)))
              for i, k in enumerate("abc")}
    tr = {k: camera.ModelTransform(pos=rng.uniform(-2.0, 2.0, 3).astype(np.float32), rot=rng.uniform(-40, 40, 3).astype(np.float32),
                                   scale=rng.uniform(0.7, 1.3, 3).astype(np.float32)) for k in scenes}
    lanes = 1 + seed % 3  # frames in flight: consecutive frames of the speculating viewer alternate between lanes
    spec = _viewer(True, host_verify=seed % 3, frames_in_flight=lanes)
    monkeypatch.setenv("GSX_NO_EDIT_CACHE", "1")
    plain = _viewer(False)
    monkeypatch.delenv("GSX_NO_EDIT_CACHE")
    for v in (spec, plain):
        for k, g in scenes.items():
            _load(v, k, g, tr[k])
    unedited = {k: False for k in scenes}
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.zeros(3, np.float32), scale=np.array([2.0, 2.0, 2.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.5, 0.0, 0.5], np.float32), scale=np.array([1.5, 1.2, 1.5], np.float32))]
    pose, size, visible = int(rng.integers(0, 240)), (W, H), ["a", "b", "c"]
    state = dict(size=1.0, mode=GaussianDisplayMode.Splat, deg=3)
    speculated = 0
    for step in range(60):
        op = int(rng.integers(0, 17))
        qpod = None
        both = lambda fn: [fn(v) for v in (spec, plain)]  # noqa: E731
        if op <= 4:
            pose = (pose + int(rng.integers(1, 3))) % 240                       # a camera step
        elif op == 5:
            pose = int(rng.integers(0, 240))                                    # a jump
        elif op == 6:
            size = [(W, H), (W - 32, H), (W, H + 16), (200, 120), (640, 368), (1000, 200)][int(rng.integers(0, 6))]  # the last two: blocks of several tiles
        elif op == 7:
            visible = [k for k in "abc" if rng.random() < 0.7] or ["b"]
        elif op == 8:
            k, expr = "abc"[int(rng.integers(0, 3))], [None, "0", "!1", "0 - 1", "0 | 1"][int(rng.integers(0, 5))]
            both(lambda v: (v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale),
                            MaskEvaluator(v).evaluate(MaskOp.parse(expr) if expr else None, k, shapes)))
        elif op == 9:
            k = "abc"[int(rng.integers(0, 3))]
            words = rng.integers(0, 2 ** 32, (scenes[k].shape[0] + 31) // 32, dtype=np.uint64).astype(np.uint32)
            if rng.random() >= 0.8:
                words = None  # clear the selection
            both(lambda v: v.models[k].gaussian_buffers.selection_buffer.upload(words))
        elif op == 10:
            flags = [0, 1, 3, 5][int(rng.integers(0, 4))]
            pod = query.GaussianEditPod(flags, tuple(rng.uniform(0, 1, 3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-1, 1)),
                                        float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.3, 1.5)))
            hl = (1.0, 0.0, 1.0, float(rng.choice([0.0, 0.4])))
            both(lambda v: (v.update_selection_edit_with_pod(pod), v.update_selection_highlight(hl)))
        elif op == 11:
            state["mode"] = [GaussianDisplayMode.Splat, GaussianDisplayMode.Ellipse, GaussianDisplayMode.Point][int(rng.integers(0, 3))]
            state["size"] = float(rng.choice([0.6, 1.0, 1.5]))
            state["deg"] = int(rng.integers(0, 4))
        elif op == 12:
            k = "abc"[int(rng.integers(0, 3))]
            tr[k] = camera.ModelTransform(pos=rng.uniform(-2.0, 2.0, 3).astype(np.float32), rot=rng.uniform(-40, 40, 3).astype(np.float32),
                                          scale=rng.uniform(0.7, 1.3, 3).astype(np.float32))
            both(lambda v: v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale))
        elif op == 13:
            spec.set_render_options(speculative=1, min_slab=2048, spec_margin=float(rng.choice([0.0, 0.25, 1.0])), spec_radius=int(rng.integers(0, 5)),
                                    host_verify=int(rng.integers(0, 3)), frames_in_flight=lanes)
        elif op == 14:   # stored edits from the host (a loaded session), or dropped
            k = "abc"[int(rng.integers(0, 3))]
            edits = None
            if rng.random() < 0.7:
                edits = query.default_edits(scenes[k].shape[0])
                on = rng.random(edits.shape[0]) < 0.3
                edits["flag"][on] = rng.choice([1, 3, 5], int(on.sum()))
                edits["color"][on] = rng.uniform(0, 1, (int(on.sum()), 3))
                edits["alpha"][on] = rng.uniform(0.3, 1.5, int(on.sum()))
            both(lambda v: v.models[k].gaussian_buffers.gaussians_edit_buffer.upload(edits))
        elif op == 15:   # a rectangle query this frame: postprocess applies its selection op
            x0, y0 = float(rng.uniform(0, size[0] * 0.6)), float(rng.uniform(0, size[1] * 0.6))
            sop = [query.QuerySelectionOp.Set, query.QuerySelectionOp.Add, query.QuerySelectionOp.Remove][int(rng.integers(0, 3))]
            qpod = query.QueryPod.rect((x0, y0), (x0 + float(rng.uniform(10, size[0] * 0.4)), y0 + float(rng.uniform(10, size[1] * 0.4))), sop)
        else:
            k = "abc"[int(rng.integers(0, 3))]
            unedited[k] = not unedited[k]
            both(lambda v: v.show_unedited(k, unedited[k]))
        cam = camera.orbit_pose(pose)
        keys = [k for k in parallel.model_render_keys(cam.pos, tr) if k in visible]
        out = []
        for v in (spec, plain):
            v.update_camera(cam, size)
            v.update_gaussian_transform(state["size"], state["mode"], GaussianShDegree.new(state["deg"]), False)
            v.update_query(qpod if qpod is not None else query.QueryPod.none())
            v.render_frame(keys)
            for k in keys:
                v.postprocessor.postprocess(k)
            v.poll()
            out.append(v.download_framebuffer())
        if qpod is not None:
            for k in keys:
                assert np.array_equal(spec.models[k].gaussian_buffers.selection_buffer.download(),
                                      plain.models[k].gaussian_buffers.selection_buffer.download()), f"seed {seed} step {step}: selections differ"
        assert np.array_equal(out[0], out[1]), f"seed {seed} step {step} op {op}: L-inf {np.abs(out[0] - out[1]).max()}"
        speculated += any(spec.frame_stats(k)["speculated"] for k in keys)
    assert speculated > (15 if lanes == 1 else 8)
    spec.close()
    plain.close()


def test_speculation_steps_aside_when_it_does_not_pay():
    """A sparse scene whose few saturating tiles keep changing (zero margin and radius, moving camera): most speculated frames
    need the repair round, and the plain progressive path is faster.  The viewer times both paths (a few frames bracketed by
    HIP events, never waited for) and stays on the faster one: after the first probes most frames are unspeculated, it keeps
    re-probing now and then — and the pixels are the same throughout."""
    g = common.small_scene(100000, 208, scale_mul=1.0)
    spec, plain = _viewer(True, spec_margin=0.0, spec_radius=0), _viewer(False)
    _load(spec, "m", g)
    _load(plain, "m", g)
    flags, repairs = [], []
    for k in range(200):
        cam = camera.orbit_pose((3 * k) % 240)
        a, b = _frame(spec, cam, ["m"]), _frame(plain, cam, ["m"])
        assert np.array_equal(a, b), f"frame {k}"
        st = spec.frame_stats("m")
        flags.append(bool(st["speculated"]))
        repairs.append(st["n_repair_tiles"] > 0)
    assert sum(repairs[:24]) >= 7, f"the setup must make the speculation fail, or nothing is tested: {repairs[:24]}"
    assert all(flags[1:32]), "the first phase speculates (that is how it finds out)"
    assert not any(flags[32:37]), "then a probe of plain frames"
    assert flags[37], "and back to speculated frames until the probe's timings are in"
    # whichever path the timing prefers, the viewer must have settled on ONE for long stretches, probing the other briefly
    tail = flags[40:]
    runs = [len(r) for r in "".join("1" if f else "0" for f in tail).replace("01", "0 1").replace("10", "1 0").split()]
    assert max(runs) >= 48, f"no long phase in {runs}"
    spec.close()
    plain.close()


def test_viewport_resizes_with_validation(monkeypatch):
    """Viewport sizes going up and down (the tile tables are allocated with slack and reused): every frame equals the frame
    of a fresh viewer, and the library's own pre-composite validation (GSX_VALIDATE, read when a viewer is created) finds
    no tile range or list index out of bounds.  Regression: a range table zeroed only up to a smaller frame's tile count."""
    monkeypatch.setenv("GSX_VALIDATE", "1")
    g = common.small_scene(20000, 77, scale_mul=8.0)
    sizes = [(224, 192), (256, 208), (200, 120), (256, 208), (256, 192), (224, 192), (320, 240), (200, 120), (320, 240)]
    viewers = [_viewer(True), _viewer(False)]
    for v in viewers:
        _load(v, "m", g)
    for k, size in enumerate(sizes * 2):
        cam = camera.orbit_pose(2 * k)
        fresh = _viewer(False)
        _load(fresh, "m", g)
        ref = _frame(fresh, cam, ["m"], size)
        fresh.close()
        for v in viewers:
            assert np.array_equal(_frame(v, cam, ["m"], size), ref), f"frame {k} at {size}"
    for v in viewers:
        v.close()


@pytest.mark.parametrize("host_verify", [0, 1, 2])
def test_speculation_at_3840x2160_with_small_layered_models(host_verify):
    """32 400 tiles (BASELINE configs[4]'s viewport) without configs[4]'s 24 M Gaussians: the verification, the window pyramids (nine levels),
    the block tables and the repair round of two layered models over poses with jumps — every speculated frame equals the unspeculated one,
    and the repair round runs when it must.  (The full-size cfg5 tests take the same path at 6 M Gaussians a model; this one is cheap enough
    to run for every host_verify policy.)"""
    size = (3840, 2160)
    ga, gb = common.small_scene(40000, 611, scale_mul=30.0), common.small_scene(30000, 612, scale_mul=30.0)
    mta = camera.ModelTransform(pos=np.array([0.0, 0.0, 1.0], np.float32))
    mtb = camera.ModelTransform(pos=np.array([0.4, 0.1, -1.5], np.float32), rot=np.array([0, 30, 0], np.float32))
    spec, plain = _viewer(True, host_verify=host_verify), _viewer(False)
    for v in (spec, plain):
        _load(v, "a", ga, mta)
        _load(v, "b", gb, mtb)
    repaired = engaged = 0
    for k, pose in enumerate([20, 21, 22, 23, 140, 141, 142, 70, 70, 71]):
        cam = camera.orbit_pose(pose)
        keys = camera.model_render_order(cam.pos, {"a": mta.world_center(), "b": mtb.world_center()})
        a, b = _frame(spec, cam, keys, size), _frame(plain, cam, keys, size)
        assert np.array_equal(a, b), f"pose {pose} (frame {k}): L-inf {np.abs(a - b).max()}"
        for key in keys:
            st = spec.frame_stats(key)
            engaged += int(st["speculated"])
            repaired += int(st["n_repair_tiles"] > 0)
    assert engaged >= 12 and repaired >= 1, (engaged, repaired)
    assert a[..., 3].min() < 1e-4, "some pixels must saturate for windows to be bounded"
    spec.close()
    plain.close()
