"""GPU: the HIP path against the float64 fixtures of tests/golden/ DIRECTLY.

The other GPU parity tests compare the kernels with oracle/gsx_oracle.c — a float32 restatement that shares the kernels'
operation order (which is what makes integer stages bit-exact, and also what makes it a twin).  The fixtures here come from
oracle/spec_f64.py: float64, world-space formulation, no tiles, no early termination, its own SH / HSV / quantisation code.
Same bars as the CPU suite holds the C oracle to: equal cull set, projection fields within float32 rounding, mask words
and quantised SH planes bit-exact, frame <= 1e-3 per-channel L-inf (expected: ~1.3e-4 = t_epsilon of the front-to-back
early termination, which the float64 frame does not have).  Every fixture is rendered through the default schedule
(progressive slabs + speculation: frames 2 and 3 are speculated) and through the plain single-pass schedule."""
import numpy as np
import pytest

from tests import golden_util
from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp
from wgpu_3dgs_viewer_app_amd.viewer import Cov3dKind, GaussianDisplayMode, GaussianShDegree, MultiModelViewer, ShKind

pytestmark = pytest.mark.gpu


def test_golden_set_covers_the_feature_list():
    names = " ".join(golden_util.IDS)
    for feature in ("sh3_identity", "2models", "sh0_nosh", "pod_norm8_half", "pod_half_single", "mask_box_minus_ellipsoid", "hsv_edit",
                    "hidden_edit", "ellipse_mode", "point_mode", "large_2models_320x240", "inria_params"):
        assert feature in names, f"tests/golden lacks a {feature} fixture (tests/golden/make_golden.py)"


@pytest.mark.parametrize("schedule", ["default", "flat"])
@pytest.mark.parametrize("path", golden_util.GOLDEN, ids=golden_util.IDS)
def test_hip_path_matches_float64_fixture(path, schedule):
    fx = golden_util.Fixture(path)
    keys = [f"m{k}" for k in range(fx.n_models)]
    with MultiModelViewer(sh=ShKind(fx.pod[0]), cov3d=Cov3dKind(fx.pod[1])) as v:
        if schedule == "flat":
            v.set_render_options(progressive=0, speculative=0)
        else:
            v.set_render_options(progressive=1, speculative=1, min_slab=64, first_slab_divisor=4)
        if fx.params:
            v.set_spec_params(**fx.params)
        for k in range(fx.n_models):
            g = fx.gaussians(k)
            n = g.shape[0]
            v.add_model(keys[k], n)
            bufs = v.models[keys[k]].gaussian_buffers
            # the app's loader: batches (scene.rs:358-375); 250 keeps several batches per model
            for s in range(0, n, 250):
                bufs.gaussians_buffer.update_range(s, g[s:s + 250])
            mp, mq, ms = fx.transform(k)
            v.update_model_transform(keys[k], mp, mq, ms)
            fx.check_pod(k, *bufs.gaussians_buffer.download_pod())
            if fx.mask_expr:
                MaskEvaluator(v).evaluate(MaskOp.parse(fx.mask_expr), keys[k], fx.mask_shapes())
                tail = np.uint32((1 << (n & 31)) - 1 if n & 31 else 0xFFFFFFFF)
                got, ref = bufs.mask_buffer.download(), fx.mask_words(k).copy()
                got[-1] &= tail
                ref[-1] &= tail
                assert np.array_equal(got, ref), "mask words differ from the float64 spec"
            if fx.selection_words(k) is not None:
                bufs.selection_buffer.upload(fx.selection_words(k))
        if fx.sel_edit is not None:
            v.update_selection_edit_with_pod(fx.edit_pod())
        if fx.highlight is not None:
            v.update_selection_highlight(fx.highlight)
        v.update_camera_with_matrices(fx.view, fx.proj, (fx.w, fx.h))
        v.update_gaussian_transform(fx.size, GaussianDisplayMode(fx.display_mode), GaussianShDegree.new(fx.sh_deg), bool(fx.no_sh0))
        order = [keys[k] for k in fx.paint_order]
        frames = []
        for rep in range(3):
            v.render_frame(order)
            frames.append(v.download_framebuffer())
            if rep == 0:
                for k in range(fx.n_models):
                    fx.check_projection(k, v.download_projection(keys[k]))
        if schedule == "default" and fx.sel_edit is None and fx.highlight is None:
            assert all(v.frame_stats(key)["speculated"] for key in order), "frames 2 and 3 should have been speculated"
    for fb in frames:
        fx.check_frame(fb, tight=2.5e-4)
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[1], frames[2])


def _load_fixture_models(v, fx, keys):
    for k in range(fx.n_models):
        g = fx.gaussians(k)
        v.add_model(keys[k], g.shape[0])
        v.models[keys[k]].gaussian_buffers.gaussians_buffer.update_range(0, g)
        v.update_model_transform(keys[k], *fx.transform(k))


@pytest.mark.parametrize("lanes", [1, 2])
def test_default_pipeline_at_full_strength_against_float64(lanes, monkeypatch):
    """the float64 spec meets the pipeline as it runs by default — 300 tiles (blocks of several tiles), two
    layered models, depth slabs, speculated frames whose windows come from ANOTHER camera (so the repair round has work), and,
    in a second viewer, pair buffers small enough that slabs are cut and their tails composited pair-free (GSX_TILE_CAP).
    Every frame <= 1e-3 from the float64 frame; all of them bit-identical to each other."""
    path = [p for p in golden_util.GOLDEN if "large_2models_320x240" in p][0]
    fx = golden_util.Fixture(path)
    assert ((fx.w + 15) // 16) * ((fx.h + 15) // 16) > 256 and fx.prior is not None
    keys = [f"m{k}" for k in range(fx.n_models)]
    order = [keys[k] for k in fx.paint_order]

    def run(v):
        v.set_render_options(min_slab=1024, first_slab_divisor=8, frames_in_flight=lanes)
        _load_fixture_models(v, fx, keys)
        v.update_gaussian_transform(fx.size, GaussianDisplayMode(fx.display_mode), GaussianShDegree.new(fx.sh_deg), False)
        for _ in range(lanes):                       # every lane gets windows that belong to the other camera
            v.update_camera_with_matrices(*fx.prior, (fx.w, fx.h))
            v.render_frame(order)
        v.poll()
        frames, repaired, speculated = [], 0, 0
        for rep in range(2 * lanes + 1):
            v.update_camera_with_matrices(fx.view, fx.proj, (fx.w, fx.h))
            v.render_frame(order)
            frames.append(v.download_framebuffer())
            st = [v.frame_stats(key) for key in order]
            repaired += sum(s["n_repair_tiles"] for s in st) if rep < lanes else 0
            speculated += all(s["speculated"] for s in st)
        return frames, repaired, speculated, sum(v.frame_stats(key)["overflow_slabs"] for key in order), [v.frame_stats(key) for key in order]

    with MultiModelViewer() as v:
        frames, repaired, speculated, overflow, stats = run(v)
    assert repaired > 0, "the first frame at the fixture's camera was speculated from another camera's windows: tiles must need the repair round"
    assert speculated == len(frames) and overflow == 0
    assert all(s["n_tile_entries"] > 0 for s in stats)
    for fb in frames:
        fx.check_frame(fb, tight=2.5e-4)
        assert np.array_equal(fb, frames[0])
    monkeypatch.setenv("GSX_TILE_CAP", "3000")       # pair buffers of 3000 entries: slabs are cut on the device, tails composited pair-free
    with MultiModelViewer() as v:
        spilled, _, _, overflow, _ = run(v)
    assert overflow > 0, "GSX_TILE_CAP=3000 must have cut slabs"
    for fb in spilled:
        assert np.array_equal(fb, frames[0]), "a frame whose slabs spilled differs from the ample-capacity frame"
