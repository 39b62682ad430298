"""GPU: frames in flight (gsx_render_options.frames_in_flight, gsx_api.cpp lanes).

gsx_render_frame deals consecutive frames to L lanes — own stream, own per-frame buffers and speculation windows, shared
Gaussian data.  Under test: every frame is bit-identical to the frame of a one-lane viewer; readback calls see the newest
frame whichever lane rendered it; model data changed between frames (mask, upload, remove + create) reaches every lane;
frames that cannot overlap (a query) fall back to the viewer itself; frames with a selection, edits or the highlight DO overlap
(the lanes view the owner's edit buffers, prepared on the owner's stream); un-synchronised loops deliver complete frames."""
import numpy as np
import pytest

from tests import common
from wgpu_3dgs_viewer_app_amd import camera
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu
W, H = 256, 176


def _viewer(lanes, speculative=1, **opts):
    v = MultiModelViewer()
    v.set_render_options(speculative=speculative, min_slab=2048, frames_in_flight=lanes, **opts)
    return v


def _load(v, key, g):
    v.add_model(key, g.shape[0])
    v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)


def _enqueue(v, pose, keys, size=(W, H)):
    v.update_camera(camera.orbit_pose(pose), size)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
    v.render_frame(keys)


def _frame(v, pose, keys, size=(W, H)):
    _enqueue(v, pose, keys, size)
    return v.download_framebuffer()


@pytest.mark.parametrize("lanes", [2, 3, 4])
@pytest.mark.parametrize("speculative", [0, 1])
def test_every_frame_equals_the_one_lane_frame(lanes, speculative):
    g = common.small_scene(30000, 301, scale_mul=10.0)
    ref, v = _viewer(1, speculative), _viewer(lanes, speculative)
    _load(ref, "m", g)
    _load(v, "m", g)
    poses = [10, 11, 12, 13, 14, 15, 130, 131, 132, 133, 60, 61, 61, 61, 200, 201]
    speculated = 0
    for k, pose in enumerate(poses):
        a, b = _frame(v, pose, ["m"]), _frame(ref, pose, ["m"])
        assert np.array_equal(a, b), f"frame {k} (lane {k % lanes}): L-inf {np.abs(a - b).max()}"
        st, sr = v.frame_stats("m"), ref.frame_stats("m")
        assert st["n_visible"] == sr["n_visible"] and st["n_gaussians"] == sr["n_gaussians"]
        speculated += st["speculated"]
    if speculative:
        assert speculated >= len(poses) - lanes, "every lane speculates from its own last frame, from its second frame on"
    else:
        assert speculated == 0
    v.close()
    ref.close()


def test_unsynchronised_loop_and_interleaved_readback():
    """40 frames enqueued without a host wait; the frame read back at the end, and at a few points inside, is that pose's frame."""
    g = common.small_scene(40000, 302, scale_mul=8.0)
    ref, v = _viewer(1), _viewer(2)
    _load(ref, "m", g)
    _load(v, "m", g)
    for k in range(40):
        _enqueue(v, 20 + k, ["m"])
        if k in (0, 7, 8, 22, 39):
            a = v.download_framebuffer()
            b = _frame(ref, 20 + k, ["m"])
            assert np.array_equal(a, b), f"frame {k}"
    v.poll()
    assert v.frame_stats("m")["overflow_slabs"] == 0
    v.close()
    ref.close()


def test_model_data_changes_reach_every_lane():
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    g = common.small_scene(25000, 303, scale_mul=10.0)
    g2 = common.small_scene(25000, 304, scale_mul=10.0)
    ref, v = _viewer(1), _viewer(3)
    for x in (ref, v):
        _load(x, "m", g)
    pose = 30

    def both(step):
        nonlocal pose
        for _ in range(4):  # every lane renders at least once after each change
            pose += 1
            a, b = _frame(v, pose, ["m"]), _frame(ref, pose, ["m"])
            assert np.array_equal(a, b), f"{step}: pose {pose}"

    both("initial")
    shapes = [MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 0.0], np.float32), scale=np.array([2.5, 2.5, 2.5], np.float32))]
    for x in (ref, v):
        MaskEvaluator(x).evaluate(MaskOp.parse("0"), "m", shapes)
    both("mask")
    for x in (ref, v):  # new Gaussians in the first half of the model, streamed like the loader does
        x.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g2[:12000])
    both("upload")
    for x in (ref, v):
        MaskEvaluator(x).evaluate(None, "m")
    both("mask reset")
    for x in (ref, v):  # the key names another model now
        x.remove_model("m")
        _load(x, "m", g2[:9000])
    both("remove + create")
    v.close()
    ref.close()


def test_two_models_layered_and_viewport_change():
    ga = common.small_scene(15000, 305, scale_mul=10.0)
    gb = common.small_scene(12000, 306, scale_mul=10.0)
    ref, v = _viewer(1), _viewer(2)
    for x in (ref, v):
        _load(x, "far", ga)
        _load(x, "near", gb)
    for k, (pose, size) in enumerate([(5, (W, H)), (6, (W, H)), (7, (W, H)), (8, (320, 200)), (9, (320, 200)), (10, (320, 200)), (11, (W, H)), (12, (W, H))]):
        a, b = _frame(v, pose, ["far", "near"], size), _frame(ref, pose, ["far", "near"], size)
        assert a.shape == b.shape and np.array_equal(a, b), f"frame {k}"
    v.close()
    ref.close()


def test_spilled_slabs_with_frames_in_flight(monkeypatch):
    """pair buffers far too small (GSX_TILE_CAP): the spill compositor finishes every lane's frame on the device."""
    g = common.small_scene(20000, 307, scale_mul=14.0)
    ref = _viewer(1)
    _load(ref, "m", g)
    monkeypatch.setenv("GSX_TILE_CAP", "4096")
    v = _viewer(2)
    monkeypatch.delenv("GSX_TILE_CAP")
    _load(v, "m", g)
    for k in range(8):
        _enqueue(v, 40 + k, ["m"])
    a = v.download_framebuffer()
    b = _frame(ref, 47, ["m"])
    assert np.array_equal(a, b)  # (a spilled slab is composited pair-free in the same order: the same frame)
    assert v.frame_stats("m")["overflow_slabs"] > 0
    v.close()
    ref.close()


def test_staged_calls_between_overlapped_frames_and_close_without_sync():
    """gsx_preprocess / gsx_sort / gsx_render (the reference's own three-call frame) mixed into a run of gsx_render_frame calls
    with frames in flight: the staged frame runs on the viewer itself after the lanes' frames, reads back as the newest frame,
    and the overlapped frames after it are unaffected.  The viewer is then destroyed with frames still in flight."""
    g = common.small_scene(30000, 308, scale_mul=10.0)
    ref, v = _viewer(1), _viewer(3)
    _load(ref, "m", g)
    _load(v, "m", g)
    for k in range(12):
        pose = 70 + k
        if k % 4 == 3:  # the three-call frame
            for x in (v, ref):
                x.update_camera(camera.orbit_pose(pose), (W, H))
                x.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
                x.preprocessor.preprocess("m")
                x.radix_sorter.sort("m")
                x.poll()
                x.renderer.render(["m"])
            a, b = v.download_framebuffer(), ref.download_framebuffer()
        else:
            a, b = _frame(v, pose, ["m"]), _frame(ref, pose, ["m"])
        assert np.array_equal(a, b), f"frame {k}"
    for k in range(7):  # seven frames enqueued, none waited for, then the viewer goes away
        _enqueue(v, 90 + k, ["m"])
    v.close()
    ref.close()


def test_viewport_change_with_frames_in_flight_unsynchronised():
    g = common.small_scene(30000, 309, scale_mul=10.0)
    ref, v = _viewer(1), _viewer(2)
    _load(ref, "m", g)
    _load(v, "m", g)
    sizes = [(W, H)] * 5 + [(640, 368)] * 5 + [(200, 120)] * 4 + [(W, H)] * 3
    for k, size in enumerate(sizes):
        _enqueue(v, 110 + k, ["m"], size)   # no host wait anywhere in the run
    a = v.download_framebuffer()
    for k, size in enumerate(sizes):
        b = _frame(ref, 110 + k, ["m"], size)
    assert a.shape == b.shape and np.array_equal(a, b)
    v.close()
    ref.close()



def test_device_side_resolve_sees_the_newest_frame_on_a_lane():
    """gsx_resolve_rgba8_device must resolve the framebuffer of the lane that rendered the newest frame (an odd
    frame count with two lanes leaves it on a lane, not on the viewer itself)."""
    import ctypes as C

    import torch

    from wgpu_3dgs_viewer_app_amd import _lib

    g = common.small_scene(20000, 305, scale_mul=10.0)
    v = _viewer(2)
    _load(v, "m", g)
    bg = (0.25, 0.5, 0.75)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
    for n_frames in (1, 2, 3, 4, 5):
        for k in range(n_frames):
            _enqueue(v, 20 + k, ["m"])
        cbg = (C.c_float * 3)(*bg)
        _lib.check(v._L.gsx_resolve_rgba8_device(v._h, cbg, 0, H, out.data_ptr()))
        v.poll()
        got = out.cpu().numpy().view(np.uint8).reshape(H, W, 4)
        want = v.download_rgba8(bg)
        assert np.array_equal(got, want), f"after {n_frames} frames the device-side resolve shows another frame"
    v.close()


@pytest.mark.parametrize("lanes", [2, 3])
def test_colour_ops_with_frames_in_flight_unsynchronised(lanes, monkeypatch):
    """A selection edit, stored edits and the highlight on frames in flight, the loop never waiting: the selection, the
    selection edit (a host-side setter: nothing orders it but the library) and the mask change in mid-flight.  Every checked
    frame equals the frame of a one-lane viewer that runs k_edit_prepare every frame."""
    from wgpu_3dgs_viewer_app_amd import query
    from wgpu_3dgs_viewer_app_amd.query import GaussianEditFlag as F

    n = 30000
    g = common.small_scene(n, 311, scale_mul=10.0)
    rng = np.random.default_rng(23)
    words = (n + 31) // 32
    monkeypatch.setenv("GSX_NO_EDIT_CACHE", "1")
    ref = _viewer(1)
    monkeypatch.delenv("GSX_NO_EDIT_CACHE")
    v = _viewer(lanes)
    for x in (ref, v):
        _load(x, "m", g)
    steps = []
    for k in range(40):
        change = None
        if k in (3, 17, 29):
            sel = rng.integers(0, 2 ** 32, words, dtype=np.uint64).astype(np.uint32)
            change = ("selection", sel if k != 29 else None)
        elif k in (5, 11, 23, 33):
            change = ("edit", query.GaussianEditPod([F.ENABLED, F.ENABLED | F.OVERRIDE_COLOR, F.ENABLED | F.HIDDEN, 0][(k // 6) % 4],
                                                    tuple(rng.uniform(0, 1, 3)), 0.1, -0.4, 1.6, float(rng.uniform(0.4, 1.3))))
        elif k in (8, 20):
            change = ("highlight", (1.0, 0.2, 0.0, 0.5 if k == 8 else 0.0))
        elif k == 26:
            change = ("mask", rng.integers(0, 2 ** 32, words, dtype=np.uint64).astype(np.uint32))
        steps.append((60 + k, change))

    def run(x, check_at):
        out = {}
        for k, (pose, change) in enumerate(steps):
            if change:
                kind, val = change
                if kind == "selection":
                    x.models["m"].gaussian_buffers.selection_buffer.upload(val)
                elif kind == "edit":
                    x.update_selection_edit_with_pod(val)
                elif kind == "highlight":
                    x.update_selection_highlight(val)
                else:
                    x.models["m"].gaussian_buffers.mask_buffer.upload(val)
            _enqueue(x, pose, ["m"])
            if k in check_at:
                out[k] = x.download_framebuffer().copy()
        return out

    check = {4, 6, 7, 12, 18, 19, 24, 27, 30, 34, 39}
    a, b = run(v, check), run(ref, set(range(len(steps))))
    for k in sorted(check):
        assert np.array_equal(a[k], b[k]), f"{lanes} lanes, frame {k}: L-inf {np.abs(a[k] - b[k]).max()}"
    assert not np.array_equal(b[4], b[2]) and not np.array_equal(b[12], b[10])
    edits_v = v.models["m"].gaussian_buffers.gaussians_edit_buffer.download()
    edits_r = ref.models["m"].gaussian_buffers.gaussians_edit_buffer.download()
    assert edits_v.tobytes() == edits_r.tobytes(), "the stored edits are the same whichever lane rendered"
    v.close()
    ref.close()


def test_a_closed_viewer_gives_every_device_byte_back():
    """gsx_debug_device_bytes counts what every device buffer of the process holds (the bench line's resident_bytes): viewers with one
    and three lanes, speculated and unspeculated frames, a viewport change, a removed and re-created model — after gsx_viewer_destroy
    the count is where it started, and while the viewer lives it covers at least the model's planes."""
    from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod
    g = common.small_scene(20000, 7)
    before = viewer_mod.device_bytes()
    for lanes in (1, 3):
        v = _viewer(lanes)
        _load(v, "m", g)
        for k, pose in enumerate((3, 4, 5, 120, 121, 6)):
            _enqueue(v, pose, ["m"], (W, H) if k < 4 else (320, 208))
        v.poll()
        held = viewer_mod.device_bytes() - before
        assert held >= 236 * g.shape[0], f"{held} bytes for {g.shape[0]} Gaussians: less than the planes alone"
        v.remove_model("m")
        _load(v, "m", g)
        v.set_render_options(speculative=0, frames_in_flight=lanes)
        _enqueue(v, 7, ["m"])
        v.poll()
        v.close()
        assert viewer_mod.device_bytes() == before, f"{lanes} lane(s): {viewer_mod.device_bytes() - before} bytes still held after close"
