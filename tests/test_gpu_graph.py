"""GPU: launches submitted as cached, patched HIP graphs (csrc/gsx_launch.h, gsx_graph.cpp) against launches submitted one by one.

The graph only changes how a frame's kernels reach the device: the same kernels with the same arguments in the same order.
So every frame must be bit-identical with ``gsx_debug_set_launch_graphs`` on and off — over an orbit (camera constants and
sort epochs are patched every frame), across schedule changes (speculated / plain / repairing frames are different kernel
sequences: other cached graphs), buffer growth (pointers change under the cached graph), viewport changes (grids change),
layered models, the split protocol (gsx_preprocess / gsx_sort / gsx_render each a scope of its own), pass timing (events cut
the frame into segments) and frames in flight."""
import numpy as np
import pytest

from tests import common
from wgpu_3dgs_viewer_app_amd import camera, viewer as viewer_mod
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _graphs_back_on():
    yield
    viewer_mod.set_launch_graphs(0)


def _load(v, key, g):
    v.add_model(key, g.shape[0])
    v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)


def _frame(v, pose, keys, size, sh=3):
    v.update_camera(camera.orbit_pose(pose), size)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
    v.render_frame(keys)
    return v.download_framebuffer()


POSES = [3, 4, 5, 6, 7, 120, 121, 122, 122, 40, 41, 42, 43, 44, 45, 46]


@pytest.mark.parametrize("opts", [dict(speculative=1), dict(speculative=0), dict(progressive=0, speculative=0),
                                  dict(speculative=1, frames_in_flight=2), dict(speculative=1, host_verify=1)])
def test_graph_frames_equal_direct_frames(opts):
    g = common.small_scene(60000, 411, scale_mul=25.0)
    size = (1280, 720)
    frames = {}
    for graphs in (0, 2):   # 2: record even when the stream is idle (these loops read every frame back)
        viewer_mod.set_launch_graphs(graphs)
        with MultiModelViewer() as v:
            v.set_render_options(min_slab=4096, **opts)
            _load(v, "m", g)
            frames[graphs] = [_frame(v, p, ["m"], size) for p in POSES]
            st = v.launch_stats()
            if graphs:
                assert st["broken"] == 0 and st["graph_launches"] >= len(POSES), st
                assert st["graph_nodes"] > 10 * len(POSES), st
                assert st["graphs_built"] <= 24, f"the cache is not hit: {st}"
            else:
                assert st["graph_launches"] == 0, st
    for k, (a, b) in enumerate(zip(frames[0], frames[2])):
        assert np.array_equal(a, b), f"{opts} frame {k}: L-inf {np.abs(a - b).max()}"


def test_graph_survives_growth_viewport_and_model_changes():
    """Pointers, grids and the kernel sequence change under the cache: pair buffers that grow (GSX_TILE_CAP unset: the host grows
    them after a spill), another viewport, a second layered model, a mask, then back."""
    ga = common.small_scene(50000, 412, scale_mul=30.0)
    gb = common.small_scene(30000, 413, scale_mul=20.0)
    script = [("size", (640, 360)), ("frames", [10, 11, 12]), ("size", (1920, 1080)), ("frames", [13, 14, 15, 16]), ("add", None),
              ("frames2", [17, 18, 19, 20]), ("mask", None), ("frames2", [21, 22, 23]), ("size", (800, 600)), ("frames2", [24, 25, 26]),
              ("frames", [27, 28])]
    out = {}
    for graphs in (0, 2):   # 2: record even when the stream is idle (these loops read every frame back)
        viewer_mod.set_launch_graphs(graphs)
        res = []
        with MultiModelViewer() as v:
            _load(v, "a", ga)
            size = (640, 360)
            for what, arg in script:
                if what == "size":
                    size = arg
                elif what == "frames":
                    res += [_frame(v, p, ["a"], size) for p in arg]
                elif what == "add":
                    _load(v, "b", gb)
                    v.update_model_transform("b", np.array([0.5, 0.2, -0.4], np.float32), np.array([0, 0, 0, 1], np.float32),
                                             np.array([1.1, 1.0, 0.9], np.float32))
                elif what == "frames2":
                    res += [_frame(v, p, ["b", "a"], size) for p in arg]
                elif what == "mask":
                    words = np.full((ga.shape[0] + 31) // 32, 0x5A5AF0F0, np.uint32)
                    v.models["a"].gaussian_buffers.mask_buffer.upload(words)
            assert v.launch_stats()["broken"] == 0
        out[graphs] = res
    for k, (a, b) in enumerate(zip(out[0], out[2])):
        assert np.array_equal(a, b), f"step {k}: L-inf {np.abs(a - b).max()}"


def test_graph_split_protocol_and_pass_timing():
    """The app's own sequence (preprocess + sort, poll, render, poll: scene.rs:856-873, 613-614) — three scopes — and a frame whose
    passes are bracketed with events (every bracket closes a segment)."""
    g = common.small_scene(40000, 414, scale_mul=25.0)
    size = (1024, 576)
    out = {}
    for graphs in (0, 2):   # 2: record even when the stream is idle (these loops read every frame back)
        viewer_mod.set_launch_graphs(graphs)
        res = []
        with MultiModelViewer() as v:
            _load(v, "m", g)
            for k, p in enumerate(POSES[:10]):
                v.update_camera(camera.orbit_pose(p), size)
                v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
                if k == 5:
                    v.set_pass_timing(True)
                v.preprocessor.preprocess("m")
                v.radix_sorter.sort("m")
                v.poll()
                v.renderer.render(["m"])
                v.poll()
                res.append(v.download_framebuffer())
            t = v.get_pass_timing()
            assert t["project"]["ms"] + t["project_geom"]["ms"] > 0.0
        out[graphs] = res
    for k, (a, b) in enumerate(zip(out[0], out[2])):
        assert np.array_equal(a, b), f"frame {k}: L-inf {np.abs(a - b).max()}"


def test_launches_per_frame_is_counted():
    g = common.small_scene(20000, 415, scale_mul=20.0)
    with MultiModelViewer() as v:
        _load(v, "m", g)
        for p in range(8):
            _frame(v, p, ["m"], (640, 360))
        v.poll()
        n0 = viewer_mod.launch_count()
        for p in range(8, 16):
            _frame(v, p, ["m"], (640, 360))
        per_frame = (viewer_mod.launch_count() - n0) / 8.0
        assert 5 <= per_frame <= 80, per_frame
