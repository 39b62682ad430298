"""GPU: block lists (kernels_bin.hip / k_composite_blocks) against per-tile lists (GSX_BIN=0), frame for frame.

Progressive frames bin by blocks of tiles (at most 256 blocks: one 8-bit sort pass) and the compositor decides per tile;
GSX_BIN=0 keeps the per-tile binning + two-pass tile sort for every frame.  Both must give the same pixels bit for bit:
viewports whose grid is below 256 tiles (block = tile), 1080p (8 x 4 tiles per block), 4K (16 x 8), sizes that are not a
multiple of the tile, depth slabs with saturated tiles, speculated frames with windows and repair rounds, layered models."""
import numpy as np
import pytest

from tests import common
from wgpu_3dgs_viewer_app_amd import camera
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, **opts):
    monkeypatch.setenv("GSX_BIN", "0")
    tiles = MultiModelViewer()
    monkeypatch.delenv("GSX_BIN")
    blocks = MultiModelViewer()
    for v in (tiles, blocks):
        v.set_render_options(**opts)
    return tiles, blocks


def _frame(v, pose, keys, size, mode=GaussianDisplayMode.Splat):
    v.update_camera(camera.orbit_pose(pose), size)
    v.update_gaussian_transform(1.0, mode, GaussianShDegree.new(3), False)
    v.render_frame(keys)
    return v.download_framebuffer()


@pytest.mark.parametrize("size", [(256, 176), (200, 136), (1920, 1080), (3840, 2160), (1000, 40)])
@pytest.mark.parametrize("speculative", [0, 1])
def test_block_lists_equal_tile_lists(monkeypatch, size, speculative):
    g = common.small_scene(40000, 401, scale_mul=8.0 if size[0] < 1000 else 30.0)
    tiles, blocks = _pair(monkeypatch, speculative=speculative, min_slab=4096)  # several depth slabs
    for v in (tiles, blocks):
        v.add_model("m", g.shape[0])
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    fewer = 0
    for k, pose in enumerate([3, 4, 5, 6, 120, 121, 122, 122, 40]):
        a, b = _frame(blocks, pose, ["m"], size), _frame(tiles, pose, ["m"], size)
        assert np.array_equal(a, b), f"{size} frame {k}: L-inf {np.abs(a - b).max()}"
        sa, sb = blocks.frame_stats("m"), tiles.frame_stats("m")
        assert sa["n_visible"] == sb["n_visible"] and sa["n_sorted"] == sb["n_sorted"] and sa["speculated"] == sb["speculated"]
        assert sa["n_repair_tiles"] == sb["n_repair_tiles"]
        assert sa["n_tile_entries"] <= sb["n_tile_entries"]
        fewer += sa["n_tile_entries"] < sb["n_tile_entries"]
    tiles_n = ((size[0] + 15) // 16) * ((size[1] + 15) // 16)
    if tiles_n > 256:
        assert fewer >= 6, "blocks of several tiles must need fewer entries than tiles"
    tiles.close()
    blocks.close()


def test_block_lists_layered_models_and_display_modes(monkeypatch):
    ga = common.small_scene(20000, 402, scale_mul=10.0)
    gb = common.small_scene(15000, 403, scale_mul=10.0)
    tiles, blocks = _pair(monkeypatch, min_slab=2048)
    for v in (tiles, blocks):
        for key, g in (("far", ga), ("near", gb)):
            v.add_model(key, g.shape[0])
            v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, g)
    for k, (pose, mode) in enumerate([(10, GaussianDisplayMode.Splat), (11, GaussianDisplayMode.Splat), (12, GaussianDisplayMode.Ellipse),
                                      (13, GaussianDisplayMode.Point), (14, GaussianDisplayMode.Splat), (15, GaussianDisplayMode.Splat)]):
        a = _frame(blocks, pose, ["far", "near"], (640, 360), mode)
        b = _frame(tiles, pose, ["far", "near"], (640, 360), mode)
        assert np.array_equal(a, b), f"frame {k}"
    tiles.close()
    blocks.close()


@pytest.mark.parametrize("speculative", [0, 1])
def test_dispatch_order_is_a_schedule_not_data(monkeypatch, speculative):
    """The block compositor takes its tiles expensive-first (tile_order_job: last frame's cost per tile, two classes, made by one more
    workgroup of the first slab's k_block_counts); GSX_TILE_ORDER=0 keeps index order.  Same pixels bit for bit — a scene whose cost
    is concentrated in part of the screen so that both classes are populated, a viewport change (the order is rebuilt), layered
    models (tiles that return early under `carry`), frames in flight (every lane has an order of its own)."""
    rng = np.random.default_rng(77)
    g = common.small_scene(60000, 404, scale_mul=20.0)
    g2 = common.small_scene(20000, 405, scale_mul=10.0)
    monkeypatch.setenv("GSX_TILE_ORDER", "0")
    plain = MultiModelViewer()
    monkeypatch.delenv("GSX_TILE_ORDER")
    ordered = MultiModelViewer()
    for v in (plain, ordered):
        v.set_render_options(speculative=speculative, min_slab=8192, frames_in_flight=2)
        v.add_model("m", g.shape[0])
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        v.add_model("near", g2.shape[0])
        v.models["near"].gaussian_buffers.gaussians_buffer.update_range(0, g2)
    poses = [int(p) for p in rng.integers(0, 200, 4)]
    plan = [((1920, 1080), ["m"], p) for p in (7, 8, 9, 10, 11, 12)] + [((1000, 600), ["m"], p) for p in (12, 13, 14)] + \
           [((1920, 1080), ["m", "near"], p) for p in (20, 21, 22, 23)] + [((1920, 1080), ["m"], p) for p in poses]
    for k, (size, keys, pose) in enumerate(plan):
        a, b = _frame(ordered, pose, keys, size), _frame(plain, pose, keys, size)
        assert np.array_equal(a, b), f"frame {k} ({size}, {keys}, pose {pose}): L-inf {np.abs(a - b).max()}"
    plain.close()
    ordered.close()


@pytest.mark.parametrize("speculative", [0, 1])
def test_records_carried_through_the_block_sort_change_nothing(monkeypatch, speculative):
    """Models whose block lists are long have the lists' {rect, key, index} records carried through the block sort's write-out, and the
    compositor reads its candidates side by side (k_composite_blocks<.., SORTED>); GSX_SORTED_RECORDS=1 / 0 forces / forbids it.
    Same pixels bit for bit: depth slabs, the repair round, layered models, a translucent scene (nothing saturates: the case the path is
    for), and the viewer that decides by itself from the statistics of earlier frames."""
    g = common.small_scene(60000, 406, scale_mul=20.0)
    thin = g.copy()
    thin["color"][:, 3] = 3   # next to transparent: lists stay long, nothing saturates
    g2 = common.small_scene(20000, 407, scale_mul=10.0)
    viewers = []
    for env in ("0", "1", None):
        if env is None:
            monkeypatch.delenv("GSX_SORTED_RECORDS", raising=False)
        else:
            monkeypatch.setenv("GSX_SORTED_RECORDS", env)
        v = MultiModelViewer()
        v.set_render_options(speculative=speculative, min_slab=8192)
        for key, data in (("m", g), ("thin", thin), ("near", g2)):
            v.add_model(key, data.shape[0])
            v.models[key].gaussian_buffers.gaussians_buffer.update_range(0, data)
        viewers.append(v)
    monkeypatch.delenv("GSX_SORTED_RECORDS", raising=False)
    plan = [((1920, 1080), ["m"], p) for p in (7, 8, 9, 10)] + [((1920, 1080), ["thin"], p) for p in (10, 11, 12, 13, 14, 15, 16, 17, 18)] + \
           [((1000, 600), ["m", "near"], p) for p in (20, 21, 22)] + [((1920, 1080), ["thin", "near"], p) for p in (30, 31, 90)]
    for k, (size, keys, pose) in enumerate(plan):
        frames = [_frame(v, pose, keys, size) for v in viewers]
        assert np.array_equal(frames[0], frames[1]), f"frame {k} ({size}, {keys}, pose {pose}): L-inf {np.abs(frames[0] - frames[1]).max()}"
        assert np.array_equal(frames[0], frames[2]), f"frame {k}: the viewer that decides by itself"
    for v in viewers:
        v.close()


def test_block_lists_under_the_library_validator(monkeypatch):
    """GSX_VALIDATE=1 (read when a viewer is created) checks every range and list index the compositor will dereference before each
    launch.  Speculated frames with depth slabs and repair rounds that find nothing to repair: the repair slab's ranges must not keep
    the main round's (regression, round 4: the fused verification left them when no tile needed a repair)."""
    monkeypatch.setenv("GSX_VALIDATE", "1")
    g = common.small_scene(40000, 401, scale_mul=8.0)
    v = MultiModelViewer()
    v.set_render_options(speculative=1, min_slab=4096)
    v.add_model("m", g.shape[0])
    v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    for size in ((256, 176), (1920, 1080)):
        for pose in (3, 4, 5, 6, 120, 121, 122, 122, 40):
            _frame(v, pose, ["m"], size)
    v.close()
