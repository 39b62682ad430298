"""GPU: gsx_shard_render_frame ITSELF with more than one rank.

`world` ranks run as host threads of this process, each with its own viewer, stream and index shard, and every rank calls the
library's own frame loop — gsx_shard_render_frame (csrc/gsx_comm.cpp: enqueue -> verdict -> whole-shard redo -> exactly sized
repair round -> band gather) — over the in-process group transport of the library (csrc/gsx_comm_group.cpp: the two
collectives as device copies ordered by HIP events, delivered in RCCL's order).  Nothing of the protocol is re-stated in
Python here (tests/test_gpu_sharded.py does that for parallel.ShardedViewer); Python only starts the threads.

Every rank's gathered frame must equal the single-viewer frame BIT FOR BIT whatever the limits, the slot size, the camera
or the number of frames in flight; a rank that leaves the loop makes the others fail with GSX_ERR_RCCL, not hang."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from tests import common
from wgpu_3dgs_viewer_app_amd import _lib, camera, parallel
from wgpu_3dgs_viewer_app_amd.viewer import CommGroup, GaussianDisplayMode, GaussianShDegree, GsxError, MultiModelViewer

pytestmark = pytest.mark.gpu
KEY_ALL = parallel.KEY_ALL
N, W, H = 9000, 208, 152
POSES = (57, 58, 61, 90, 91, 200)   # a coherent stretch, then two camera jumps
TILES = ((H + 15) // 16, (W + 15) // 16)


def _scene():
    return common.small_scene(N, 91, scale_mul=14.0)  # opaque enough that tiles saturate


def _uniforms(v, pose, size=(W, H)):
    v.update_camera(camera.orbit_pose(pose), size)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)


def _single_frames(g, poses=POSES, size=(W, H)):
    out = []
    with MultiModelViewer() as v:
        v.add_model("m", g.shape[0])
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        for pose in poses:
            _uniforms(v, pose, size)
            v.render_frame(["m"])
            out.append(v.download_framebuffer().copy())
    return out


def run_group(world, body, timeout_ms=30000):
    """body(rank, group) on `world` threads over one CommGroup; returns the results, re-raises the first failure that is not
    merely the echo (GSX_ERR_RCCL) of another rank's failure."""
    group = CommGroup(world, timeout_ms)
    results, errors = [None] * world, [None] * world

    def main(r):
        try:
            results[r] = body(r, group)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    group.close()
    real = [e for e in errors if e is not None and not (isinstance(e, GsxError) and e.status == _lib.GSX_ERR_RCCL)]
    echo = [e for e in errors if e is not None]
    if real or echo:
        raise (real or echo)[0]
    return results


def _rank_viewer(g, n_total, rank, world, group, lanes=1, **opts):
    s0, c = parallel.shard_range(n_total, rank, world)
    v = MultiModelViewer()
    v.set_render_options(frames_in_flight=lanes, **opts)
    v.add_model("m", c)
    v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g[s0:s0 + c])
    v.comm_init_group(group, rank)
    return v, (n_total + world - 1) // world


@pytest.mark.parametrize("world,mode", [(2, "natural"), (3, "natural"), (8, "natural"), (2, "off"), (3, "all_refusing"),
                                        (8, "all_refusing"), (5, "all_open"), (4, "stale"), (3, "tiny_slots"), (8, "tiny_slots")])
def test_library_frame_loop_with_peers(world, mode):
    g = _scene()
    ref = _single_frames(g)
    assert ref[0][..., 3].min() < 1e-4, "the scene must saturate some pixels"

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        rng = np.random.default_rng(5)   # the same stream on every rank: limits are a global input
        frames = []
        for pose in POSES:
            _uniforms(v, pose)
            if mode == "all_refusing":   # every tile refuses all but the nearest records: verdict -> exactly sized repair round
                v.shard_set_limits("m", np.full(TILES, 0x40400000, np.uint32))
            elif mode == "all_open":
                v.shard_set_limits("m", np.full(TILES, KEY_ALL, np.uint32))
            elif mode == "stale":        # limits unrelated to the frame: random depths 2..8, a third unbounded
                lim = rng.uniform(2.0, 8.0, TILES).astype(np.float32).view(np.uint32)
                v.shard_set_limits("m", np.where(rng.random(TILES) < 0.33, np.uint32(KEY_ALL), lim).astype(np.uint32))
            elif mode == "tiny_slots":   # 64-record slots: the verdict reports the overflow, round 0 is redone with whole-shard slots
                v.shard_set_slot_records("m", 64)
            v.shard_render_frame("m", shard_max, speculate=mode != "off")
            frames.append(v.download_framebuffer().copy())
        stats = v.shard_stats()
        limits = v.shard_download_limits("m")
        v.close()
        return frames, stats, limits

    res = run_group(world, body)
    for rank, (frames, stats, limits) in enumerate(res):
        for k, fb in enumerate(frames):
            assert np.array_equal(fb, ref[k]), f"rank {rank} frame {k} ({mode}): L-inf {np.abs(fb - ref[k]).max()}"
        assert stats["frames"] == len(POSES)
        assert np.array_equal(limits, res[0][2]), "every rank derives the same limits from the gathered saturation map"
        # (one model = the frame's last: its repair round is decided when the frame is retired, exactly sized, only where needed)
        if mode == "all_refusing":
            assert stats["repair_frames"] == len(POSES) and stats["exchange_rounds"] == 2 * len(POSES)
        elif mode in ("off", "all_open"):
            assert stats["repair_frames"] == 0 and stats["redo_frames"] == 0 and stats["exchange_rounds"] == len(POSES)
        elif mode == "tiny_slots":
            assert stats["redo_frames"] == len(POSES), "every frame's 64-record slots overflow: each is redone once"
        assert stats["wire_bytes"] > 0
    print(mode, world, "rank 0:", res[0][1])


@pytest.mark.parametrize("world,lanes", [(2, 2), (3, 2), (4, 3)])
def test_library_frame_loop_with_peers_and_frames_in_flight(world, lanes):
    """frames_in_flight = L: frame k is enqueued before the verdict of frame k - L + 1 is looked at.  Synchronised reads and
    un-synchronised runs, natural limits and limits that force the repair round of every frame."""
    g = _scene()
    ref = _single_frames(g)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group, lanes=lanes)
        bad = []
        for k, pose in enumerate(POSES):          # readback after every call completes every frame in flight
            _uniforms(v, pose)
            v.shard_render_frame("m", shard_max)
            fb = v.download_framebuffer()
            if not np.array_equal(fb, ref[k]):
                bad.append(("sync", k, float(np.abs(fb - ref[k]).max())))
        for rep in range(3):                      # free-running: verdicts are read one or two calls late
            for k, pose in enumerate(POSES):
                _uniforms(v, pose)
                if rep == 2:
                    v.shard_set_limits("m", np.full(TILES, 0x40400000, np.uint32))
                v.shard_render_frame("m", shard_max)
            fb = v.download_framebuffer()
            if not np.array_equal(fb, ref[len(POSES) - 1]):
                bad.append(("run", rep, float(np.abs(fb - ref[-1]).max())))
        stats = v.shard_stats()
        v.close()
        return bad, stats

    for rank, (bad, stats) in enumerate(run_group(world, body)):
        assert not bad, f"rank {rank}: {bad}"
        assert stats["frames"] == 4 * len(POSES) and stats["repair_frames"] >= len(POSES)


@pytest.mark.parametrize("world,lanes", [(1, 2), (2, 2), (2, 3)])
def test_viewport_resized_between_calls_with_frames_in_flight(world, lanes):
    """with frames in flight lane 0 is the owner itself, and gsx_update_camera resizes it without retiring what is in
    flight.  A frame that is retired by a LATER call must be gathered under the viewport it was enqueued with — not the caller's new
    one (wrong row bytes and band offsets; a reallocated framebuffer).  Nothing that completes frames is called inside the loop:
    frame k is read one call late, from its lane, at ITS size."""
    g = _scene()
    sizes = [(W, H), (176, 120), (W, H), (240, 168), (176, 120), (W, H), (240, 168), (176, 120)]
    poses = [POSES[k % 3] for k in range(len(sizes))]   # (a coherent stretch: natural limits, few repairs)
    ref = []
    with MultiModelViewer() as v1:
        v1.add_model("m", N)
        v1.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        for pose, size in zip(poses, sizes):
            _uniforms(v1, pose, size)
            v1.render_frame(["m"])
            ref.append(v1.download_framebuffer().copy())

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group, lanes=lanes)
        bad = []
        for k, (pose, size) in enumerate(zip(poses, sizes)):
            _uniforms(v, pose, size)
            v.shard_render_frame("m", shard_max)
            if k >= lanes - 1:   # the frame of call k - (lanes - 1) has just been retired; its lane is idle until call k + 1
                j = k - (lanes - 1)
                fb = v.debug_download_lane_framebuffer(j % lanes, sizes[j])
                if fb.shape != ref[j].shape or not np.array_equal(fb, ref[j]):
                    bad.append((j, sizes[j], float(np.abs(fb - ref[j]).max()) if fb.shape == ref[j].shape else "shape"))
        fb = v.download_framebuffer()
        if not np.array_equal(fb, ref[-1]):
            bad.append(("last", float(np.abs(fb - ref[-1]).max())))
        v.close()
        return bad

    for rank, bad in enumerate(run_group(world, body)):
        assert not bad, f"rank {rank} (world {world}, {lanes} lanes): {bad}"


def test_a_rank_that_leaves_the_loop_is_an_error_not_a_hang():
    """Rank 1 stops calling after the first frame.  Rank 0's next frame gets GSX_ERR_RCCL from the rendezvous within the
    group's timeout — from gsx_shard_render_frame, through the C ABI — and so does every later collective call."""
    g = _scene()
    world = 2
    t_fail = [None]

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        _uniforms(v, POSES[0])
        v.shard_render_frame("m", shard_max)
        v.poll()
        if rank == 1:
            time.sleep(3.0)  # stays alive (its buffers too) while rank 0 runs into the timeout
            v.close()
            return "left"
        t0 = time.perf_counter()
        with pytest.raises(GsxError) as e:
            _uniforms(v, POSES[1])
            v.shard_render_frame("m", shard_max)
            v.poll()
        t_fail[0] = time.perf_counter() - t0
        assert e.value.status == _lib.GSX_ERR_RCCL and "waited" in str(e.value)
        with pytest.raises(GsxError) as e2:   # the group stays failed: no later call blocks either
            v.shard_render_frame("m", shard_max)
        assert e2.value.status == _lib.GSX_ERR_RCCL
        v.close()
        return "failed cleanly"

    res = run_group(world, body, timeout_ms=1000)
    assert res == ["failed cleanly", "left"]
    assert 0.9 < t_fail[0] < 2.9, t_fail[0]


def test_ranks_that_disagree_about_a_collective_fail_on_every_rank():
    """Rank 1 is told a different slot size (a violated contract: slot sizes must be global).  The group notices that the
    two all-to-alls do not match and every rank gets GSX_ERR_RCCL."""
    g = _scene()
    world = 2

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        _uniforms(v, POSES[0])
        v.shard_set_slot_records("m", 64 if rank == 0 else 128)
        with pytest.raises(GsxError) as e:
            v.shard_render_frame("m", shard_max)
        assert e.value.status == _lib.GSX_ERR_RCCL and "disagree" in str(e.value)
        v.close()
        return True

    assert run_group(world, body, timeout_ms=5000) == [True, True]


def test_custom_transport_from_the_caller():
    """gsx_viewer_comm_init_custom: the caller brings the two collectives as C function pointers (here: Python callbacks that
    enqueue hipMemcpyAsync on the stream they are handed).  One rank — the point is the boundary: the library calls out for
    every exchange and the frame still equals the single viewer's."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemcpyAsync.restype = C.c_int
    calls = {"a2a": 0, "ag": 0}

    def all_to_all(send, recv, nbytes, stream):
        calls["a2a"] += 1
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes, 3, stream) == 0 else _lib.GSX_ERR_HIP

    def all_gather(send, recv, nbytes, stream):
        calls["ag"] += 1
        if send == recv:
            return 0
        return 0 if hip.hipMemcpyAsync(recv, send, nbytes, 3, stream) == 0 else _lib.GSX_ERR_HIP

    g = _scene()
    ref = _single_frames(g)
    with MultiModelViewer() as v:
        v.add_model("m", N)
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        v.comm_init_custom(1, 0, all_to_all, all_gather)
        for k, pose in enumerate(POSES):
            _uniforms(v, pose)
            if k == 3:
                v.shard_set_limits("m", np.full(TILES, 0x40400000, np.uint32))
            v.shard_render_frame("m", N)
            assert np.array_equal(v.download_framebuffer(), ref[k]), f"frame {k}"
        st = v.shard_stats()
        assert calls["a2a"] == st["exchange_rounds"] >= len(POSES) + 1 and calls["ag"] >= 2 * len(POSES)
        # a transport that fails: its status comes back through the C ABI
        v.comm_destroy()
        v.comm_init_custom(1, 0, lambda *a: _lib.GSX_ERR_IO, all_gather)
        with pytest.raises(GsxError) as e:
            v.shard_render_frame("m", N)
        assert e.value.status == _lib.GSX_ERR_IO and "custom transport" in str(e.value)


def test_full_size_cfg4_world4_through_the_library():
    """BASELINE.json configs[3] (10 M Gaussians, 1920x1080), index-sharded over 4 ranks, every rank inside
    gsx_shard_render_frame: along the bench orbit and across a jump, each rank's gathered frame equals the single-GPU frame."""
    from wgpu_3dgs_viewer_app_amd import scene

    n, sh, w, h, seed = scene.CONFIGS["cfg4"]
    g = scene.synthetic_gaussians(n, seed, sh)
    poses = (0, 1, 2, 3, 120, 121)
    ref = _single_frames(g, poses, (w, h))
    world = 4

    def body(rank, group):
        v, shard_max = _rank_viewer(g, n, rank, world, group)
        bad = []
        for k, pose in enumerate(poses):
            _uniforms(v, pose, (w, h))
            v.shard_render_frame("m", shard_max)
            fb = v.download_framebuffer()
            if not np.array_equal(fb, ref[k]):
                bad.append((pose, float(np.abs(fb - ref[k]).max())))
        stats = v.shard_stats()
        v.close()
        return bad, stats

    res = run_group(world, body, timeout_ms=120000)
    for rank, (bad, stats) in enumerate(res):
        assert not bad, f"rank {rank}: frames differ from the single-GPU frames: {bad}"
    print("cfg4 world 4:", res[0][1])


# ---- layered models inside the library call: gsx_shard_render_frame_keys ---------------------------------

LAYERS = {"a": (4000, 11, camera.ModelTransform(pos=np.array([0.0, 0.0, 1.5], np.float32))),
          "b": (3000, 12, common.odd_transform()),
          "c": (2500, 13, camera.ModelTransform(pos=np.array([-1.0, 0.3, -2.0], np.float32), rot=np.array([0, 40, 0], np.float32))),
          "d": (2000, 14, camera.ModelTransform(pos=np.array([1.5, -0.2, 0.0], np.float32), scale=np.array([0.8, 0.8, 0.8], np.float32)))}
LW, LH = 240, 160
LTILES = ((LH + 15) // 16, (LW + 15) // 16)
LPOSES = (30, 31, 32, 150, 151, 90)   # the far -> near order of the four models changes along the way


def _layer_scenes():
    return {k: common.small_scene(n, seed, scale_mul=8.0) for k, (n, seed, _) in LAYERS.items()}


def _layer_keys(pose):
    return parallel.model_render_keys(camera.orbit_pose(pose).pos, {k: mt for k, (_, _, mt) in LAYERS.items()})


def _layer_viewer(scenes, rank, world, group=None, lanes=1):
    v = MultiModelViewer()
    v.set_render_options(frames_in_flight=lanes)
    for k, g in scenes.items():
        s0, c = parallel.shard_range(g.shape[0], rank, world)
        v.add_model(k, c)
        v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, g[s0:s0 + c])
        mt = LAYERS[k][2]
        v.update_model_transform(k, mt.pos, mt.quat(), mt.scale)
    if group is not None:
        v.comm_init_group(group, rank)
    return v


def _layer_reference(scenes, poses=LPOSES):
    out = []
    v = _layer_viewer(scenes, 0, 1)
    for pose in poses:
        _uniforms(v, pose, (LW, LH))
        v.render_frame(_layer_keys(pose))
        out.append(v.download_framebuffer().copy())
    v.close()
    return out


@pytest.fixture(params=["model_by_model", "all_at_once"])
def layer_schedule(request):
    return request.param


@pytest.mark.parametrize("world,mode,lanes", [(2, "natural", 1), (4, "natural", 1), (3, "all_refusing", 1), (4, "tiny_slots", 1),
                                              (3, "off", 1), (2, "natural", 2), (4, "all_refusing", 2), (3, "tiny_slots", 2), (2, "natural", 3)])
def test_layered_models_inside_the_library_call(world, mode, lanes, monkeypatch, layer_schedule):
    """Four models with their own TRS, layered far -> near in an order that changes with the camera, every model
    index-sharded over `world` ranks, ONE C-ABI call per frame and rank: equal to gsx_render_frame(keys) on one GPU.
    layer_schedule: model by model with host-decided repairs (the default) / all models at once with the inner models' repair
    exchanges always enqueued and decided on the device (GSX_SHARD_LAYER_PIPELINE=0)."""
    if layer_schedule == "all_at_once":
        monkeypatch.setenv("GSX_SHARD_LAYER_PIPELINE", "0")
    scenes = _layer_scenes()
    ref = _layer_reference(scenes)
    assert len({tuple(_layer_keys(p)) for p in LPOSES}) >= 2, "the layer order must change along the path"
    assert not np.array_equal(ref[0], ref[3])

    def body(rank, group):
        v = _layer_viewer(scenes, rank, world, group, lanes)
        shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
        bad = []
        for rep in range(2 if lanes > 1 else 1):
            for k, pose in enumerate(LPOSES):
                _uniforms(v, pose, (LW, LH))
                keys = _layer_keys(pose)
                for key in keys:
                    if mode == "all_refusing":
                        v.shard_set_limits(key, np.full(LTILES, 0x40400000, np.uint32))
                    elif mode == "tiny_slots":
                        v.shard_set_slot_records(key, 32)
                v.shard_render_frame_keys(keys, [shard_max[x] for x in keys], speculate=mode != "off")
                if rep == 0 or k == len(LPOSES) - 1:   # second pass (frames in flight): free-running, read the last frame only
                    fb = v.download_framebuffer()
                    if not np.array_equal(fb, ref[k]):
                        bad.append((rep, k, float(np.abs(fb - ref[k]).max())))
        stats = v.shard_stats()
        v.close()
        return bad, stats

    res = run_group(world, body)
    for rank, (bad, stats) in enumerate(res):
        assert not bad, f"rank {rank} ({mode}, {lanes} lanes): {bad}"
        if mode == "all_refusing":
            assert stats["repair_frames"] == stats["frames"]
        if mode == "tiny_slots":
            assert stats["redo_frames"] == stats["frames"]
        if mode == "off":
            assert stats["exchange_rounds"] == 4 * stats["frames"] and stats["repair_frames"] == 0
    print("layered", mode, world, lanes, res[0][1])


@pytest.mark.parametrize("world,mode,lanes", [(2, "natural", 2), (3, "all_refusing", 2), (3, "tiny_slots", 2), (2, "natural", 3)])
def test_layered_frames_that_go_out_model_by_model_are_whole_frames(world, mode, lanes):
    """With frames in flight a layered frame is enqueued model by model, half of it by the call after its own — after the caller has moved
    the camera on.  Nothing that completes frames is called inside the loop; frame k is looked at one call late, in its lane's
    framebuffer (gsx_debug_download_lane_framebuffer: frame k was retired by call k + 1): EVERY frame equals the single-viewer frame,
    incl. the frames whose repairs are exchanged between two models and the ones that are redone because their slots were too small."""
    scenes = _layer_scenes()
    ref = _layer_reference(scenes)
    poses = list(LPOSES) * 2

    def body(rank, group):
        v = _layer_viewer(scenes, rank, world, group, lanes)
        shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
        bad = []
        for k, pose in enumerate(poses):
            _uniforms(v, pose, (LW, LH))
            keys = _layer_keys(pose)
            for key in keys:
                if mode == "all_refusing":
                    v.shard_set_limits(key, np.full(LTILES, 0x40400000, np.uint32))
                elif mode == "tiny_slots":
                    v.shard_set_slot_records(key, 32)
            v.shard_render_frame_keys(keys, [shard_max[x] for x in keys])
            if k >= lanes - 1:   # the frame of call k - (lanes - 1) has just been retired; its lane is not used again before call k + 1
                j = k - (lanes - 1)
                fb = v.debug_download_lane_framebuffer(j % lanes)
                if not np.array_equal(fb, ref[j % len(LPOSES)]):
                    bad.append((j, float(np.abs(fb - ref[j % len(LPOSES)]).max())))
        fb = v.download_framebuffer()   # (completes what is in flight: the newest frame)
        if not np.array_equal(fb, ref[(len(poses) - 1) % len(LPOSES)]):
            bad.append(("last", float(np.abs(fb - ref[(len(poses) - 1) % len(LPOSES)]).max())))
        stats = v.shard_stats()
        v.close()
        return bad, stats

    res = run_group(world, body)
    for rank, (bad, stats) in enumerate(res):
        assert not bad, f"rank {rank} ({mode}, {lanes} lanes): {bad}"
        assert stats["frames"] == len(poses)
        if mode == "all_refusing":
            assert stats["repair_frames"] == stats["frames"]
        if mode == "tiny_slots":
            assert stats["redo_frames"] == stats["frames"]


def test_cfg5_shape_mask_selection_edit_inside_the_library_call():
    """cfg5's shape at test size through gsx_shard_render_frame_keys: four models with TRS, a `0 - 1` mask on one, a rect
    selection evaluated on the sharded frame (gsx_postprocess per shard) and an HSV edit + highlight of what it selected —
    every rank's frame equals the single-viewer frame, the shards' selections partition the single viewer's."""
    from wgpu_3dgs_viewer_app_amd import query
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    world = 3
    scenes = _layer_scenes()
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 0.5], np.float32), scale=np.array([3.0, 2.5, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.2, 0.1, 0.8], np.float32), scale=np.array([1.2, 1.0, 1.4], np.float32))]
    rect = query.QueryPod.rect((60.0, 40.0), (180.0, 120.0), query.QuerySelectionOp.Set)
    edit = query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.45, 1.2, 0.9), 0.1, 0.3, 1.0, 0.8)
    poses = (30, 31, 32)

    def drive(v, render):
        MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "b", shapes)
        frames = []
        for step, pose in enumerate(poses):
            _uniforms(v, pose, (LW, LH))
            v.update_query(rect if step == 0 else query.QueryPod.none())
            if step == 1:
                v.update_selection_edit_with_pod(edit)
                v.update_selection_highlight((1.0, 0.0, 1.0, 0.3))
            render(v, _layer_keys(pose))
            for k in LAYERS:
                v.postprocessor.postprocess(k)
            frames.append(v.download_framebuffer().copy())
        nsel = sum(int(np.unpackbits(v.models[k].gaussian_buffers.selection_buffer.download().view(np.uint8)).sum()) for k in LAYERS)
        return frames, nsel

    single = _layer_viewer(scenes, 0, 1)
    ref, nsel_ref = drive(single, lambda v, keys: v.render_frame(keys))
    single.close()
    assert nsel_ref > 100 and not np.array_equal(ref[0], ref[1])

    def body(rank, group):
        v = _layer_viewer(scenes, rank, world, group)
        shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
        out = drive(v, lambda vv, keys: vv.shard_render_frame_keys(keys, [shard_max[x] for x in keys]))
        v.close()
        return out

    res = run_group(world, body)
    assert sum(r[1] for r in res) == nsel_ref, "the shards' selections partition the single-viewer selection"
    for rank, (frames, _) in enumerate(res):
        for i, fb in enumerate(frames):
            assert np.array_equal(fb, ref[i]), f"rank {rank} frame {i}: L-inf {np.abs(fb - ref[i]).max()}"


def _cfg5_setup(v, rank, world, scenes, tr):
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    for k, g in scenes.items():
        s0, c = parallel.shard_range(g.shape[0], rank, world)
        v.add_model(k, c)
        v.models[k].gaussian_buffers.gaussians_buffer.update_range(0, g[s0:s0 + c])
        v.update_model_transform(k, tr[k].pos, tr[k].quat(), tr[k].scale)
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([3.0, 3.0, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.0, 0.0, 2.5], np.float32), scale=np.array([1.5, 1.5, 1.5], np.float32))]
    MaskEvaluator(v).evaluate(MaskOp.parse("0 - 1"), "a", shapes)


def test_full_size_cfg5_world4_and_rccl_world1_through_the_library():
    """BASELINE.json configs[4]: 4 models x 6 M Gaussians (24 M, SH-3), TRS per model, `0 - 1` mask on one, a rect selection
    with an HSV edit, 3840x2160 — ONE gsx_shard_render_frame_keys call per frame: 4 ranks as threads over the in-process
    group, and one rank over real RCCL.  Equal to gsx_render_frame(keys) on one GPU, bit for bit."""
    from wgpu_3dgs_viewer_app_amd import query, scene

    n_total, sh, w, h, seed = scene.CONFIGS["cfg5"]
    n = n_total // 4
    tr = {"a": camera.ModelTransform(pos=np.array([0.0, 0.0, 2.5], np.float32)),
          "b": camera.ModelTransform(pos=np.array([2.0, 0.2, -1.0], np.float32), rot=np.array([0, 35, 0], np.float32)),
          "c": camera.ModelTransform(pos=np.array([-2.5, -0.1, -0.5], np.float32), scale=np.array([0.9, 0.9, 0.9], np.float32)),
          "d": camera.ModelTransform(pos=np.array([0.3, -0.2, 0.5], np.float32), rot=np.array([20, -35, 50], np.float32),
                                     scale=np.array([1.2, 0.9, 1.1], np.float32))}
    scenes = {k: scene.synthetic_gaussians(n, seed + i, sh) for i, k in enumerate(tr)}
    poses = (0, 1, 2, 100, 101)
    rect = query.QueryPod.rect((1200.0, 600.0), (2600.0, 1500.0), query.QuerySelectionOp.Set)
    edit = query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.5, 1.0, 1.2), 0.1, 0.2, 1.0, 0.9)

    def drive(v, render):
        frames = []
        for step, pose in enumerate(poses):
            _uniforms(v, pose, (w, h))
            keys = parallel.model_render_keys(camera.orbit_pose(pose).pos, tr)
            v.update_query(rect if step == 1 else query.QueryPod.none())
            if step == 2:
                v.update_selection_edit_with_pod(edit)
            render(v, keys)
            if step == 1:
                for k in keys:
                    v.postprocessor.postprocess(k)
            frames.append(v.download_framebuffer().copy())
        return frames

    single = MultiModelViewer()
    _cfg5_setup(single, 0, 1, scenes, tr)
    ref = drive(single, lambda v, keys: v.render_frame(keys))
    single.close()
    assert not np.array_equal(ref[1], ref[2]), "the edit must change the frame"

    # one rank over real RCCL (send / recv to itself, like the world-1 tests of tests/test_gpu_sharded.py)
    import os

    os.environ["GSX_COMM_SELF_VIA_RCCL"] = "1"
    try:
        v = MultiModelViewer()
        _cfg5_setup(v, 0, 1, scenes, tr)
        uid = (C.c_uint8 * 128)()
        _lib.check(v._L.gsx_comm_unique_id(uid))
        v.comm_init_rccl(1, 0, bytes(uid))
        got = drive(v, lambda vv, keys: vv.shard_render_frame_keys(keys, [n] * len(keys)))
        v.close()
    finally:
        del os.environ["GSX_COMM_SELF_VIA_RCCL"]
    for i, fb in enumerate(got):
        assert np.array_equal(fb, ref[i]), f"RCCL world 1, frame {i}: L-inf {np.abs(fb - ref[i]).max()}"

    world = 4

    def body(rank, group):
        v = MultiModelViewer()
        _cfg5_setup(v, rank, world, scenes, tr)
        v.comm_init_group(group, rank)
        frames = drive(v, lambda vv, keys: vv.shard_render_frame_keys(keys, [(n + world - 1) // world] * len(keys)))
        bad = [(i, float(np.abs(fb - ref[i]).max())) for i, fb in enumerate(frames) if not np.array_equal(fb, ref[i])]
        stats = v.shard_stats()
        v.close()
        return bad, stats

    res = run_group(world, body, timeout_ms=180000)
    for rank, (bad, stats) in enumerate(res):
        assert not bad, f"cfg5 rank {rank}: frames differ from the single-GPU frames: {bad}"
    print("cfg5 world 4:", res[0][1])


# ---- seeded random walks through the API, sharded against single (the multi-GPU twin of test_fuzz_operation_sequences) --------

def _fuzz_ops(seed, steps=40):
    """The walk is drawn up front: every rank (and the single viewer) replays the same list."""
    rng = np.random.default_rng(7000 + seed)
    ops, pose, size, visible = [], int(rng.integers(0, 240)), (LW, LH), list(LAYERS)
    state = dict(size=1.0, mode=0, deg=3, speculate=True)
    for _ in range(steps):
        op = int(rng.integers(0, 16))
        extra = None
        if op <= 4:
            pose = (pose + int(rng.integers(1, 3))) % 240
        elif op == 5:
            pose = int(rng.integers(0, 240))
        elif op == 6:
            size = [(LW, LH), (LW - 32, LH), (LW, LH + 16), (200, 120), (640, 368)][int(rng.integers(0, 5))]
        elif op == 7:
            visible = [k for k in LAYERS if rng.random() < 0.7] or ["b"]
        elif op == 8:
            extra = ("mask", "abcd"[int(rng.integers(0, 4))], [None, "0", "!1", "0 - 1", "0 | 1"][int(rng.integers(0, 5))])
        elif op == 9:
            extra = ("rect", tuple(float(x) for x in rng.uniform(0, 120, 2)), tuple(float(x) for x in rng.uniform(100, 240, 2)),
                     int(rng.integers(0, 3)))
        elif op == 10:
            extra = ("edit", [0, 1, 3, 5][int(rng.integers(0, 4))], tuple(float(x) for x in rng.uniform(0, 1, 3)),
                     float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-1, 1)), float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.3, 1.5)),
                     float(rng.choice([0.0, 0.4])))
        elif op == 11:
            state.update(mode=int(rng.integers(0, 3)), size=float(rng.choice([0.6, 1.0, 1.5])), deg=int(rng.integers(0, 4)))
        elif op == 12:
            k = "abcd"[int(rng.integers(0, 4))]
            extra = ("trs", k, rng.uniform(-2.0, 2.0, 3).astype(np.float32), rng.uniform(-40, 40, 3).astype(np.float32),
                     rng.uniform(0.7, 1.3, 3).astype(np.float32))
        elif op == 13:
            extra = ("limits", int(rng.integers(0, 1 << 30)), float(rng.uniform(0.2, 0.9)))   # seed of a limit map, share of bounded tiles
        elif op == 14:
            extra = ("slots", int(rng.choice([0, 16, 64, 100000])))
        else:
            state["speculate"] = bool(rng.random() < 0.8)
        ops.append(dict(pose=pose, size=size, visible=list(visible), state=dict(state), extra=extra))
    return ops


def _fuzz_replay(v, ops, render, world, on_frame):
    """Applies the walk to viewer v (a rank's, or the single one: world = 1, render = gsx_render_frame)."""
    from wgpu_3dgs_viewer_app_amd import query
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    shapes = [MaskShape(MaskShapeKind.Box, pos=np.zeros(3, np.float32), scale=np.array([2.0, 2.0, 2.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.5, 0.0, 0.5], np.float32), scale=np.array([1.5, 1.2, 1.5], np.float32))]
    tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
    slots = 0
    for step, o in enumerate(ops):
        e = o["extra"]
        query_on = False
        limits = None
        if e and e[0] == "mask":
            mt = tr[e[1]]
            v.update_model_transform(e[1], mt.pos, mt.quat(), mt.scale)
            MaskEvaluator(v).evaluate(MaskOp.parse(e[2]) if e[2] else None, e[1], shapes)
        elif e and e[0] == "rect":
            v.update_query(query.QueryPod.rect(e[1], e[2], [query.QuerySelectionOp.Set, query.QuerySelectionOp.Add, query.QuerySelectionOp.Remove][e[3]]))
            query_on = True
        elif e and e[0] == "edit":
            v.update_selection_edit_with_pod(query.GaussianEditPod(e[1], e[2], e[3], e[4], e[5], e[6]))
            v.update_selection_highlight((1.0, 0.0, 1.0, e[7]))
        elif e and e[0] == "trs":
            tr[e[1]] = camera.ModelTransform(pos=e[2], rot=e[3], scale=e[4])
            v.update_model_transform(e[1], tr[e[1]].pos, tr[e[1]].quat(), tr[e[1]].scale)
        elif e and e[0] == "limits":
            limits = e
        elif e and e[0] == "slots":
            slots = e[1]
        w, h = o["size"]
        cam = camera.orbit_pose(o["pose"])
        keys = [k for k in parallel.model_render_keys(cam.pos, tr) if k in o["visible"]]
        v.update_camera(cam, (w, h))
        v.update_gaussian_transform(o["state"]["size"], GaussianDisplayMode(o["state"]["mode"]), GaussianShDegree.new(o["state"]["deg"]), False)
        if world > 1:
            tiles = ((h + 15) // 16, (w + 15) // 16)
            for k in keys:
                v.shard_set_slot_records(k, slots)
                if limits is not None:   # the same map on every rank (limits are a global input)
                    r = np.random.default_rng(limits[1])
                    lim = r.uniform(2.0, 9.0, tiles).astype(np.float32).view(np.uint32)
                    v.shard_set_limits(k, np.where(r.random(tiles) < limits[2], lim, np.uint32(KEY_ALL)).astype(np.uint32))
        render(v, keys, o["state"]["speculate"])
        for k in keys:
            v.postprocessor.postprocess(k)
        if query_on:
            v.update_query(query.QueryPod.none())
        on_frame(step, o, v.download_framebuffer())


@pytest.mark.parametrize("seed,world,lanes", [(1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 2, 2), (5, 3, 1)])
def test_fuzz_sharded_against_single(seed, world, lanes):
    """Random walks — camera steps and jumps, viewports (also grids of more than 256 tiles), models shown / hidden, masks, rect
    selections + postprocess, edits + highlight, display mode / size / SH degree, model transforms, imposed limit maps, slot
    sizes from 16 records to ample, speculation on / off — through gsx_shard_render_frame_keys on `world` ranks (threads) and
    through gsx_render_frame on one viewer: every frame of every rank is the same bytes."""
    scenes = _layer_scenes()
    ops = _fuzz_ops(seed)
    ref = {}
    single = _layer_viewer(scenes, 0, 1)
    _fuzz_replay(single, ops, lambda v, keys, spec: v.render_frame(keys), 1, lambda step, o, fb: ref.__setitem__(step, fb.copy()))
    single.close()
    assert len({o["extra"][0] for o in ops if o["extra"]}) >= 4, "the walk must mix kinds of operations"

    def body(rank, group):
        v = _layer_viewer(scenes, rank, world, group, lanes)
        shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
        bad = []

        def check(step, o, fb):
            if not np.array_equal(fb, ref[step]):
                bad.append((step, o["extra"][0] if o["extra"] else None, float(np.abs(fb - ref[step]).max())))

        _fuzz_replay(v, ops, lambda vv, keys, spec: vv.shard_render_frame_keys(keys, [shard_max[k] for k in keys], speculate=spec), world, check)
        stats = v.shard_stats()
        v.close()
        return bad, stats

    for rank, (bad, stats) in enumerate(run_group(world, body, timeout_ms=60000)):
        assert not bad, f"seed {seed} rank {rank}: {bad[:5]}"
        assert stats["frames"] == len(ops)


@pytest.mark.parametrize("world,lanes", [(2, 2), (3, 3)])
def test_selection_edit_set_in_mid_flight_of_sharded_frames(world, lanes):
    """Sharded frames in flight on lanes, nothing read back in between, and the host changes the selection edit with a plain setter
    (gsx_update_selection_edit orders nothing by itself): the library completes the frames that still read the old edit records
    before it prepares the new ones.  Checked frames equal the single viewer's."""
    from wgpu_3dgs_viewer_app_amd import query
    from wgpu_3dgs_viewer_app_amd.query import GaussianEditFlag as F

    g = _scene()
    rng = np.random.default_rng(31)
    sel = rng.integers(0, 2 ** 32, (N + 31) // 32, dtype=np.uint64).astype(np.uint32)
    sel[-1] &= np.uint32((1 << (N % 32)) - 1) if N % 32 else np.uint32(0xFFFFFFFF)
    pods = {0: query.GaussianEditPod(F.ENABLED, (0.3, 1.5, 0.8), 0.25, -0.75, 2.2, 0.6),
            5: query.GaussianEditPod(F.ENABLED | F.OVERRIDE_COLOR, (0.9, 0.2, 0.1), -0.5, 1.5, 0.45, 1.7),
            9: query.GaussianEditPod(F.ENABLED | F.HIDDEN)}
    poses = [57 + k for k in range(14)]
    check = {4, 8, 13}

    def drive(v, render, sel_words):
        v.models["m"].gaussian_buffers.selection_buffer.upload(sel_words)
        v.update_selection_highlight((1.0, 0.0, 1.0, 0.4))
        out = {}
        for k, pose in enumerate(poses):
            if k in pods:
                v.update_selection_edit_with_pod(pods[k])
            _uniforms(v, pose)
            render(v)
            if k in check:
                out[k] = v.download_framebuffer().copy()
        return out

    with MultiModelViewer() as s:
        s.add_model("m", N)
        s.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        ref = drive(s, lambda v: v.render_frame(["m"]), sel)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group, lanes=lanes)
        s0, c = parallel.shard_range(N, rank, world)
        bits = np.unpackbits(sel.view(np.uint8), bitorder="little")[:N][s0:s0 + c]
        words = np.packbits(np.concatenate([bits, np.zeros((-c) % 32, np.uint8)]), bitorder="little").view(np.uint32)
        out = drive(v, lambda vv: vv.shard_render_frame("m", shard_max), words)
        v.close()
        return out

    for rank, out in enumerate(run_group(world, body)):
        for k in sorted(check):
            assert np.array_equal(out[k], ref[k]), f"rank {rank} frame {k}: L-inf {np.abs(out[k] - ref[k]).max()}"
    assert not np.array_equal(ref[4], ref[8])


@pytest.mark.parametrize("world,root,lanes", [(3, 1, 1), (4, 0, 2)])
def test_bands_gathered_to_one_rank_over_the_group(world, root, lanes):
    """gsx_shard_set_gather_root over the in-process group: the root's framebuffer holds every frame whole, every other rank
    keeps (at least) its own band of tile rows and receives nothing; ranks that name different roots fail the frame."""
    g = _scene()
    ref = _single_frames(g)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group, lanes=lanes)
        v.shard_set_gather_root(root)
        frames = []
        for pose in POSES:
            _uniforms(v, pose)
            v.shard_render_frame("m", shard_max)
            fb = v.download_framebuffer().copy()
            frames.append((fb, v.shard_get_band_edges(world).copy()))   # (the bands are balanced: the layout moves from frame to frame)
        v.close()
        return frames

    for rank, frames in enumerate(run_group(world, body)):
        for k, (fb, edges) in enumerate(frames):
            lo, hi = (0, H) if rank == root else (min(16 * int(edges[rank]), H), min(16 * int(edges[rank + 1]), H))
            assert np.array_equal(fb[lo:hi], ref[k][lo:hi]), f"rank {rank} frame {k}"

    def disagree(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        v.shard_set_gather_root(rank % 2)
        _uniforms(v, POSES[0])
        try:
            v.shard_render_frame("m", shard_max)
            v.poll()
        finally:
            v.close()

    # the feedback all-gather carries every rank's root; the verdict says whether they agree, BEFORE any rank gathers (over RCCL a
    # Send nobody receives would hang): every rank fails with the same message
    with pytest.raises(GsxError) as e:
        run_group(world, disagree, timeout_ms=2000)
    assert e.value.status == _lib.GSX_ERR_INVALID_ARG and "different gather roots" in str(e.value)


# ---- band layout: balanced by the previous frame's per-row work (SURVEY 8e "load-balanced") ----
def _open_sky_scene(n=30000):
    """Everything below the horizon: the upper half of the screen stays empty — equal bands would leave the top ranks idle."""
    g = common.small_scene(n, 93, scale_mul=14.0)
    g["pos"][:, 1] = -np.abs(g["pos"][:, 1]) - 0.3
    return g


def test_bands_are_balanced_by_the_previous_frames_work():
    # 40 x 60 tiles (round 5: a rank's blocks are sized for its band — finer lists, a sharper per-row work figure — and at 30 rows one
    # row alone weighed 1.46 x an eighth of the frame: bands are whole rows)
    world, size = 8, (640, 960)
    g = _open_sky_scene()
    poses = (10, 11, 12, 13, 14, 15, 16, 130, 131, 132)   # (a jump: the work moves, the edges follow)
    ref = _single_frames(g, poses, size)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, g.shape[0], rank, world, group)
        frames, edges, stats = [], [], []
        for pose in poses:
            _uniforms(v, pose, size)
            v.shard_render_frame("m", shard_max)
            frames.append(v.download_framebuffer().copy())
            edges.append(v.shard_get_band_edges(world).copy())
            stats.append(v.shard_stats())
        v.close()
        return frames, edges, stats

    res = run_group(world, body)
    for rank, (frames, edges, stats) in enumerate(res):
        for k, fb in enumerate(frames):
            assert np.array_equal(fb, ref[k]), f"rank {rank} frame {k}: L-inf {np.abs(fb - ref[k]).max()}"
        for k in range(len(poses)):
            assert np.array_equal(edges[k], res[0][1][k]), "every rank derives the same edges"
    edges = res[0][1]
    assert np.array_equal(edges[0], np.arange(world + 1) * 8), "first frame: equal bands (nothing is known yet)"
    assert any(not np.array_equal(edges[k], edges[k + 1]) for k in range(1, len(poses) - 1)), "the edges move with the work"
    assert edges[-1][1] > 8, f"the empty sky must go to few ranks: {edges[-1]}"
    stats = res[0][2]
    work = [st["last_work_permille"] / 1000.0 for st in stats]            # busiest rank's walked entries / mean, frame by frame
    entries = [st["last_entries_max"] * world / max(st["last_entries_sum"], 1) for st in stats]
    print("edges", [e.tolist() for e in edges], "work max/mean", work, "entries max/mean", [round(x, 2) for x in entries])
    # frame 0 ran on equal bands (its verdict is stats[0]); the steady frames on bands balanced by the frame before
    assert work[0] > 1.5, f"equal bands on this scene: {work[0]}"
    assert max(work[3:7]) <= 1.25, f"balanced bands: busiest rank / mean of the work = {work[3:7]}"
    assert max(entries[3:7]) < entries[0], (entries[0], entries[3:7])


def test_the_parallel_verdict_arithmetic_posts_what_the_one_thread_walk_posts(monkeypatch):
    """k_shard_verify's last workgroup derives the work shares and the next band edges from prefix sums, all threads at once; frames of
    more than 1024 tile rows (and GSX_SHARD_VERIFY_SERIAL=1) take the one-thread walk over the rows it replaced.  Same gathered data in,
    same edges, same figures out — frame by frame, incl. the frames where the edges move and the camera jump."""
    world, size = 4, (640, 960)
    g = _open_sky_scene()
    poses = (10, 11, 12, 13, 130, 131, 132)

    def run():
        def body(rank, group):
            v, shard_max = _rank_viewer(g, g.shape[0], rank, world, group)
            edges, stats, sums = [], [], []
            for pose in poses:
                _uniforms(v, pose, size)
                v.shard_render_frame("m", shard_max)
                fb = v.download_framebuffer()
                sums.append(int(np.frombuffer(fb.tobytes(), np.uint32).astype(np.uint64).sum()))
                edges.append(v.shard_get_band_edges(world).tolist())
                st = v.shard_stats()
                stats.append((st["last_work_permille"], st["last_entries_max"], st["last_entries_sum"], st["last_slot_records"], st["repair_frames"]))
            v.close()
            return edges, stats, sums
        return run_group(world, body)[0]

    monkeypatch.delenv("GSX_SHARD_VERIFY_SERIAL", raising=False)
    parallel_edges, parallel_stats, parallel_sums = run()
    monkeypatch.setenv("GSX_SHARD_VERIFY_SERIAL", "1")
    serial_edges, serial_stats, serial_sums = run()
    assert any(e != parallel_edges[0] for e in parallel_edges[1:]), "the scene must move the edges, or the comparison says nothing"
    assert serial_edges == parallel_edges
    assert serial_stats == parallel_stats
    assert serial_sums == parallel_sums


@pytest.mark.parametrize("edges", [(0, 1, 1, 9, 10), (0, 0, 5, 5, 10), (0, 10, 10, 10, 10)])
def test_forced_band_edges_with_empty_and_uneven_bands(edges):
    world = 4
    g = _scene()
    ref = _single_frames(g)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        v.shard_set_band_edges(world, np.array(edges, np.uint32))
        frames = []
        for pose in POSES:
            _uniforms(v, pose)
            v.shard_render_frame("m", shard_max)
            frames.append(v.download_framebuffer().copy())
        got = v.shard_get_band_edges(world)
        v.close()
        return frames, got

    for rank, (frames, got) in enumerate(run_group(world, body)):
        assert tuple(got) == edges
        for k, fb in enumerate(frames):
            assert np.array_equal(fb, ref[k]), f"rank {rank} frame {k}: L-inf {np.abs(fb - ref[k]).max()}"


def test_forced_band_edges_that_no_longer_cover_the_viewport_fail_the_frame():
    """gsx_shard_set_band_edges validates against the viewport of the moment; a later, taller viewport must not be rendered with
    bottom tile rows nobody owns: the frame fails with GSX_ERR_INVALID_ARG on every rank until the edges are set again."""
    world = 2
    g = _scene()
    tall = (W, 2 * H)
    ref = _single_frames(g, POSES[:1], tall)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, N, rank, world, group)
        _uniforms(v, POSES[0])
        v.shard_set_band_edges(world, np.array([0, 4, TILES[0]], np.uint32))
        v.shard_render_frame("m", shard_max)
        v.poll()
        _uniforms(v, POSES[0], tall)
        with pytest.raises(GsxError) as ei:
            v.shard_render_frame("m", shard_max)
        assert ei.value.status == _lib.GSX_ERR_INVALID_ARG and "band edges" in str(ei.value)
        v.shard_set_band_edges(world, np.array([0, 4, (tall[1] + 15) // 16], np.uint32))
        v.shard_render_frame("m", shard_max)
        fb = v.download_framebuffer().copy()
        v.close()
        return fb

    for rank, fb in enumerate(run_group(world, body)):
        assert np.array_equal(fb, ref[0]), f"rank {rank}: L-inf {np.abs(fb - ref[0]).max()}"


def test_balanced_bands_gathered_to_one_rank_with_frames_in_flight():
    world, lanes, size = 4, 2, (640, 480)
    g = _open_sky_scene(20000)
    poses = tuple(range(20, 32))
    ref = _single_frames(g, poses, size)

    def body(rank, group):
        v, shard_max = _rank_viewer(g, g.shape[0], rank, world, group, lanes=lanes)
        v.shard_set_gather_root(1)
        frames = []
        for pose in poses:
            _uniforms(v, pose, size)
            v.shard_render_frame("m", shard_max)
            if rank == 1:
                frames.append(v.download_framebuffer().copy())
            else:
                v.poll()
        edges = v.shard_get_band_edges(world)
        v.close()
        return frames, edges

    res = run_group(world, body)
    for k, fb in enumerate(res[1][0]):
        assert np.array_equal(fb, ref[k]), f"root frame {k}: L-inf {np.abs(fb - ref[k]).max()}"
    assert not np.array_equal(res[0][1], np.arange(world + 1) * 8), "balanced edges were in use"
