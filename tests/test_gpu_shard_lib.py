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
