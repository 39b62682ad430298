"""GPU: the multi-GPU path on ONE device.  `world` ranks run as threads, each with its own viewer, stream and index
shard, and execute the REAL protocol of ``parallel.ShardedViewer`` (pack by destination band -> exchange -> import
with the tile filter -> sort -> band-restricted progressive compositing -> feedback -> optional verified second
exchange -> band all-gather); only the collectives are replaced by device copies in the order RCCL delivers them
(tests/common.py ThreadComm).  Every rank's assembled frame must equal the single-viewer frame BIT FOR BIT whatever
the prediction was.  The pack kernel is also checked against the oracle's routing."""
import ctypes as C

import numpy as np
import pytest

import oracle
from tests import common
from tests.oracle_stages import pack_by_destination
from wgpu_3dgs_viewer_app_amd import _lib, camera, parallel
from wgpu_3dgs_viewer_app_amd.hip_stages import HipStages

pytestmark = pytest.mark.gpu
KEY_ALL = parallel.KEY_ALL
N, W, H = 9000, 208, 152
POSES = (57, 58, 61, 90)


def _scene():
    return common.small_scene(N, 91, scale_mul=14.0)  # opaque enough that tiles saturate


def _single_frames(g):
    out = []
    st = HipStages()
    st.load_shard("shard", g, 0, g.shape[0])
    for pose in POSES:
        st.set_uniforms("shard", camera.orbit_pose(pose), (W, H))
        st.render_local("shard")
        out.append((st.framebuffer().copy(), dict(st.stats("shard"))))
    st.close()
    return out


@pytest.mark.parametrize("world,mode", [(2, "off"), (3, "natural"), (8, "natural"), (2, "all_saturated"), (5, "all_open"),
                                        (4, "stale"), (3, "tiny_slots")])
def test_threaded_world_matches_single_viewer(world, mode):
    g = _scene()
    ref = _single_frames(g)
    tiles = ((H + 15) // 16, (W + 15) // 16)
    assert ref[0][0][..., 3].min() < 1e-4, "the scene must saturate some pixels"

    def rank_main(rank, comm):
        s0, c = parallel.shard_range(N, rank, world)
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm)
        v.load_shard(g[s0:s0 + c], s0, N)
        frames, rounds = [], []
        rng = np.random.default_rng(5)
        for pose in POSES:
            if mode == "all_saturated":   # every tile refuses all but the nearest records: verified second exchange
                v._limit = np.full(tiles, 0x40400000, np.uint32)
            elif mode == "all_open":
                v._limit = np.full(tiles, KEY_ALL, np.uint32)
            elif mode == "stale":   # limits unrelated to the frame (same on every rank): random depths 2..8, a third unbounded
                lim = rng.uniform(2.0, 8.0, tiles).astype(np.float32).view(np.uint32)
                v._limit = np.where(rng.random(tiles) < 0.33, np.uint32(KEY_ALL), lim).astype(np.uint32)
            elif mode == "off":
                v.speculate = False
            elif mode == "tiny_slots":   # round 0's slots hold 64 records: the verdict reports the overflow, round 0 is redone
                v.force_slot = 64
            v.render_frame(camera.orbit_pose(pose), (W, H))
            if mode == "tiny_slots":
                assert v.last_verdict["overflow"] is False and v.last_verdict["max_records"] > 64
            v.poll()
            frames.append(v.framebuffer().copy())
            rounds.append(v.rounds)
        sent = comm.bytes_sent
        v.close()
        return frames, rounds, sent

    res = common.run_ranks(world, rank_main)
    for rank, (frames, rounds, _) in enumerate(res):
        for k, fb in enumerate(frames):
            assert np.array_equal(fb, ref[k][0]), (f"rank {rank} frame {k} ({mode}) differs from the single-GPU frame: "
                                                   f"L-inf {np.abs(fb - ref[k][0]).max()}")
        if mode in ("off", "all_open"):
            assert rounds == [1] * len(POSES)
        elif mode == "all_saturated":
            assert rounds == [2] * len(POSES)
        elif mode == "natural":
            assert rounds[0] == 1  # no prediction yet: one full exchange
    print(mode, "rounds", res[0][1], "bytes sent by rank 0", res[0][2])


def test_speculation_sends_fewer_records():
    """The point of the prediction: with it, fewer bytes cross the links than with one full exchange."""
    g = _scene()
    world = 4

    def run(speculate):
        def rank_main(rank, comm):
            s0, c = parallel.shard_range(N, rank, world)
            v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm)
            v.speculate = speculate
            v.load_shard(g[s0:s0 + c], s0, N)
            for pose in (57, 57, 57, 58):
                v.render_frame(camera.orbit_pose(pose), (W, H))
                v.poll()
            v.close()
            return comm.bytes_sent
        return sum(common.run_ranks(world, rank_main))

    full, spec = run(False), run(True)
    assert spec < full, (spec, full)


def test_pack_matches_oracle_routing():
    import torch

    world = 3
    cam = camera.orbit_pose(57)
    g = _scene()
    f = common.oracle_frame(cam, W, H)
    rng = np.random.default_rng(3)
    tiles = ((H + 15) // 16, (W + 15) // 16)
    lim = np.where(rng.random(tiles) < 0.4, np.uint32(KEY_ALL), rng.uniform(2.0, 8.0, tiles).astype(np.float32).view(np.uint32)).astype(np.uint32)
    need = rng.random(tiles) < 0.3
    for rank in range(world):
        s0, c = parallel.shard_range(N, rank, world)
        st = HipStages(use_torch=True)
        st.load_shard("shard", g[s0:s0 + c], s0, N)
        st.set_uniforms("shard", cam, (W, H))
        pr = oracle.project(f, *oracle.convert(g[s0:s0 + c]))
        with st.stream_ctx():
            st.begin_frame("shard", world, rank)
            for window in (None, parallel.windows_first(lim), parallel.windows_second(lim, need)):
                send, cnt = st.pack("shard", world, window)
                st.poll()
                torch.cuda.synchronize()
                rsend, rcnt = pack_by_destination(pr, world, f.tiles_x, f.tiles_y, window)
                assert cnt == rcnt, (cnt, rcnt)
                assert np.array_equal(send[: sum(cnt)].cpu().numpy().view(np.uint32), rsend.view(np.uint32)), "packed records differ"
            # windows announced before the projection: geometry-only projection, candidates, sparse shading; the packed
            # records must be the very same, and so must the repair exchange's (travellers shaded on demand)
            w1, w2 = parallel.windows_first(lim), parallel.windows_second(lim, need)
            st.begin_frame("shard", world, rank, w1)
            for window, ref_window in ((None, w1), (w2, w2)):
                send, cnt = st.pack("shard", world, window)
                st.poll()
                torch.cuda.synchronize()
                rsend, rcnt = pack_by_destination(pr, world, f.tiles_x, f.tiles_y, ref_window)
                assert cnt == rcnt, (cnt, rcnt)
                assert np.array_equal(send[: sum(cnt)].cpu().numpy().view(np.uint32), rsend.view(np.uint32)), "lazily packed records differ"
        st.close()


def test_pack_capacity_error():
    import torch

    g = common.small_scene(500, 92)
    st = HipStages(use_torch=True)
    st.load_shard("shard", g, 0, 500)
    st.set_uniforms("shard", camera.orbit_pose(1), (128, 96))
    v = st.viewer
    v.preprocessor.preprocess("shard")
    small = torch.empty((4, 12), dtype=torch.float32, device="cuda")
    counts = (C.c_uint64 * 2)()
    with pytest.raises(_lib.GsxError):
        _lib.check(v._L.gsx_shard_pack(v._h, b"shard", 2, None, small.data_ptr(), 4, counts))
    st.close()


def test_cfg5_like_layered_models_mask_selection_edit():
    """cfg5's shape at test size: several models with their own TRS, a `0 - 1` mask op, a rect selection with an HSV
    edit, sharded over 3 ranks: every rank's frame equals the single-viewer frame bit for bit (the per-Gaussian state —
    mask, selection, edits — lives with the shard; queries are evaluated per shard)."""
    from wgpu_3dgs_viewer_app_amd import query
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    world, w, h = 3, 240, 160
    models = {"a": (4000, 11, camera.ModelTransform(pos=np.array([0.0, 0.0, 1.5], np.float32))),
              "b": (3000, 12, common.odd_transform()),
              "c": (2500, 13, camera.ModelTransform(pos=np.array([-1.0, 0.3, -2.0], np.float32), rot=np.array([0, 40, 0], np.float32))),
              "d": (2000, 14, camera.ModelTransform(pos=np.array([1.5, -0.2, 0.0], np.float32), scale=np.array([0.8, 0.8, 0.8], np.float32)))}
    scenes = {k: common.small_scene(n, seed, scale_mul=8.0) for k, (n, seed, _) in models.items()}
    tr = {k: mt for k, (_, _, mt) in models.items()}
    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, 0.0, 0.5], np.float32), scale=np.array([3.0, 2.5, 3.0], np.float32)),
              MaskShape(MaskShapeKind.Ellipsoid, pos=np.array([0.2, 0.1, 0.8], np.float32), scale=np.array([1.2, 1.0, 1.4], np.float32))]
    rect = query.QueryPod.rect((60.0, 40.0), (180.0, 120.0), query.QuerySelectionOp.Set)
    edit = query.GaussianEditPod(query.GaussianEditFlag.ENABLED, (0.45, 1.2, 0.9), 0.1, 0.3, 1.0, 0.8)
    poses = (30, 31)

    def drive(v, raw):
        """the app's sequence: mask the model 'b', select by rectangle in every model, edit the selection"""
        for k, mt in tr.items():
            raw.update_model_transform(k, mt.pos, mt.quat(), mt.scale)
        MaskEvaluator(raw).evaluate(MaskOp.parse("0 - 1"), "b", shapes)
        frames = []
        for step, pose in enumerate(poses):
            cam = camera.orbit_pose(pose)
            raw.update_query(rect if step == 0 else query.QueryPod.none())
            if step == 1:
                raw.update_selection_edit_with_pod(edit)
                raw.update_selection_highlight((1.0, 0.0, 1.0, 0.3))
            v.render_frame(cam, (w, h), keys=parallel.model_render_keys(cam.pos, tr), transforms=tr)
            for k in tr:
                raw.postprocessor.postprocess(k)
            v.poll()
            frames.append(v.framebuffer().copy())
        nsel = sum(int(np.unpackbits(raw.models[k].gaussian_buffers.selection_buffer.download().view(np.uint8)).sum()) for k in tr)
        return frames, nsel

    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False)
    for k, g in scenes.items():
        single.load_shard(g, 0, g.shape[0], key=k)
    ref, nsel_ref = drive(single, single.stages.viewer)
    single.close()
    assert nsel_ref > 100 and not np.array_equal(ref[0], ref[1])

    def rank_main(rank, comm):
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm)
        for k, g in scenes.items():
            s0, c = parallel.shard_range(g.shape[0], rank, world)
            v.load_shard(g[s0:s0 + c], s0, g.shape[0], key=k)
        with v.stages.stream_ctx():
            out = drive(v, v.stages.viewer)
        v.close()
        return out

    res = common.run_ranks(world, rank_main)
    assert sum(r[1] for r in res) == nsel_ref, "the shards' selections partition the single-viewer selection"
    for rank, (frames, _) in enumerate(res):
        for i, fb in enumerate(frames):
            assert np.array_equal(fb, ref[i]), f"rank {rank} frame {i}: L-inf {np.abs(fb - ref[i]).max()}"


@pytest.mark.parametrize("world", [2, 3, 8])
def test_screen_band_mode_matches_single_viewer(world):
    """mode="screen": whole scene on every rank, rank g renders band g (speculation per band included), bands gathered:
    every rank's frame equals the single-viewer frame bit for bit, along a path with a camera jump."""
    g = _scene()
    poses = (57, 58, 59, 150, 151)
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False)
    single.load_shard(g, 0, N)
    ref = []
    for pose in poses:
        single.render_frame(camera.orbit_pose(pose), (W, H))
        single.poll()
        ref.append(single.framebuffer().copy())
    single.close()

    def rank_main(rank, comm):
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm, mode="screen")
        v.load_shard(g, 0, N)
        frames, spec = [], []
        for pose in poses:
            v.render_frame(camera.orbit_pose(pose), (W, H))
            v.poll()
            frames.append(v.framebuffer().copy())
            spec.append(v.last_stats()["speculated"])
        v.close()
        return frames, spec

    for rank, (frames, spec) in enumerate(common.run_ranks(world, rank_main)):
        assert spec == [False] + [True] * (len(poses) - 1)
        for k, fb in enumerate(frames):
            assert np.array_equal(fb, ref[k]), f"rank {rank} frame {k}: L-inf {np.abs(fb - ref[k]).max()}"


@pytest.mark.parametrize("world,overlap", [(2, False), (4, True), (3, True)])
def test_screen_band_mode_rgba8_gather(world, overlap):
    """gather="rgba8": every rank resolves its band on the device (gsx_resolve_rgba8_device) and the bands travel as 4 bytes
    a pixel, with ``overlap`` on a second stream under the next frame; the gathered frame equals the single viewer's
    gsx_download_rgba8 of the same frame, byte for byte, on every rank and for every frame of an un-synchronised run."""
    from wgpu_3dgs_viewer_app_amd.viewer import MultiModelViewer

    g = _scene()
    poses = (57, 58, 59, 150, 151, 152)
    bg = (0.25, 0.5, 0.75)
    single = MultiModelViewer()
    single.add_model("m", N)
    single.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    ref = []
    for pose in poses:
        single.update_camera(camera.orbit_pose(pose), (W, H))
        single.render_frame(["m"])
        ref.append(single.download_rgba8(bg).reshape(H, W, 4).copy())
    single.close()

    def rank_main(rank, comm):
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm, mode="screen", gather="rgba8",
                                   overlap_gather=overlap, background=bg)
        v.load_shard(g, 0, N)
        frames = []
        for k, pose in enumerate(poses):
            v.render_frame(camera.orbit_pose(pose), (W, H))
            if k >= 3:                      # the first frames run ahead without any host synchronisation
                frames.append((k, v.frame_rgba8()))
        v.close()
        return frames

    for rank, frames in enumerate(common.run_ranks(world, rank_main)):
        for k, fb in frames:
            assert fb.shape == ref[k].shape and np.array_equal(fb, ref[k]), f"rank {rank} frame {k} (world {world}, overlap {overlap})"


@pytest.mark.parametrize("world", [2, 4])
def test_frame_parallel_mode(world):
    """mode="frames": rank g renders frame r * world + g of the orbit (whole frames; its speculation looks `world` poses
    back), the resolved frames of a round are all-gathered on the second stream: every slot of every round equals the
    single viewer's gsx_download_rgba8 at that pose, on every rank, without host synchronisation between rounds."""
    from wgpu_3dgs_viewer_app_amd.viewer import MultiModelViewer

    g = _scene()
    rounds, first, bg = 4, 40, (0.1, 0.2, 0.3)
    single = MultiModelViewer()
    single.add_model("m", N)
    single.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    ref = []
    for f in range(rounds * world + 1):
        single.update_camera(camera.orbit_pose(first + f), (W, H))
        single.render_frame(["m"])
        ref.append(single.download_rgba8(bg).reshape(H, W, 4).copy())
    single.close()

    def rank_main(rank, comm):
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm, mode="frames", overlap_gather=True, background=bg)
        v.load_shard(g, 0, N)
        got = []
        for r in range(rounds):
            v.render_frame(camera.orbit_pose(first + r * world + rank), (W, H))
            if r >= 2:
                got.append((r, v.frames_rgba8()))
        spec = v.last_stats()["speculated"]
        # a last, partial round: only rank 0 has a frame, the others join the gather with the frame they rendered last
        if rank == 0:
            v.render_frame(camera.orbit_pose(first + rounds * world), (W, H))
        else:
            v.skip_frame()
        tail = v.frames_rgba8()
        v.close()
        return got, spec, tail

    for rank, (got, spec, tail) in enumerate(common.run_ranks(world, rank_main)):
        assert spec
        assert np.array_equal(tail[0], ref[rounds * world]), f"rank {rank}: partial round, slot 0"
        for slot in range(1, world):
            assert np.array_equal(tail[slot], ref[(rounds - 1) * world + slot]), f"rank {rank}: partial round, slot {slot}"
        for r, frames in got:
            assert frames.shape == (world, H, W, 4)
            for slot in range(world):
                assert np.array_equal(frames[slot], ref[r * world + slot]), f"rank {rank} round {r} slot {slot}"


@pytest.mark.parametrize("mode,world", [("index", 4), ("screen", 4)])
def test_full_size_sharded_frames(mode, world):
    """BASELINE.json's headline scene (10 M Gaussians, 1920x1080) through both multi-GPU partitionings with 4 ranks as
    threads on one device: every rank's gathered frame equals the single-viewer frame bit for bit, along the bench orbit
    and across a jump (at this size the index mode moves ~100 k records per rank and frame and repairs some frames)."""
    from wgpu_3dgs_viewer_app_amd import scene

    n, sh, w, h, seed = scene.CONFIGS["cfg4"]
    g = scene.synthetic_gaussians(n, seed, sh)
    poses = (0, 1, 2, 120, 121)
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False)
    single.load_shard(g, 0, n)
    ref = []
    for pose in poses:
        single.render_frame(camera.orbit_pose(pose), (w, h))
        single.poll()
        ref.append(single.framebuffer().copy())
    single.close()

    def rank_main(rank, comm):
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, comm=comm, mode=mode)
        if mode == "screen":
            v.load_shard(g, 0, n)
        else:
            s0, c = parallel.shard_range(n, rank, world)
            v.load_shard(g[s0:s0 + c], s0, n)
        bad, rounds = [], []
        for k, pose in enumerate(poses):
            v.render_frame(camera.orbit_pose(pose), (w, h))
            v.poll()
            if not np.array_equal(v.framebuffer(), ref[k]):
                bad.append((pose, float(np.abs(v.framebuffer() - ref[k]).max())))
            rounds.append(v.rounds)
        v.close()
        return bad, rounds

    res = common.run_ranks(world, rank_main)
    for rank, (bad, rounds) in enumerate(res):
        assert not bad, f"{mode} rank {rank}: frames differ from the single-GPU frames: {bad}"
    print(mode, "rounds", res[0][1])


def test_library_transport_world1_over_rccl(monkeypatch):
    """The product transport: the collectives inside libgsx over RCCL (gsx_viewer_comm_init, gsx_comm_all_to_all /
    _all_gather) and the whole frame as ONE library call (gsx_shard_render_frame).  A 1-GPU box has one rank, which is enough
    to run every code path of the protocol for real — slots, headers, device-side counts, verification, repair round,
    next limits, the in-place band gather — through RCCL's own send / recv / all-gather.  Frames must equal the single
    viewer's bit for bit, with the library call and with the stage calls driven from Python over the same communicator."""
    g = _scene()
    ref = _single_frames(g)
    tiles = ((H + 15) // 16, (W + 15) // 16)
    for drive in ("library", "library_self_copy", "stages", "stages_refusing"):
        # a rank's own slot is a device copy by default; here it goes through ncclSend / ncclRecv like a peer's, except in one leg
        if drive == "library_self_copy":
            monkeypatch.delenv("GSX_COMM_SELF_VIA_RCCL", raising=False)
        else:
            monkeypatch.setenv("GSX_COMM_SELF_VIA_RCCL", "1")
        v = parallel.ShardedViewer(world=1, rank=0, use_dist=True, comm="lib")
        assert isinstance(v.comm, parallel.LibComm)
        v.load_shard(g, 0, N)
        rounds = []
        for k, pose in enumerate(POSES):
            if drive == "stages":
                v.profile = {}          # the per-section profiler takes the Python-driven path over the same transport
            if drive == "stages_refusing":
                v._limit = np.full(tiles, 0x40400000, np.uint32)   # every tile refuses all but the nearest records: repair round
            v.render_frame(camera.orbit_pose(pose), (W, H))
            v.poll()
            fb = v.framebuffer()
            assert np.array_equal(fb, ref[k][0]), f"{drive}: frame {k} differs, L-inf {np.abs(fb - ref[k][0]).max()}"
            rounds.append(v.rounds)
        if drive == "stages_refusing":
            assert rounds == [2] * len(POSES)
        else:
            assert rounds[0] == 1
        lim = v.stages.limits(v.KEY)
        assert lim.shape == tiles and (lim >= 1).all()
        v.close()


@pytest.mark.parametrize("lanes", [2, 3])
def test_library_transport_with_frames_in_flight(lanes, monkeypatch):
    """gsx_shard_render_frame with gsx_render_options.frames_in_flight > 1: frame k is enqueued on lane k mod L before the verdict
    of the frame before is looked at; every collective goes through one stream in program order.  One rank over real RCCL:
    every frame, read back right away or after a run of un-synchronised calls, equals the single viewer's."""
    g = _scene()
    ref = _single_frames(g)
    monkeypatch.setenv("GSX_COMM_SELF_VIA_RCCL", "1")   # the rank's own slot through ncclSend / ncclRecv (the comm stream's hand-overs)
    v = parallel.ShardedViewer(world=1, rank=0, use_dist=True, comm="lib")
    v.stages.viewer.set_render_options(frames_in_flight=lanes)
    v.load_shard(g, 0, N)
    for k, pose in enumerate(POSES):           # readback after every frame: completes the frame just enqueued
        v.render_frame(camera.orbit_pose(pose), (W, H))
        fb = v.framebuffer()
        assert np.array_equal(fb, ref[k][0]), f"lanes {lanes}: frame {k} differs, L-inf {np.abs(fb - ref[k][0]).max()}"
    for rep in range(3):                       # runs without a host wait: the verdict of a frame is read a call later
        for k, pose in enumerate(POSES):
            v.render_frame(camera.orbit_pose(pose), (W, H))
        v.poll()
        fb = v.framebuffer()
        assert np.array_equal(fb, ref[len(POSES) - 1][0]), f"lanes {lanes}: run {rep}"
        st = v.stages.viewer.frame_stats(v.KEY)
        assert st["n_gaussians"] == N and st["overflow_slabs"] == 0
    # a change of the model data between frames reaches every lane (mask: half the Gaussians disappear, then come back)
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind

    shapes = [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -5.0, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))]
    MaskEvaluator(v.stages.viewer).evaluate(MaskOp.parse("0"), v.KEY, shapes)
    masked = []
    for pose in POSES[:4]:
        v.render_frame(camera.orbit_pose(pose), (W, H))
        masked.append(v.framebuffer())
    MaskEvaluator(v.stages.viewer).evaluate(None, v.KEY)
    for k, pose in enumerate(POSES[:4]):
        v.render_frame(camera.orbit_pose(pose), (W, H))
        fb = v.framebuffer()
        assert np.array_equal(fb, ref[k][0]), f"lanes {lanes}: after the mask reset, frame {k}"
        assert not np.array_equal(fb, masked[k]), "the mask must have changed the frames"
    v.close()


def test_verdict_reports_a_slot_that_was_too_small():
    """Stage level: a slot smaller than what the rank has for a destination -> the header says so, the verdict's overflow
    flag is set and max_records is what was wanted (ShardedViewer / gsx_shard_render_frame then redo round 0 with the
    whole-shard slot size: test_threaded_world_matches_single_viewer[tiny_slots])."""
    import torch

    g = _scene()
    st = HipStages(use_torch=True)
    st.load_shard("shard", g, 0, N)
    st.set_uniforms("shard", camera.orbit_pose(57), (W, H))
    with st.stream_ctx():
        st.frame_begin("shard", 1, 0, True, None)
        assert st.slot_records("shard", 1, N) == N          # no windows, no history: the whole-shard size
        send = st.pack_slots("shard", 1, 0, 64)              # 64 records where thousands are visible
        st.import_slots("shard", send, 1, 0, 0, 64)
        mine = st.feedback("shard", 1, 0)
        seq = st.verify("shard", 1, mine)
        verdict = st.wait_verdict("shard", seq)
        hdr = send[0, 0, :2].cpu().numpy().view(np.uint32)
    assert hdr[0] > 64 and hdr[1] == 64
    assert verdict["overflow"] and verdict["max_records"] == int(hdr[0]) and verdict["need_tiles"] == 0
    st.poll()
    st.close()
    torch.cuda.synchronize()
