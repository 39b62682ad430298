"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of ``parallel.ShardedViewer`` with the oracle as the
stage backend.  Checks that index sharding + tile-row routing + strip gather reproduce the single-process
oracle frame BIT-FOR-BIT (same per-pixel blend order), i.e. the exchange logic is correct by construction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common
from tests.oracle_stages import OracleStages, pack_by_destination
from wgpu_3dgs_viewer_app_amd import camera, parallel

N, W, H, SEED, POSE = 3000, 176, 120, 81, 33


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = common.small_scene(N, SEED, scale_mul=14.0)  # opaque enough that many tiles saturate
        start, count = parallel.shard_range(N, rank, world)
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, stages=OracleStages())
        v.load_shard(g[start:start + count], start, N)
        tiles = ((H + 15) // 16, (W + 15) // 16)
        rounds, frames = [], []
        for i, pose in enumerate(POSES):
            if mode == "all_saturated":      # worst limits: every tile refuses all but the nearest records -> verified second exchange
                v._limit = np.full(tiles, 0x40400000, np.uint32)
            elif mode == "all_open":         # unbounded windows on every tile -> everything travels in one exchange
                v._limit = np.full(tiles, parallel.KEY_ALL, np.uint32)
            elif mode == "off":
                v.speculate = False
            elif mode == "tiny_slots":       # round 0's slots hold 40 records: the verdict reports the overflow, round 0 is redone
                v.force_slot = 40
            v.render_frame(camera.orbit_pose(pose), (W, H))
            if mode == "tiny_slots":
                assert v.last_verdict["overflow"] is False and v.last_verdict["max_records"] > 40
            rounds.append(v.rounds)
            if rank == 0:
                frames.append(v.framebuffer())
        if rank == 0:
            np.save(out_path, np.stack(frames))
            np.save(out_path + ".rounds.npy", np.array(rounds))
        dist.barrier()
    finally:
        dist.destroy_process_group()


POSES = (33, 36, 60)


def _single_frames():
    g = common.small_scene(N, SEED, scale_mul=14.0)
    out = []
    for pose in POSES:
        v = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=OracleStages())
        v.load_shard(g, 0, N)
        v.render_frame(camera.orbit_pose(pose), (W, H))
        out.append(v.framebuffer().copy())
    return np.stack(out)


@pytest.mark.parametrize("world,mode", [(2, "off"), (3, "natural"), (2, "all_saturated"), (3, "all_open"), (2, "natural"), (2, "tiny_slots")])
def test_sharded_frames_equal_single_process(world, mode, tmp_path):
    """Index shards + band routing + (speculative) exchange + band gather reproduce the single-process oracle frames,
    whatever the prediction: off = one full exchange; natural = frame k uses frame k-1's feedback; all_saturated forces
    the verified second exchange; all_open sends every record in one exchange."""
    ref = _single_frames()
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, mode), nprocs=world, join=True)
    fb = np.load(out)
    rounds = list(np.load(out + ".rounds.npy"))
    assert fb.shape == ref.shape
    if mode in ("off", "all_open"):
        assert rounds == [1, 1, 1]
        assert np.array_equal(fb, ref), f"sharded frames differ: L-inf {np.abs(fb - ref).max()}"
    else:
        if mode == "all_saturated":
            assert rounds == [2, 2, 2]
        else:
            print("natural rounds", rounds)
            assert rounds[0] == 1  # the first frame has no prediction: one full exchange
        # a saturated tile (every pixel T < t_epsilon = 1e-4) takes no deeper record, exactly like the product's early
        # termination: such pixels differ from the untruncated oracle frame by < t_epsilon x colour; every other pixel
        # matches up to the one extra float32 multiply-add of this adapter's (front) over (back) merge
        err = np.abs(fb - ref).max(axis=-1)
        open_px = ref[..., 3] >= 1e-4
        assert err[open_px].max() <= 2e-6
        assert err.max() <= 1e-4 * max(1.0, float(ref[..., :3].max()))
    assert ref[..., 3].min() < 1e-4, "the test scene must saturate some pixels"


def _layers_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, stages=OracleStages())
        for key, (n, seed, _) in LAYERS.items():
            g = common.small_scene(n, seed, scale_mul=9.0)
            start, count = parallel.shard_range(n, rank, world)
            v.load_shard(g[start:start + count], start, n, key=key)
        cam = camera.orbit_pose(21)
        tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
        v.render_frame(cam, (W, H), keys=parallel.model_render_keys(cam.pos, tr), transforms=tr)
        if rank == 0:
            np.save(out_path, v.framebuffer())
        dist.barrier()
    finally:
        dist.destroy_process_group()


LAYERS = {"a": (1500, 5, camera.ModelTransform(pos=np.array([0.0, 0.0, 1.5], np.float32))),
          "b": (1100, 6, common.odd_transform()),
          "c": (900, 7, camera.ModelTransform(pos=np.array([-1.0, 0.3, -2.0], np.float32), scale=np.array([0.7, 0.7, 0.7], np.float32)))}


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_layered_models_equal_single_process(world, tmp_path):
    """Several models, each with its own TRS, layered far -> near (scene.rs:533-558), through the slot / verdict protocol
    model by model (nearest first, each behind the ones before it): the sharded frame equals the single-process oracle frame
    up to the one extra float32 multiply-add of this adapter's (front) over (back) merge — the HIP path composites behind in
    place and is held to bit-identity on the GPU (tests/test_gpu_shard_lib.py)."""
    cam = camera.orbit_pose(21)
    tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
    keys = parallel.model_render_keys(cam.pos, tr)
    assert sorted(keys) == ["a", "b", "c"]
    d = [float(((tr[k].pos - cam.pos) ** 2).sum()) for k in keys]
    assert d == sorted(d, reverse=True)
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=OracleStages())
    for key, (n, seed, _) in LAYERS.items():
        single.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n, key=key)
    single.render_frame(cam, (W, H), keys=keys, transforms=tr)
    ref = single.framebuffer().copy()
    out = str(tmp_path / "fb.npy")
    mp.spawn(_layers_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    fb = np.load(out)
    err = np.abs(fb - ref).max(axis=-1)
    open_px = ref[..., 3] >= 1e-4
    assert err[open_px].max() <= 2e-6 and err.max() <= 1e-4 * max(1.0, float(ref[..., :3].max())), f"layered sharded frame differs: L-inf {err.max()}"
    # the layering matters: another order gives another image
    single.render_frame(cam, (W, H), keys=keys[::-1], transforms=tr)
    assert not np.array_equal(single.framebuffer(), ref)


def _screen_worker(rank, world, port, out_path, gather="float"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, stages=OracleStages(), mode="screen", gather=gather,
                                   background=(0.2, 0.4, 0.6))
        for key, (n, seed, _) in LAYERS.items():
            v.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n, key=key)  # the whole model on every rank
        cam = camera.orbit_pose(21)
        tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
        v.render_frame(cam, (W, H), keys=parallel.model_render_keys(cam.pos, tr), transforms=tr)
        if rank == world - 1:
            np.save(out_path, v.framebuffer() if gather == "float" else v.frame_rgba8())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _frames_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, stages=OracleStages(), mode="frames", background=(0.2, 0.4, 0.6))
        n, seed, _ = LAYERS["b"]
        v.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n)
        rounds = []
        for r in range(2):                      # two rounds of `world` consecutive frames: rank g renders frame r * world + g
            v.render_frame(camera.orbit_pose(30 + r * world + rank), (W, H))
            rounds.append(v.frames_rgba8())
        if rank == 0:
            np.save(out_path, np.stack(rounds))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_frame_parallel_mode(tmp_path):
    """mode="frames": every rank renders whole frames of its own, the resolved frames of a round are all-gathered: frame
    r * world + g of the orbit is slot g of round r, equal to the single process's frame at that pose."""
    world = 2
    n, seed, _ = LAYERS["b"]
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=OracleStages())
    single.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n)
    out = str(tmp_path / "frames.npy")
    mp.spawn(_frames_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (2, world, H, W, 4)
    for r in range(2):
        for g in range(world):
            single.render_frame(camera.orbit_pose(30 + r * world + g), (W, H))
            ref = OracleStages.resolve_rgba8(single.framebuffer(), (0.2, 0.4, 0.6)).view(np.uint8).reshape(H, W, 4)
            assert np.array_equal(got[r, g], ref), f"round {r} rank {g}"
    assert not np.array_equal(got[0, 0], got[0, 1])


def test_screen_band_mode_rgba8_gather(tmp_path):
    """gather="rgba8": the bands travel resolved against the background; the gathered frame is the single frame resolved."""
    cam = camera.orbit_pose(21)
    tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
    keys = parallel.model_render_keys(cam.pos, tr)
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=OracleStages())
    for key, (n, seed, _) in LAYERS.items():
        single.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n, key=key)
    single.render_frame(cam, (W, H), keys=keys, transforms=tr)
    ref = OracleStages.resolve_rgba8(single.framebuffer(), (0.2, 0.4, 0.6)).view(np.uint8).reshape(H, W, 4)
    out = str(tmp_path / "fb8.npy")
    mp.spawn(_screen_worker, args=(2, _free_port(), out, "rgba8"), nprocs=2, join=True)
    got = np.load(out)
    assert got.dtype == np.uint8 and np.array_equal(got, ref)
    assert got[..., 3].max() == 255 and got[..., :3].max() > 0


@pytest.mark.parametrize("world", [2, 3])
def test_screen_band_mode_equals_single_process(world, tmp_path):
    """mode="screen": every rank holds the whole scene and composites one band; the gathered frame is the single frame."""
    cam = camera.orbit_pose(21)
    tr = {k: mt for k, (_, _, mt) in LAYERS.items()}
    keys = parallel.model_render_keys(cam.pos, tr)
    single = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=OracleStages())
    for key, (n, seed, _) in LAYERS.items():
        single.load_shard(common.small_scene(n, seed, scale_mul=9.0), 0, n, key=key)
    single.render_frame(cam, (W, H), keys=keys, transforms=tr)
    out = str(tmp_path / "fb.npy")
    mp.spawn(_screen_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), single.framebuffer())


def test_limit_policy():
    import struct

    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    sat = np.zeros((6, 7), np.uint32)            # 0 = open
    sat[1:5, 1:6] = bits(4.0)
    sat[2, 3] = bits(8.0)
    lim = parallel.next_limits(sat, 0.25, 1)
    assert lim[0, 0] == parallel.KEY_ALL and lim[1, 1] == parallel.KEY_ALL      # an open neighbour: unbounded
    assert lim[3, 3] == bits(10.0) and lim[2, 3] == bits(10.0)                  # 1.25 x the deepest neighbour
    assert lim[3, 1] == parallel.KEY_ALL and lim[2, 5] == parallel.KEY_ALL
    full = np.full((3, 3), bits(4.0), np.uint32)
    assert (parallel.next_limits(full, 0.25, 2) == bits(5.0)).all()             # outside the frame counts as nothing
    w1 = parallel.windows_first(lim)
    assert (w1[..., 0] == 0).all() and np.array_equal(w1[..., 1], lim)
    need = np.zeros_like(lim, bool)
    need[3, 3] = True
    w2 = parallel.windows_second(lim, need)
    assert tuple(w2[3, 3]) == (bits(10.0), parallel.KEY_ALL) and w2.reshape(-1, 2).any(1).sum() == 1


def test_shard_ranges_cover():
    for n in (0, 1, 7, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_pack_routes_every_touched_row():
    """A record reaches exactly the ranks that own at least one tile row of its rectangle."""
    import oracle

    g = common.small_scene(2000, 82)
    cam = camera.orbit_pose(3)
    f = common.oracle_frame(cam, W, H)
    pr = oracle.project(f, *oracle.convert(g))
    vis = pr["key"] != 0xFFFFFFFF
    for world in (1, 2, 5, 8):
        send, counts = pack_by_destination(pr, world, f.tiles_x, f.tiles_y)
        assert send.shape[0] == sum(counts)
        rows = [set(range(r[1], r[3])) for r in pr["rect"][vis]]
        rpr = (f.tiles_y + world - 1) // world
        for gdst in range(world):
            expect = sum(1 for rs in rows if any(ty // rpr == gdst for ty in rs))
            assert counts[gdst] == expect
        # uniform windows split the set by depth: [0, mid) + [mid, inf) = everything, disjoint
        mid = int(np.median(pr["key"][vis]))
        lim = np.full((f.tiles_y, f.tiles_x), mid, np.uint32)
        front = pack_by_destination(pr, world, f.tiles_x, f.tiles_y, parallel.windows_first(lim))[1]
        back = pack_by_destination(pr, world, f.tiles_x, f.tiles_y, parallel.windows_second(lim, np.ones(lim.shape, bool)))[1]
        assert [a + b for a, b in zip(front, back)] == counts
        assert pack_by_destination(pr, world, f.tiles_x, f.tiles_y, parallel.windows_first(np.full_like(lim, parallel.KEY_ALL)))[1] == counts
        assert pack_by_destination(pr, world, f.tiles_x, f.tiles_y, parallel.windows_second(lim, np.zeros(lim.shape, bool)))[1] == [0] * world
        if world == 1:
            assert counts[0] == vis.sum()
