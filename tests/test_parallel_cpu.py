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


def _single_frame():
    g = common.small_scene(N, SEED)
    st = OracleStages()
    v = parallel.ShardedViewer(world=1, rank=0, use_dist=False, stages=st)
    v.load_shard(g, 0, N)
    v.render_frame(camera.orbit_pose(POSE), (W, H))
    return v.framebuffer().copy()


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = common.small_scene(N, SEED)
        start, count = parallel.shard_range(N, rank, world)
        v = parallel.ShardedViewer(world=world, rank=rank, use_dist=True, stages=OracleStages())
        v.load_shard(g[start:start + count], start, N)
        for pose in (POSE, POSE):  # twice: buffers are reused across frames
            v.render_frame(camera.orbit_pose(pose), (W, H))
        if rank == 0:
            np.save(out_path, v.framebuffer())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process(world, tmp_path):
    ref = _single_frame()
    out = str(tmp_path / "fb.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    fb = np.load(out)
    assert fb.shape == ref.shape
    assert np.array_equal(fb, ref), f"sharded frame differs: L-inf {np.abs(fb - ref).max()}"


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
        send, counts = pack_by_destination(pr, world)
        assert send.shape[0] == sum(counts)
        rows = [set(range(r[1], r[3])) for r in pr["rect"][vis]]
        for gdst in range(world):
            expect = sum(1 for rs in rows if any(ty % world == gdst for ty in rs))
            assert counts[gdst] == expect
        if world == 1:
            assert counts[0] == vis.sum()
