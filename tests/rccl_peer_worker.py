"""One rank of tests/test_gpu_rccl_peers.py: a PROCESS of its own that renders index-sharded frames through the library's frame loop
(gsx_shard_render_frame) over a real RCCL communicator of `world` ranks — every rank on the box's one GPU.

RCCL refuses two ranks of one communicator on the same device of the same host; NCCL_HOSTID (set by the test, one value per
rank) makes every rank look like a host of its own, so the ranks talk through RCCL's network transport (sockets on `lo`): real
ncclSend / ncclRecv to a PEER, real all-gathers, two processes, the library's own control flow on both sides.  Nothing here
re-states the protocol; the worker uploads its shard, calls the library and compares what comes back with the single-viewer
frame of the whole scene, bit for bit.

usage: rccl_peer_worker.py <rank> <world> <uid file> <mode> <lanes>      exit 0 = frames identical, 77 = no communicator here"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import common  # noqa: E402
from wgpu_3dgs_viewer_app_amd import _lib, camera, parallel  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, GsxError, MultiModelViewer  # noqa: E402

N, W, H = 9000, 208, 152
POSES = (57, 58, 61, 90, 91, 200)
TILES = ((H + 15) // 16, (W + 15) // 16)


def uniforms(v, pose):
    v.update_camera(camera.orbit_pose(pose), (W, H))
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)


def connect(v, rank, world, uid_path):
    """the unique id travels through a file: rank 0 draws it, the others wait for it; 77 = no communicator to be had here"""
    if rank == 0:
        uid = (C.c_uint8 * 128)()
        try:
            _lib.check(v._L.gsx_comm_unique_id(uid))
        except GsxError as e:
            print(f"rank {rank}: no RCCL here: {e}", flush=True)
            return 77
        with open(uid_path + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(uid_path + ".tmp", uid_path)
        uid = bytes(uid)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 60:
                print(f"rank {rank}: no unique id after 60 s", flush=True)
                return 77
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    try:
        v.comm_init_rccl(world, rank, uid)
    except GsxError as e:
        print(f"rank {rank}: communicator of {world} ranks on one GPU refused: {e}", flush=True)
        return 77
    print(f"rank {rank}: communicator up", flush=True)
    return 0


def layered(rank, world, uid_path, mode, lanes):
    """four models with their own TRS, layered far -> near in an order that changes with the camera (gsx_shard_render_frame_keys)"""
    from tests import test_gpu_shard_lib as T

    scenes = T._layer_scenes()
    ref = T._layer_reference(scenes)
    v = T._layer_viewer(scenes, rank, world, None, lanes)
    rc = connect(v, rank, world, uid_path)
    if rc:
        return rc
    shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
    bad = []
    for k, pose in enumerate(T.LPOSES):
        T._uniforms(v, pose, (T.LW, T.LH))
        keys = T._layer_keys(pose)
        if mode == "layered_refusing":
            for key in keys:
                v.shard_set_limits(key, np.full(T.LTILES, 0x40400000, np.uint32))
        v.shard_render_frame_keys(keys, [shard_max[x] for x in keys])
        if lanes == 1 or k == len(T.LPOSES) - 1:
            fb = v.download_framebuffer()
            if not np.array_equal(fb, ref[k]):
                bad.append((k, float(np.abs(fb - ref[k]).max())))
    stats = v.shard_stats()
    v.close()
    print(f"rank {rank}: stats {stats}", flush=True)
    if bad or stats["wire_bytes"] == 0 or (mode == "layered_refusing" and stats["repair_frames"] != stats["frames"]):
        print(f"rank {rank}: layered frames differ / nothing sent / no repair: {bad}", flush=True)
        return 1
    print(f"rank {rank}: OK ({mode}, {lanes} lane(s), {stats['wire_bytes']} bytes on the links)", flush=True)
    return 0


def main():
    rank, world, uid_path, mode, lanes = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    if mode.startswith("layered"):
        return layered(rank, world, uid_path, mode, lanes)
    g = common.small_scene(N, 91, scale_mul=14.0)
    ref = []
    with MultiModelViewer() as v:
        v.add_model("m", N)
        v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
        for pose in POSES:
            uniforms(v, pose)
            v.render_frame(["m"])
            ref.append(v.download_framebuffer().copy())

    s0, c = parallel.shard_range(N, rank, world)
    v = MultiModelViewer()
    v.set_render_options(frames_in_flight=lanes)
    v.add_model("m", c)
    v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g[s0:s0 + c])
    rc = connect(v, rank, world, uid_path)
    if rc:
        return rc

    if mode == "root_gather":
        v.shard_set_gather_root(0)
    shard_max = (N + world - 1) // world
    if mode == "long":   # hundreds of frames in flight with jumps (repairs, redone frames), compared at a few points
        rng = np.random.default_rng(77)
        walk, pose = [], 57
        for k in range(240):
            pose = int(rng.integers(0, 240)) if rng.random() < 0.08 else (pose + 1) % 240
            walk.append(pose)
        check = {59, 119, 179, 239}
        with MultiModelViewer() as s1:
            s1.add_model("m", N)
            s1.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
            want = {}
            for k in check:
                uniforms(s1, walk[k])
                s1.render_frame(["m"])
                want[k] = s1.download_framebuffer().copy()
        bad = []
        for k, pz in enumerate(walk):
            uniforms(v, pz)
            if k % 37 == 5:
                v.shard_set_slot_records("m", 64)   # now and then a slot that overflows: the frame is redone
            v.shard_render_frame("m", shard_max)
            if k in check and not np.array_equal(v.download_framebuffer(), want[k]):
                bad.append(k)
        stats = v.shard_stats()
        v.close()
        print(f"rank {rank}: stats {stats}", flush=True)
        if bad or stats["frames"] != len(walk) or stats["repair_frames"] == 0 or stats["redo_frames"] == 0:
            print(f"rank {rank}: long run: frames {bad} differ, or no repair / redo happened", flush=True)
            return 1
        print(f"rank {rank}: OK ({mode}, {lanes} lane(s), {stats['wire_bytes']} bytes on the links)", flush=True)
        return 0
    frames, bands = [], []
    for pose in POSES:
        uniforms(v, pose)
        if mode == "all_refusing":   # verdict -> exactly sized repair round over the links
            v.shard_set_limits("m", np.full(TILES, 0x40400000, np.uint32))
        elif mode == "tiny_slots":   # overflow verdict -> round 0 redone with whole-shard slots
            v.shard_set_slot_records("m", 64)
        v.shard_render_frame("m", shard_max)
        if lanes == 1:
            frames.append(v.download_framebuffer().copy())
            bands.append(v.shard_get_band_edges(world).copy())   # (balanced bands: the layout moves from frame to frame)
    if lanes > 1:  # frames in flight: the last frame is what the viewer holds after a sync
        frames = [None] * (len(POSES) - 1) + [v.download_framebuffer().copy()]
        bands = [None] * (len(POSES) - 1) + [v.shard_get_band_edges(world).copy()]
    stats = v.shard_stats()
    v.close()
    if mode == "root_gather" and rank != 0:   # only rank 0 holds the frame; this rank still holds its own band of tile rows
        cut = [None if e is None else (min(16 * int(e[rank]), H), min(16 * int(e[rank + 1]), H)) for e in bands]
        frames = [None if fb is None else fb[cut[k][0]:cut[k][1]] for k, fb in enumerate(frames)]
        ref = [fb if cut[k] is None else fb[cut[k][0]:cut[k][1]] for k, fb in enumerate(ref)]
    bad = [k for k, fb in enumerate(frames) if fb is not None and not np.array_equal(fb, ref[k])]
    print(f"rank {rank}: stats {stats}", flush=True)
    if bad:
        print(f"rank {rank}: frames {bad} differ from the single-viewer frames (L-inf {max(np.abs(frames[k] - ref[k]).max() for k in bad)})", flush=True)
        return 1
    if stats["frames"] != len(POSES):
        print(f"rank {rank}: {stats['frames']} frames counted", flush=True)
        return 1
    if world > 1 and stats["wire_bytes"] == 0:
        print(f"rank {rank}: nothing went over the links", flush=True)
        return 1
    if mode == "all_refusing" and stats["repair_frames"] == 0:
        print(f"rank {rank}: no repair round ran", flush=True)
        return 1
    if mode == "tiny_slots" and stats["redo_frames"] == 0:
        print(f"rank {rank}: no frame was redone", flush=True)
        return 1
    print(f"rank {rank}: OK ({mode}, {lanes} lane(s), {stats['wire_bytes']} bytes on the links)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
