#!/usr/bin/env python3
"""Development: layered sharded frames that go out model by model (gsx_shard_render_frame_keys with frames in flight), for thousands of
frames: world 2 / 3 as threads over the in-process group, two and three frames in flight, the camera on the orbit with random jumps and
viewport changes, slot sizes and limits left to the library.  EVERY frame is looked at one call late in its lane's framebuffer
(gsx_debug_download_lane_framebuffer — nothing in the loop completes the frames in flight) and compared with the single-viewer frame of
the same pose; device memory must be stable.  usage: python3 tools/long_run_layers.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np  # noqa: E402

from test_gpu_shard_lib import LAYERS, LH, LW, _layer_keys, _layer_scenes, _layer_viewer, _uniforms, run_group  # noqa: E402
from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod  # noqa: E402

FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
scenes = _layer_scenes()
rng = np.random.default_rng(11)
poses = [int(rng.integers(0, 240)) if i % 41 == 0 else (7 * i // 3) % 240 for i in range(FRAMES)]
sizes = [(LW, LH) if (i // 300) % 2 == 0 else (208, 144) for i in range(FRAMES)]

ref_cache = {}
ref_viewer = _layer_viewer(scenes, 0, 1)


def reference(pose, size):
    if (pose, size) not in ref_cache:
        _uniforms(ref_viewer, pose, size)
        ref_viewer.render_frame(_layer_keys(pose))
        ref_cache[(pose, size)] = ref_viewer.download_framebuffer().copy()
    return ref_cache[(pose, size)]


for k in range(FRAMES):
    reference(poses[k], sizes[k])
ref_viewer.close()

for world, lanes in ((2, 2), (3, 3), (3, 2)):
    def body(rank, group):
        v = _layer_viewer(scenes, rank, world, group, lanes)
        shard_max = {k: (g.shape[0] + world - 1) // world for k, g in scenes.items()}
        bad = 0
        mem = None
        for k in range(FRAMES):
            _uniforms(v, poses[k], sizes[k])
            keys = _layer_keys(poses[k])
            v.shard_render_frame_keys(keys, [shard_max[x] for x in keys])
            j = k - (lanes - 1)   # the frame that call k has just retired; its lane is not used again before call k + 1
            if j >= 0:
                fb = v.debug_download_lane_framebuffer(j % lanes, sizes[j])
                if not np.array_equal(fb, reference(poses[j], sizes[j])):
                    bad += 1
            if k == FRAMES // 3:
                mem = viewer_mod.device_bytes()
        v.poll()
        grown = viewer_mod.device_bytes() - mem
        st = v.shard_stats()
        v.close()
        return bad, grown, st

    t0 = time.perf_counter()
    res = run_group(world, body, timeout_ms=120000)
    dt = time.perf_counter() - t0
    st = res[0][2]
    print(f"world {world}, {lanes} frames in flight: {FRAMES} frames in {dt:.1f} s; frames that differ from the single-viewer frame per rank: {[r[0] for r in res]}; "
          f"device bytes grown since a third of the way (process-wide): {res[0][1]}; frames {st['frames']}, repaired {st['repair_frames']}, redone {st['redo_frames']}", flush=True)
    assert all(r[0] == 0 for r in res)
print("long run OK")
