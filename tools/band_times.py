#!/usr/bin/env python3
"""What one rank of an N-GPU screen-band frame costs on a GPU of its own: renders band g of N of the bench scene alone and
times it (frames enqueued back to back, like bench.py).  The slowest band + the band all-gather is the N-GPU frame.
Dev tool: the 1-GPU box cannot run N ranks at once without contention, but it can run them one after the other."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import _lib, camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--workload", default="cfg4")
args = ap.parse_args()
n, sh, w, h, seed = scene.CONFIGS[args.workload]
g = scene.synthetic_gaussians(n, seed, sh)
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
del g
out = []
for rank in range(args.world):
    lay = _lib.ShardLayout()
    _lib.check(v._L.gsx_shard_layout(v._h, args.world, rank, C.byref(lay)))
    v.update_camera(camera.orbit_pose(0), (w, h))
    _lib.check(v._L.gsx_shard_layout(v._h, args.world, rank, C.byref(lay)))
    _lib.check(v._L.gsx_viewer_set_band(v._h, lay.row_lo, lay.row_hi))

    def frame(i):
        v.update_camera(camera.orbit_pose(i), (w, h))
        v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
        v.render_frame(["m"])

    for i in range(10):
        frame(i)
    v.poll()
    t0 = time.perf_counter()
    for i in range(args.frames):
        frame(10 + i)
    v.poll()
    ms = 1e3 * (time.perf_counter() - t0) / args.frames
    st = v.frame_stats("m")
    out.append(ms)
    print(f"band {rank}/{args.world} rows [{lay.row_lo},{lay.row_hi}): {ms:.3f} ms/frame  n_visible {st['n_visible']} n_sorted {st['n_sorted']} "
          f"entries {st['n_tile_entries']}")
print(f"world {args.world}: slowest band {max(out):.3f} ms -> <= {1e3 / max(out):.0f} fps before the band all-gather; mean {sum(out) / len(out):.3f} ms")
