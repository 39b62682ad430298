"""Development: what does every TILE of a frame's block compositor launch cost?  (GSX_TILE_PROFILE=1, gsx_debug_tile_profile.)
Prints the distribution of per-tile durations, chunks walked and takers of a few speculated cfg4 frames — the kernel is as slow as
its slowest tiles, not as the sum of its work.   usage: python tools/tile_profile.py [workload]"""
import ctypes as C
import os
import sys

import numpy as np

os.environ["GSX_TILE_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import _lib, camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

n, sh, w, h, seed = scene.CONFIGS[sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "open_sky" else "cfg4"]
g = scene.synthetic_gaussians(n, seed, sh)
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
del g
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
if "open_sky" in sys.argv[1:]:   # bench.py's robustness scene: a mask box keeps the Gaussians with y <= 0.5, the screen above the horizon stays open
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind
    MaskEvaluator(v).evaluate(MaskOp.parse("0"), "m", [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -4.5, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))])
tx, ty = (w + 15) // 16, (h + 15) // 16
for spec in (1, 0):
    v.set_render_options(speculative=spec)
    for pose in range(40):
        v.update_camera(camera.orbit_pose(pose), (w, h))
        v.render_frame(["m"])
    v.poll()
    out = np.zeros((ty * tx, 4), np.uint32)
    _lib.check(v._L.gsx_debug_tile_profile(v._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), ty * tx))
    start, dur = out[:, 0].astype(np.int64), out[:, 1].astype(np.float64) * 0.01   # us
    walked, listc, taken = out[:, 2] & 0xFFFF, out[:, 2] >> 16, out[:, 3]
    t0 = start[dur > 0].min() if (dur > 0).any() else 0
    end = ((start - t0) & 0xFFFFFFFF) * 0.01 + dur
    print(f"speculative={spec}: first slab's compositor launch: {int((dur > 0).sum())} tiles ran; last tile ends at {end[dur > 0].max():.1f} us;"
          f" sum of tile durations {dur.sum() / 1e3:.2f} ms = {dur.sum() / (256 * 12):.1f} us at 12 workgroups per CU on 256 CUs")
    for q in (50, 90, 99, 99.9, 100):
        print(f"   p{q}: duration {np.percentile(dur, q):7.1f} us, chunks walked {np.percentile(walked, q):6.0f} of {np.percentile(listc, q):6.0f} in the list, takers {np.percentile(taken, q):7.0f}")
    worst = np.argsort(-dur)[:8]
    print("   slowest tiles (x, y: us, chunks walked / in list, takers, start us):", [(int(t % tx), int(t // tx), round(float(dur[t]), 1), int(walked[t]), int(listc[t]), int(taken[t]), round(float(((start[t] - t0) & 0xFFFFFFFF) * 0.01), 1)) for t in worst])
    # list scheduling of the measured durations on 256 x 12 slots: the order the dispatcher uses (tile index) against longest first
    import heapq

    def makespan(order):
        slots = [0.0] * (256 * 12)
        heapq.heapify(slots)
        end_ = 0.0
        for t in order:
            s0 = heapq.heappop(slots)
            e0 = s0 + dur[t]
            end_ = max(end_, e0)
            heapq.heappush(slots, e0)
        return end_

    ran = np.nonzero(dur > 0)[0]
    if os.environ.get("GSX_TILE_PROFILE_DUMP"):
        np.savez_compressed(os.environ["GSX_TILE_PROFILE_DUMP"] + f"_{spec}.npz", out=out)
    print(f"   list scheduling of these durations on 3072 slots: index order {makespan(ran):.1f} us, longest first {makespan(ran[np.argsort(-dur[ran])]):.1f} us,"
          f" by takers (what a previous frame would know) {makespan(ran[np.argsort(-(taken[ran].astype(np.int64) * 4 + walked[ran] * 128), kind='stable')]):.1f} us")
    # how late do the slow tiles START?  (the dispatcher hands out tiles in index order)
    rel = ((start - t0) & 0xFFFFFFFF) * 0.01
    hist, _ = np.histogram(rel[dur > 0], bins=np.arange(0, 200, 10))
    print("   tiles starting per 10 us:", hist.tolist())
    hist, _ = np.histogram(end[dur > 0], bins=np.arange(0, 200, 10))
    print("   tiles ending per 10 us:  ", hist.tolist())
    rows = [(int(r), round(float(dur[r * tx:(r + 1) * tx].max()), 1), int(walked[r * tx:(r + 1) * tx].max()), int(taken[r * tx:(r + 1) * tx].max())) for r in range(ty)]
    print("   per tile row (row, slowest tile us, most chunks walked, most takers):", [r for r in rows if r[1] > 60])
    late = dur > np.percentile(dur, 99)
    print(f"   the slowest 1 % of the tiles start at {np.percentile((((start - t0) & 0xFFFFFFFF) * 0.01)[late], [0, 50, 100])} us (min / median / max)")
v.close()
