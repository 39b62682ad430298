"""Development: one robustness scene of bench.py (cfg4 orbit | random | open_sky | translucent | surfaces) on the default schedule, for a kernel trace:
rocprofv3 --kernel-trace ... -- python3 tools/trace_scene.py open_sky 160; then tools/kernel_gaps.py <dir> 60 k_project timeline 1"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

sc = sys.argv[1] if len(sys.argv) > 1 else "open_sky"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 160
n, sh, w, h, seed = scene.CONFIGS["cfg4"]
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
if sc == "random":   # bench.py's random_pose_order leg: the same 240 poses in a seeded random order (no temporal coherence)
    orbit = [orbit[k] for k in np.random.default_rng(7).permutation(240)]
g = scene.synthetic_gaussians(n, seed, sh, 0, n, variant=sc) if sc in ("translucent", "surfaces") else scene.synthetic_gaussians(n, seed, sh)
v = MultiModelViewer()
v.add_model("m", n)
v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(sh), False)
if sc == "open_sky":
    from wgpu_3dgs_viewer_app_amd.mask import MaskEvaluator, MaskOp, MaskShape, MaskShapeKind
    MaskEvaluator(v).evaluate(MaskOp.parse("0"), "m", [MaskShape(MaskShapeKind.Box, pos=np.array([0.0, -4.5, 0.0], np.float32), scale=np.array([10.0, 5.0, 10.0], np.float32))])
for i in range(60):
    v.update_camera(orbit[i % 240], (w, h))
    v.render_frame(["m"])
v.poll()
t0 = time.perf_counter()
for i in range(60, 60 + frames):
    v.update_camera(orbit[i % 240], (w, h))
    v.render_frame(["m"])
v.poll()
print(f"{sc}: {frames / (time.perf_counter() - t0):.1f} fps", v.frame_stats("m"))
v.close()
