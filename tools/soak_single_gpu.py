"""Development: a long single-GPU run of the default schedule (frames in flight, speculation, tuner probes, random pose jumps) — every
`check` frames the streamed frame is compared with the frame a second viewer renders unspeculated in one pass.  Looks for what short tests
cannot: a look-back kernel that never finishes, an epoch that comes round, buffers that keep growing.
usage: python tools/soak_single_gpu.py [frames] [lanes] [gaussians]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd import viewer as viewer_mod  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2000000
w, h, check = 1920, 1080, 997
g = scene.synthetic_gaussians(n, 11, 3)
orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
v, ref = MultiModelViewer(), MultiModelViewer()
for x in (v, ref):
    x.add_model("m", n)
    x.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    x.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
v.set_render_options(frames_in_flight=lanes)
ref.set_render_options(speculative=0, progressive=0)
rng = np.random.default_rng(5)
pose, bad, checked, bytes_third = 0, 0, 0, None
t0 = time.perf_counter()
for i in range(frames):
    pose = int(rng.integers(240)) if rng.random() < 0.02 else (pose + 1) % 240   # an orbit with a jump every ~50 frames
    v.update_camera(orbit[pose], (w, h))
    v.render_frame(["m"])
    if i % check == check - 1:
        a = v.download_framebuffer().copy()
        ref.update_camera(orbit[pose], (w, h))
        ref.render_frame(["m"])
        b = ref.download_framebuffer()
        checked += 1
        bad += 0 if np.array_equal(a, b) else 1
    if i == frames // 3:
        bytes_third = viewer_mod.device_bytes()
v.poll()
el = time.perf_counter() - t0
print(f"soak: {frames} frames, {lanes} lanes, {n} Gaussians in {el:.1f} s ({frames / el:.0f} fps); {checked} frames compared with the single-pass frame, {bad} differ; "
      f"device bytes grown since a third of the way: {viewer_mod.device_bytes() - bytes_third}; stats {v.frame_stats('m')}")
v.close()
ref.close()
sys.exit(1 if bad else 0)
