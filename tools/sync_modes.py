"""Dev aid: what a host that WAITS for every frame gets (gsx_render_frame + gsx_sync, the reference's protocol) under
gsx_render_options.host_verify = 0 / 1 / 2, on cfg4 / cfg3 / cfg2.  usage: python tools/sync_modes.py [cfg4 cfg2 ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgpu_3dgs_viewer_app_amd import camera, scene  # noqa: E402
from wgpu_3dgs_viewer_app_amd.viewer import GaussianDisplayMode, GaussianShDegree, MultiModelViewer  # noqa: E402

for cfg in (sys.argv[1:] or ["cfg4"]):
    n, sh, w, h, seed = scene.CONFIGS[cfg]
    g = scene.synthetic_gaussians(n, seed, sh)
    orbit = [camera.PrecomputedCamera(camera.orbit_pose(k), w / h) for k in range(240)]
    v = MultiModelViewer()
    v.add_model("m", n)
    v.models["m"].gaussian_buffers.gaussians_buffer.update_range(0, g)
    v.update_gaussian_transform(1.0, GaussianDisplayMode.Splat, GaussianShDegree.new(3), False)
    for hv in (0, 1, 2, 0, 1, 2):
        v.set_render_options(host_verify=hv)
        for i in range(60):
            v.update_camera(orbit[i % 240], (w, h))
            v.render_frame(["m"])
            v.poll()
        t0 = time.perf_counter()
        K = 360
        for i in range(60, 60 + K):
            v.update_camera(orbit[i % 240], (w, h))
            v.render_frame(["m"])
            v.poll()
        print(cfg, "host_verify", hv, "synchronised fps", round(K / (time.perf_counter() - t0), 1), flush=True)
    v.close()
